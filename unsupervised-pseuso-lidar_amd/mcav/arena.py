"""Flat parameter / gradient arenas.

All parameters of the depth and pose nets are re-pointed at views of ONE contiguous fp32 buffer, and their .grad
at views of a second one.  That gives: a single fused-Adam launch per step, a single RCCL all-reduce of the
gradients per step (63.7 MB for the R18 configuration), and a one-kernel zero_grad.  Parameters
keep the reference's names and shapes, so state_dict()/load_state_dict() are unchanged.
"""
import torch


class Arena:
    def __init__(self, params):
        params = [p for p in params]
        if not params:
            raise ValueError("Arena: no parameters")
        dev = params[0].device
        for p in params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("Arena: all parameters must be fp32 on one device")
        self.params = params
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4           # 16-byte aligned slots
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.epoch = 0
        for p, o in zip(params, self.offsets):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p._mcav_arena = self
            p._mcav_grad_view = (lambda o=o, n=n, shape=p.shape: self.gflat[o:o + n].view(shape))
            p._mcav_epoch = (lambda: self.epoch)
            p.grad = p._mcav_grad_view()

    def intact(self):
        """False once something (e.g. module.to()) has re-allocated a parameter behind our back."""
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    def attach_grads(self):
        base = self.gflat.data_ptr()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != base + 4 * o:
                p.grad = p._mcav_grad_view()

    def zero_grad(self):
        self.gflat.zero_()

    def bump(self):
        """Parameters were changed through raw pointers (fused Adam): invalidate packed weight copies."""
        self.epoch += 1


def arena_of(params):
    """The arena that holds exactly these parameters (in any order), creating it when needed."""
    params = list(params)
    a = getattr(params[0], "_mcav_arena", None)
    if a is not None and a.intact() and len(a.params) == len(params) and all(getattr(p, "_mcav_arena", None) is a for p in params):
        return a
    return Arena(params)
