"""A small define-by-run tape over the HIP ops, used by the secondary networks (DispNetS, PoseFc).

The primary nets (DispResNet, PoseNet) have hand-scheduled backward passes (mcav/depthnet.py, mcav/posenet.py) with
fused epilogues.  DispNetS (reference models/depth/disp_net.py:51-141) and PoseFc (models/pose/pose_fc.py:21-84) have
irregular graphs (3-way concats with a 1-channel disparity map, cropped transposed convolutions, BatchNorm after ReLU),
so they are expressed op by op; every op is still one of the library's kernels and records its own backward here.
All tensors are NHWC fp32 on the GPU.
"""
import torch

from . import lib as L
from . import nn as N

L.register({
    "mcav_copy_channels": (L.c_i, [L.c_p, L.c_sz, L.c_i, L.c_i, L.c_p, L.c_i, L.c_i, L.c_i, L.c_i, L.c_p]),
    "mcav_resize_bilinear_fwd": (L.c_i, [L.c_p, L.c_i, L.c_i, L.c_i, L.c_p, L.c_i, L.c_i, L.c_f, L.c_f, L.c_p]),
    "mcav_resize_bilinear_bwd": (L.c_i, [L.c_p, L.c_i, L.c_i, L.c_i, L.c_p, L.c_i, L.c_i, L.c_f, L.c_f, L.c_i, L.c_p]),
    "mcav_affine": (L.c_i, [L.c_p, L.c_f, L.c_f, L.c_sz, L.c_p, L.c_p]),
    "mcav_mul": (L.c_i, [L.c_p, L.c_p, L.c_sz, L.c_p, L.c_p]),
    "mcav_colsum_workspace_bytes": (L.c_sz, [L.c_i]),
    "mcav_colsum": (L.c_i, [L.c_p, L.c_sz, L.c_i, L.c_p, L.c_i, L.c_p, L.c_sz, L.c_p]),
})

P = N.P


class Tape:
    def __init__(self, enabled=True):
        self.steps = []
        self.grads = {}
        self.enabled = enabled

    def record(self, fn):
        if self.enabled:
            self.steps.append(fn)

    def add_grad(self, t, g):
        k = id(t)
        cur = self.grads.get(k)
        self.grads[k] = g if cur is None else N.add(cur, g)

    def grad(self, t):
        return self.grads.get(id(t))

    def backward(self, seeds):
        """seeds: list of (tensor, gradient)."""
        for t, g in seeds:
            self.add_grad(t, g)
        for fn in reversed(self.steps):
            fn()
        self.steps = []


# ------------------------------------------------------------------------------------------------ ops
def conv(tape, spec, x, act=N.ACT_NONE, stats=False, x_needs_grad=True):
    out = N.conv_fwd(spec, x, act=act, stats=stats)
    y, slab = out if stats else (out, None)

    def bwd():
        gy = tape.grad(y)
        if gy is None:
            return
        dpre = gy if act == N.ACT_NONE else N.act_bwd(gy, y, act)
        N.conv_wgrad(spec, x, dpre)
        if x_needs_grad:       # x may carry zero channels past spec.cin (concat padding): their gradient rows are zero filters
            tape.add_grad(x, N.conv_dgrad(spec, dpre, (x.shape[1], x.shape[2]), n_count=x.shape[3]))
    tape.record(bwd)
    return (y, slab) if stats else y


class DeconvSpec:
    """ConvTranspose2d(3, stride 2, pad 1, output_padding 1) parameters [Cin, Cout, 3, 3] seen as three conv problems."""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias
        cin, cout, kh, kw = weight.shape
        self.cin, self.cout = cin, cout
        # forward gather-GEMM (ADJ_STRIDE2): rows = Cout, K = Cin: the weight viewed as OIHW [O=Cin][I=Cout] packed transposed
        self.as_conv = N.ConvSpec(weight, None, 2, 1, N.PAD_ZERO)      # O = Cin, I = Cout: its forward conv IS the deconv's dgrad


def deconv(tape, ds, x, out_hw, act=N.ACT_RELU):
    """y[:, :H, :W] of the transposed conv (the reference crops to a skip tensor's size: disp_net.py:46-48,106-136)."""
    import ctypes
    B, Hs, Ws, C = x.shape
    H, W = out_hw
    spec = ds.as_conv                                   # cout(spec) = Cin of the deconv, cin(spec) = Cout of the deconv
    y = N.empty((B, H, W, ds.cout), x)
    d = N.IgemmDesc()
    d.x1, d.x2 = P(x), None
    d.B, d.Hs, d.Ws, d.C1, d.C2, d.up1 = B, Hs, Ws, C, 0, 0
    d.w = P(spec.packed_bwd())                          # [n = I (deconv Cout)][tap][k = O (deconv Cin)]
    d.kh, d.kw, d.Np, d.Kp = 3, 3, N.up16(ds.cout), N.up16(ds.cin)
    d.mode, d.stride, d.sign, d.offset, d.pad_mode = N.G_ADJ_STRIDE2, 2, -1, 1, N.PAD_ZERO
    d.y, d.Hd, d.Wd, d.Cd, d.n_begin, d.n_count, d.y_choff = P(y), H, W, ds.cout, 0, ds.cout, 0
    d.bias, d.act = P(ds.bias), act
    L.check(L.lib().mcav_igemm(ctypes.byref(d), L.stream()), "mcav_igemm(deconv)")

    def bwd():
        gy = tape.grad(y)
        if gy is None:
            return
        dpre = gy if act == N.ACT_NONE else N.act_bwd(gy, y, act)
        # weight gradient [Cin][Cout][3][3]: "source" = dpre gathered with stride 2, "dy" = x
        gw = N.grad_buffer(ds.weight)
        wd = N.WgradDesc()
        wd.x1, wd.x2 = P(dpre), None
        wd.B, wd.Hs, wd.Ws, wd.C1, wd.C2, wd.up1 = B, H, W, ds.cout, 0, 0
        wd.kh, wd.kw, wd.Kp = 3, 3, N.up16(ds.cout)
        wd.mode, wd.stride, wd.sign, wd.offset, wd.pad_mode = N.G_DIRECT, 2, 1, -1, N.PAD_ZERO
        wd.dy, wd.Hd, wd.Wd, wd.Cdy, wd.dy_choff = P(x), Hs, Ws, C, 0
        wd.Cout, wd.Cin = ds.cin, ds.cout
        wd.dw_oihw, wd.accumulate, wd.dbias, wd.tile = P(gw), 1, None, 0
        N.launch_wgrad(wd, (dpre, x))
        colsum(dpre, N.grad_buffer(ds.bias), accumulate=True)
        # data gradient: the ordinary stride-2 conv of dpre with the same weights (O = Cin)
        tape.add_grad(x, N.conv_fwd(spec, dpre))
    tape.record(bwd)
    return y


def colsum(x, out, accumulate=False):
    C = x.shape[-1]
    h = L.lib()
    ws = L.workspace(h.mcav_colsum_workspace_bytes(C), x.device, "colsum")
    L.check(h.mcav_colsum(P(x), x.numel() // C, C, P(out), int(accumulate), P(ws), ws.numel(), L.stream()), "mcav_colsum")


def batchnorm(tape, bn, x, slab, train):
    """BatchNorm2d on an activation whose per-tile statistics came from the producing conv's epilogue (no ReLU after it)."""
    if train:
        st = N.bn_train_coeffs(bn, slab, x.shape[0] * x.shape[1] * x.shape[2])
    else:
        st = N.bn_eval_coeffs(bn)
    y = N.bn_apply(x, st, False)

    def bwd():
        gy = tape.grad(y)
        if gy is not None:
            tape.add_grad(x, N.bn_backward(bn, st, gy, None, x, False))
    tape.record(bwd)
    return y


def concat(tape, parts):
    """Channel concat into a buffer whose channel count is rounded up to 4 (zero filled)."""
    B, H, W, _ = parts[0].shape
    C = sum(p.shape[3] for p in parts)
    Cp = (C + 3) // 4 * 4
    out = torch.zeros((B, H, W, Cp), dtype=torch.float32, device=parts[0].device) if Cp != C else N.empty((B, H, W, Cp), parts[0])
    h = L.lib()
    off = 0
    offs = []
    for p_ in parts:
        c = p_.shape[3]
        L.check(h.mcav_copy_channels(P(p_), B * H * W, c, 0, P(out), Cp, off, c, 0, L.stream()), "mcav_copy_channels")
        offs.append(off)
        off += c

    def bwd():
        g = tape.grad(out)
        if g is None:
            return
        for p_, o in zip(parts, offs):
            c = p_.shape[3]
            gp = N.empty(tuple(p_.shape), p_)
            L.check(h.mcav_copy_channels(P(g), B * H * W, g.shape[3], o, P(gp), c, 0, c, 0, L.stream()), "mcav_copy_channels")
            tape.add_grad(p_, gp)
    tape.record(bwd)
    return out


def resize_bilinear(tape, x, out_hw, scale=None):
    """1-channel map [B,h,w,1] -> [B,H,W,1], align_corners=False.  scale = source step per output pixel (None: h/H, w/W)."""
    B, h_, w_, C = x.shape
    assert C == 1
    H, W = out_hw
    sy, sx = (scale, scale) if scale is not None else (0.0, 0.0)
    y = N.empty((B, H, W, 1), x)
    L.check(L.lib().mcav_resize_bilinear_fwd(P(x), B, h_, w_, P(y), H, W, sy, sx, L.stream()), "mcav_resize_bilinear_fwd")

    def bwd():
        g = tape.grad(y)
        if g is not None:
            gx = N.empty(tuple(x.shape), x)
            L.check(L.lib().mcav_resize_bilinear_bwd(P(g), B, h_, w_, P(gx), H, W, sy, sx, 0, L.stream()), "mcav_resize_bilinear_bwd")
            tape.add_grad(x, gx)
    tape.record(bwd)
    return y


def affine(tape, x, a, b):
    y = torch.empty_like(x)
    L.check(L.lib().mcav_affine(P(x), a, b, x.numel(), P(y), L.stream()), "mcav_affine")

    def bwd():
        g = tape.grad(y)
        if g is not None:
            gx = torch.empty_like(x)
            L.check(L.lib().mcav_affine(P(g), a, 0.0, g.numel(), P(gx), L.stream()), "mcav_affine")
            tape.add_grad(x, gx)
    tape.record(bwd)
    return y


def mul_const(tape, x, mask):
    y = torch.empty_like(x)
    L.check(L.lib().mcav_mul(P(x), P(mask), x.numel(), P(y), L.stream()), "mcav_mul")

    def bwd():
        g = tape.grad(y)
        if g is not None:
            gx = torch.empty_like(x)
            L.check(L.lib().mcav_mul(P(g), P(mask), g.numel(), P(gx), L.stream()), "mcav_mul")
            tape.add_grad(x, gx)
    tape.record(bwd)
    return y
