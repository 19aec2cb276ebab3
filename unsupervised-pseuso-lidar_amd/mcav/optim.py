"""Fused Adam over the flat arena (replaces optim.Adam at reference trainer.py:75; defaults as torch.optim.Adam)."""
import torch

from . import lib as L
from . import nn as N
from .arena import arena_of


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._arena = None
        self._m = self._v = None
        self._step = 0
        self._step_t = torch.zeros(())          # shared by every state entry (torch.optim.Adam keeps one per parameter)
        self.grad_scale = 1.0          # 1/world_size under data parallelism (gradients are summed by the all-reduce)

    def arena(self):
        params = [p for g in self.param_groups for p in g["params"]]
        if self._arena is None or not self._arena.intact():
            self._arena = arena_of(params)
            self._m = torch.zeros_like(self._arena.flat)
            self._v = torch.zeros_like(self._arena.flat)
            for p, o in zip(self._arena.params, self._arena.offsets):
                n = p.numel()
                self.state[p] = {"step": self._step_t, "exp_avg": self._m[o:o + n].view(p.shape),
                                 "exp_avg_sq": self._v[o:o + n].view(p.shape)}
        return self._arena

    def state_dict(self):
        self._step_t.fill_(float(self._step))          # (step_capturable / graph replays advance the count on the device)
        return super().state_dict()

    def zero_grad(self, set_to_none=False):
        a = self.arena()
        a.attach_grads()
        a.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        a = self.arena()
        g = self.param_groups[0]
        self._step += 1
        b1, b2 = g["betas"]
        L.check(L.lib().mcav_adam_step(N.P(a.flat), N.P(a.gflat), N.P(self._m), N.P(self._v), a.numel, g["lr"], b1, b2, g["eps"], self._step,
                                       self.grad_scale, L.stream()), "mcav_adam_step")
        a.bump()
        self._step_t.fill_(float(self._step))

    # ---- capturable form: the per-step scalars live in a device record, so the launch can sit inside a hipGraph (mcav/graph.py)
    def device_state(self):
        """float32[8] on the arena's device: [step count, lr, grad_scale, ...]; synchronised from the host values on every call."""
        a = self.arena()
        if getattr(self, "_dev_state", None) is None or self._dev_state.device != a.flat.device:
            self._dev_state = torch.zeros(8, dtype=torch.float32, device=a.flat.device)
            self._dev_mirror = None
        want = (float(self._step), float(self.param_groups[0]["lr"]), float(self.grad_scale))
        if self._dev_mirror != want:              # one small copy, only when a host-side value moved (lr schedule, resume, world size)
            self._dev_state[:3].copy_(torch.tensor(want, dtype=torch.float32), non_blocking=False)
            self._dev_mirror = want
        return self._dev_state

    @torch.no_grad()
    def step_capturable(self):
        """The same update as step(), reading step count / lr / grad_scale from device memory (mcav_adam_step_dev): valid under
        hipGraph capture and replay.  Call note_replayed() after each replay so the host-side counters follow."""
        a = self.arena()
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        st = self.device_state()
        L.check(L.lib().mcav_adam_step_dev(N.P(a.flat), N.P(a.gflat), N.P(self._m), N.P(self._v), a.numel, b1, b2, g["eps"], N.P(st), L.stream()),
                "mcav_adam_step_dev")
        self.note_replayed()

    def note_replayed(self):
        """Host bookkeeping of one update done on the device (eagerly by step_capturable or by a graph replay that contains it)."""
        self._step += 1
        if getattr(self, "_dev_mirror", None) is not None:
            self._dev_mirror = (float(self._step),) + self._dev_mirror[1:]
        self._arena.bump()

    def load_state_dict(self, state_dict):
        """Accepts the dict torch.optim.Adam writes (reference trainer.py:136,148: 'optimizer_state_dict' of a checkpoint):
        the moments are copied INTO the flat buffers (the fused kernel keeps reading those) and the step count is restored."""
        a = self.arena()                       # first: creating the arena (re)initialises the per-parameter state entries
        super().load_state_dict(state_dict)
        step = 0
        for p, o in zip(a.params, a.offsets):
            st = self.state.get(p)
            n = p.numel()
            if st:
                if "exp_avg" in st:
                    self._m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                    self._v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                if "step" in st:
                    step = max(step, int(float(st["step"])))
            self.state[p] = {"step": self._step_t, "exp_avg": self._m[o:o + n].view(p.shape), "exp_avg_sq": self._v[o:o + n].view(p.shape)}
        self._step = step
        self._step_t.fill_(float(step))
