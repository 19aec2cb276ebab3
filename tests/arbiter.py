"""fp64 arbiter for the gradient tolerances (test infrastructure).

The loss is only piecewise smooth (L1 sign, bilinear cell edges, ReLU / max-pool ties), so two correct fp32 evaluations differ by more than
rounding in a few elements.  To tell that apart from a real indexing error, the same oracle code is evaluated in float64 and both fp32
results -- the HIP path's and the CPU oracle's -- are measured against it.  Rule, per tensor (L2-relative errors):

    |HIP - fp64|  <=  factor * max(|CPUfp32 - fp64|, envelope, floor)

`envelope` is the conditioning of the problem itself: the float64 evaluation repeated on inputs and weights perturbed by 1e-6 relative
noise (the size of the rounding error fp32 accumulates through the layers).  A ReLU unit or an L1 residual within that noise of its kink
flips in SOME fp32 evaluation -- which one is chance (measured on MI355X: HIP 6e-6 vs CPU 2.9e-3 on one pose gradient, HIP 3.5e-4 vs CPU
5e-6 on one stem gradient, both a single flipped unit) -- and moves every gradient downstream of it by the same discrete amount in the
perturbed float64 run.  An indexing error of a per cent in one layer stays far outside that envelope.

The step tests add to the envelope the CPU fp32 oracle's own step on inputs perturbed by 1e-5: the pose gradient of the loss moves in
discrete steps of 2e-4 .. 1.5e-3 (one pixel's L1 sign / SSIM clamp / in-view test changing side), and float64 perturbed by 1e-6 does not
reach those pixels while either fp32 evaluation does (tools/flip_probe.py; numbers in tests/test_step_gpu.py).
"""
import copy

import torch


def l2_rel(x, ref):
    x = torch.as_tensor(x).detach().cpu().double().reshape(-1)
    ref = torch.as_tensor(ref).detach().cpu().double().reshape(-1)
    return float((x - ref).norm() / ref.norm().clamp_min(1e-300))


def to_double(obj):
    if torch.is_tensor(obj):
        return obj.detach().double()
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_double(o) for o in obj)
    if isinstance(obj, dict):
        return {k: to_double(v) for k, v in obj.items()}
    return obj


def double_copy(module):
    return copy.deepcopy(module).double()


class Verdicts:
    """Collects (name, err_hip, err_cpu32) triples; check() applies the rule and prints the table (shown with pytest -s or on failure)."""

    def __init__(self, factor=2.0, floor=2e-6):
        """floor: what every fp32 evaluation is allowed regardless of the other two columns.  The loss-only tests keep 2e-6; the network tests
        use 2.5e-4: cancellation in fp32 (SSIM's E[x^2] - mu^2, BatchNorm variances over a few dozen samples) puts individual tensors of
        either fp32 evaluation at the 1e-4 level (measured: the SSIM pose gradient HIP 2.0e-4 / CPU 2.6e-5; the same tensor under L1
        HIP 7.8e-4 / CPU 7.2e-4), two orders below what an indexing error produces."""
        self.rows, self.factor, self.floor = [], factor, floor

    def add(self, name, hip, cpu32, ref64, perturbed64=()):
        """perturbed64: the same tensor from oracle runs on perturbed inputs (the envelope: float64 at 1e-6, and where a test adds them the
        fp32 oracle at 1e-5); may be empty."""
        env = max([l2_rel(p, ref64) for p in perturbed64], default=0.0)
        self.rows.append((name, l2_rel(hip, ref64), l2_rel(cpu32, ref64), env))

    def _allow(self, r):
        return max(r[2], r[3], self.floor)

    def worst_hip(self):
        return max(r[1] for r in self.rows)

    def table(self, top=12):
        rows = sorted(self.rows, key=lambda r: -r[1] / self._allow(r))[:top]
        return "\n".join("  %-50s |HIP-fp64| %.3e  |CPUfp32-fp64| %.3e  envelope %.3e  ratio %.2f" % (n, a, b, e, a / self._allow((n, a, b, e)))
                         for n, a, b, e in rows)

    def check(self, what="", hip_abs=None):
        """hip_abs: optional absolute bound on every |HIP - fp64| (what the table showed on MI355X, with margin)."""
        print("fp64 arbiter %s (L2-relative errors, worst ratios first; worst |HIP-fp64| %.3e):\n%s" % (what, self.worst_hip(), self.table()))
        bad = [r for r in self.rows if r[1] > self.factor * self._allow(r)]
        assert not bad, "HIP further from fp64 than %gx max(CPU fp32 oracle, 1e-6-perturbation envelope): %s" % (self.factor, bad[:5])
        if hip_abs is not None:
            over = [r for r in self.rows if r[1] > hip_abs]
            assert not over, "|HIP - fp64| above %g: %s" % (hip_abs, over[:5])


def perturb_(module, rel, seed):
    """In place: every floating-point parameter *= 1 + rel * N(0, 1) (a float64 copy of a network)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.mul_(1.0 + rel * torch.randn(p.shape, generator=g, dtype=p.dtype))
    return module


def perturb_tensor(t, rel, seed):
    g = torch.Generator().manual_seed(seed)
    return t * (1.0 + rel * torch.randn(t.shape, generator=g, dtype=t.dtype))
