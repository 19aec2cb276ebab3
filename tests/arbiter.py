"""fp64 arbiter for the gradient tolerances (test infrastructure).

The loss is only piecewise smooth (L1 sign, bilinear cell edges, ReLU / max-pool ties), so two correct fp32 evaluations differ by more than
rounding in a few elements.  To tell that apart from a real indexing error, the same oracle code is evaluated in float64 and both fp32
results -- the HIP path's and the CPU oracle's -- are measured against it: the HIP error must not exceed what stock fp32 PyTorch shows.
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
        self.rows, self.factor, self.floor = [], factor, floor

    def add(self, name, hip, cpu32, ref64):
        self.rows.append((name, l2_rel(hip, ref64), l2_rel(cpu32, ref64)))

    def worst(self):
        return max(self.rows, key=lambda r: r[1] / max(r[2], self.floor))

    def table(self, top=12):
        rows = sorted(self.rows, key=lambda r: -r[1] / max(r[2], self.floor))[:top]
        return "\n".join("  %-52s |HIP-fp64| %.3e   |CPUfp32-fp64| %.3e   ratio %.2f" % (n, a, b, a / max(b, self.floor)) for n, a, b in rows)

    def check(self, what=""):
        print("fp64 arbiter %s (L2-relative errors, worst ratios first):\n%s" % (what, self.table()))
        bad = [(n, a, b) for n, a, b in self.rows if a > self.factor * max(b, self.floor)]
        assert not bad, "HIP further from fp64 than %gx the CPU fp32 oracle: %s" % (self.factor, bad[:5])
