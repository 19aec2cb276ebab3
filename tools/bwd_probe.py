#!/usr/bin/env python3
"""PoseNet backward in isolation: loss = <poses, fixed random vector>; per-parameter L2-relative gradient error of the HIP net and of the
CPU fp32 oracle net against the float64 oracle net.  usage: python tools/bwd_probe.py [worktree root]"""
import os
import sys

ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import conftest  # noqa: F401,E402
import torch  # noqa: E402
import test_step_gpu as T  # noqa: E402
from arbiter import double_copy, to_double  # noqa: E402
from oracle.step import synthetic_batch  # noqa: E402

DEV = "cuda"
hip_d, hip_p, ref_d, ref_p = T.build_pair(layers=18)
p64 = double_copy(ref_p)
s = synthetic_batch(2, 64, 128, seed=5)
s64 = to_double(s)
g = torch.Generator().manual_seed(3)
wvec = torch.randn(2, 2, 6, generator=g)


def l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())


(hip_p(s["tgt"].to(DEV), [r.to(DEV) for r in s["ref_imgs"]]) * wvec.to(DEV)).sum().backward()
(ref_p(s["tgt"], s["ref_imgs"]) * wvec).sum().backward()
(p64(s64["tgt"], s64["ref_imgs"]) * wvec.double()).sum().backward()
hp, rp, dp = dict(hip_p.named_parameters()), dict(ref_p.named_parameters()), dict(p64.named_parameters())
for n in dp:
    print("%-22s |HIP-fp64| %.3e   |CPU32-fp64| %.3e" % (n, l2(hp[n].grad, dp[n].grad), l2(rp[n].grad, dp[n].grad)))
