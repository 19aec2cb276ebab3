#!/usr/bin/env python3
"""Forward deviation of the HIP nets from the float64 oracle nets (same weights), beside the CPU fp32 oracle's: poses, disparities,
and the gradient the loss hands to the pose net.  usage: python tools/fwd_probe.py [worktree root]"""
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
d64, p64 = double_copy(ref_d), double_copy(ref_p)
s = synthetic_batch(2, 64, 128, seed=5)
s64 = to_double(s)
tgt, refs = s["tgt"], s["ref_imgs"]


def l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())


with torch.no_grad():
    p_hip = hip_p(tgt.to(DEV), [r.to(DEV) for r in refs])
    p_32 = ref_p(tgt, refs)
    p_64 = p64(s64["tgt"], s64["ref_imgs"])
    print("poses   |HIP-fp64| %.3e   |CPU32-fp64| %.3e" % (l2(p_hip, p_64), l2(p_32, p_64)))
    d_hip = hip_d(tgt.to(DEV))
    d_32 = ref_d(tgt)
    d_64 = d64(s64["tgt"])
    for i in range(len(d_64)):
        print("disp[%d] |HIP-fp64| %.3e   |CPU32-fp64| %.3e" % (i, l2(d_hip[i], d_64[i]), l2(d_32[i], d_64[i])))

# the stacked tgt / ref0 pass (per-image-set BatchNorm statistics through the conv epilogues)
with torch.no_grad():
    hip_d2, _, ref_d2, _ = T.build_pair(layers=18)
    d64b = double_copy(ref_d2)
    pa, pb = hip_d2.forward_pair(tgt.to(DEV), refs[0].to(DEV))
    qa, qb = d64b(s64["tgt"]), d64b(s64["ref_imgs"][0])
    ra_, rb_ = ref_d2(tgt), ref_d2(refs[0])
    print("pair tgt  disp[0] |HIP-fp64| %.3e   |CPU32-fp64| %.3e" % (l2(pa[0], qa[0]), l2(ra_[0], qa[0])))
    print("pair ref0 disp[0] |HIP-fp64| %.3e   |CPU32-fp64| %.3e" % (l2(pb[0], qb[0]), l2(rb_[0], qb[0])))
