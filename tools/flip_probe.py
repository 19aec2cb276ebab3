#!/usr/bin/env python3
"""How sensitive is dL/dposes to last-bit differences of the loss's inputs?  The SSIM step case of test_train_step_vs_oracle:
the HIP loss kernel and the CPU oracle (fp32 and fp64) on IDENTICAL inputs (the HIP nets' disparities and poses), then on inputs perturbed
by a few 1e-6 relative.  usage: python tools/flip_probe.py [root]"""
import os
import sys

ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import conftest  # noqa: F401,E402
import torch  # noqa: E402
import test_step_gpu as T  # noqa: E402
from arbiter import perturb_tensor  # noqa: E402
from losses import Losses  # noqa: E402
from oracle import losses as ol  # noqa: E402
from oracle.step import synthetic_batch  # noqa: E402

DEV = "cuda"


def l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())


for ssim in (False, True):
    hip_d, hip_p, ref_d, ref_p = T.build_pair(layers=18)
    s = synthetic_batch(2, 64, 128, seed=5)
    tgt, refs, K = s["tgt"], s["ref_imgs"], s["intrinsics"]
    with torch.no_grad():
        da, db = hip_d.forward_pair(tgt.to(DEV), refs[0].to(DEV))
        dt0, dr0, p0 = da[0].cpu(), db[0].cpu(), hip_p(tgt.to(DEV), [r.to(DEV) for r in refs]).cpu()

    def hip(dt, dr, p):
        a, b, c = (t.to(DEV).clone().requires_grad_() for t in (dt, dr, p))
        loss = Losses(ssim=ssim).forward(tgt.to(DEV), [r.to(DEV) for r in refs], [[a], [b]], c, K.to(DEV), None)
        sum(loss).backward()
        return c.grad.cpu()

    def cpu(dt, dr, p, dtype):
        a, b, c = (t.to(dtype).clone().requires_grad_() for t in (dt, dr, p))
        loss = ol.losses_forward(tgt.to(dtype), [r.to(dtype) for r in refs], [[a], [b]], c, K, 0.85 if ssim else 0.0)
        sum(loss).backward()
        return c.grad

    g64 = cpu(dt0, dr0, p0, torch.float64)
    print("ssim %-5s identical inputs:   |HIP-fp64| %.3e   |CPU32-fp64| %.3e" % (ssim, l2(hip(dt0, dr0, p0), g64), l2(cpu(dt0, dr0, p0, torch.float32), g64)))
    for rel in (1e-6, 3e-6, 1e-5):
        for seed in (1, 2, 3):
            dt1, dr1, p1 = perturb_tensor(dt0, rel, seed), perturb_tensor(dr0, rel, seed + 10), perturb_tensor(p0, rel, seed + 20)
            print("ssim %-5s inputs x (1 + %.0e u) seed %d:   |HIP'-fp64| %.3e   |CPU32'-fp64| %.3e   |fp64'-fp64| %.3e" % (
                ssim, rel, seed, l2(hip(dt1, dr1, p1), g64), l2(cpu(dt1, dr1, p1, torch.float32), g64), l2(cpu(dt1, dr1, p1, torch.float64), g64)))
