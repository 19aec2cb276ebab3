#!/usr/bin/env python3
"""Times the fused loss stage alone (forward call = forward + backward of the loss in ONE launch) at bench.py's shapes.
usage: python tools/loss_bench.py [--ssim] [--batch 12 --height 192 --width 640] [--iters 200]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "unsupervised-pseuso-lidar_amd")]
import torch  # noqa: E402
from losses import Losses  # noqa: E402
from oracle.step import synthetic_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ssim", action="store_true")
ap.add_argument("--batch", type=int, default=12)
ap.add_argument("--height", type=int, default=192)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--iters", type=int, default=200)
a = ap.parse_args()
dev = "cuda"
B, H, W = a.batch, a.height, a.width
s = synthetic_batch(B, H, W, seed=3)
tgt, refs, K = s["tgt"].to(dev), [r.to(dev) for r in s["ref_imgs"]], s["intrinsics"].to(dev)
g = torch.Generator().manual_seed(4)
# disparities as a depth network gives them: smooth maps around 0.5 (white-noise disparities scatter the gathers over the whole image and
# take the kernel twice as long: 110 us instead of 54 at 12 x 192 x 640)


def smooth_disp():
    z = torch.randn(B, 1, H // 8 + 2, W // 8 + 2, generator=g)
    z = torch.nn.functional.interpolate(z, size=(H, W), mode="bilinear", align_corners=False)
    return torch.sigmoid(0.3 * z).contiguous().to(dev)


dt, dr = smooth_disp(), smooth_disp()
poses = (0.01 * torch.randn(B, 2, 6, generator=g)).to(dev)
crit = Losses(ssim=a.ssim)
from mcav import nn as N  # noqa: E402

with torch.no_grad():
    for _ in range(20):
        out = crit.forward(tgt, refs, [[dt], [dr]], poses, K, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        out = crit.forward(tgt, refs, [[dt], [dr]], poses, K, None)
    e1.record()
    torch.cuda.synchronize()
    call_us = 1000.0 * e0.elapsed_time(e1) / a.iters
    # the kernel's own duration: per-dispatch HIP events (csrc/kernel_timer.h), as bench.py's roofline_warp
    N.kernel_timer_begin()
    for _ in range(50):
        out = crit.forward(tgt, refs, [[dt], [dr]], poses, K, None)
    torch.cuda.synchronize()
    durs = sorted(N.kernel_timer_end())
us = 1000.0 * durs[len(durs) // 2]
print("loss stage %s %dx%dx%d: kernel %.1f us (median of %d dispatches, min %.1f) = %.3f of the 8 TB/s HBM roofline at 52 B/pixel; %.1f us per "
      "back-to-back call incl. the host; losses %s" % ("SSIM+L1" if a.ssim else "L1", B, H, W, us, len(durs), 1000.0 * durs[0],
                                                       52.0 * B * H * W / (us * 1e-6) / 8e12, call_us, [round(float(x), 6) for x in out]))
