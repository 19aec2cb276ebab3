#!/usr/bin/env python3
"""The depth stem's forward launch alone (24 x 192 x 640 images, BatchNorm statistics of two stacked passes): fp32 MFMA kernel against the split form.
usage: [MCAV_LIB_PATH=...] python tools/stem_bench.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "unsupervised-pseuso-lidar_amd"))
sys.path.insert(0, os.path.join(REPO, "tools"))
import torch  # noqa: E402
from mcav import nn as N  # noqa: E402
from conv_bench import timeit  # noqa: E402

B, H, W = 24, 192, 640
x4 = N.nchw_to_nhwc(torch.randn(B, 3, H, W, device="cuda"), 4)
for name, mma in (("fp32 MFMA", N.MMA_FP32), ("split", N.MMA_SPLIT_ALL)):
    spec = N.ConvSpec(torch.nn.Parameter(torch.randn(64, 3, 7, 7, device="cuda") * 0.1), None, 2, 3, 0, smallc=True)
    spec.mma = mma
    ms = timeit(lambda: N.conv_fwd(spec, x4, stats=True, groups=2))
    print("%-10s %.3f ms  %.1f TF/s" % (name, ms, 2.0 * B * 96 * 320 * 64 * 147 / ms / 1e9))
