"""Forward error of the split form against the fp32 MFMA kernel, both against float64, over seeds: is a max-relative difference on one small case a
property of the kernel or of the draw?  python tools/split_seed_scan.py B H W Cin Cout pad_mode [seeds]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "unsupervised-pseuso-lidar_amd"))
from test_bf16_gpu import ref_conv64          # noqa: E402
from test_conv_gpu import nchw, nhwc          # noqa: E402
from test_split_gpu import spec_of            # noqa: E402
from mcav import nn as N                      # noqa: E402

B, H, W, Cin, Cout, pm = [int(a) for a in sys.argv[1:7]]
seeds = int(sys.argv[7]) if len(sys.argv) > 7 else 8
for seed in range(1, seeds + 1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(1.5 * torch.randn(1, Cin, 1, 1, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5 * torch.exp(torch.randn(Cout, 1, 1, 1, generator=g))
    want = ref_conv64(x, w, None, 1, 1, pm)
    dy = torch.randn(want.shape, generator=g) * torch.exp(1.5 * torch.randn(1, Cout, 1, 1, generator=g))
    xr = x.double().requires_grad_()
    ref_conv64(xr, w.double(), None, 1, 1, pm).backward(dy.double())
    out = []
    for mma in (N.MMA_FP32, N.MMA_SPLIT_ALL):
        spec = spec_of(w, None, 1, 1, pm, mma)
        y = nchw(N.conv_fwd(spec, nhwc(x))).double().cpu()
        dx = nchw(N.conv_dgrad(spec, nhwc(dy), (H, W))).double().cpu()
        for got, ref in ((y, want), (dx, xr.grad)):
            e = got - ref
            out.append((float(e.abs().max() / ref.abs().max()), float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())))
    print("seed %d  fwd max fp32 %.2e split %.2e | rms fp32 %.2e split %.2e || dgrad max fp32 %.2e split %.2e | rms fp32 %.2e split %.2e" %
          (seed, out[0][0], out[2][0], out[0][1], out[2][1], out[1][0], out[3][0], out[1][1], out[3][1]))
