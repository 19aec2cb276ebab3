#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv kernels on representative layer shapes of the 192x640 R18 step.

usage: python tools/conv_bench.py [fwd|dgrad|wgrad|all] [--tiles 0,1,2,5]
Each line: kind, shape, tile id, median ms over 20 launches, achieved TFLOP/s (algorithmic FLOPs).
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "unsupervised-pseuso-lidar_amd"))
import torch  # noqa: E402
from mcav import nn as N  # noqa: E402

# B, H, W (input), Cin, Cout, k, stride, pad, pad_mode
SHAPES = [
    (12, 48, 160, 64, 64, 3, 1, 1, 0),      # layer1
    (12, 24, 80, 128, 128, 3, 1, 1, 0),     # layer2
    (12, 12, 40, 256, 256, 3, 1, 1, 0),     # layer3
    (12, 6, 20, 512, 512, 3, 1, 1, 0),      # layer4
    (12, 48, 160, 64, 128, 3, 2, 1, 0),     # layer2.0.conv1 (stride 2)
    (12, 96, 320, 32, 16, 3, 1, 1, 1),      # decoder (0,0)
    (12, 192, 640, 16, 16, 3, 1, 1, 1),     # decoder (0,1)
    (12, 96, 320, 96, 32, 3, 1, 1, 1),      # decoder (1,1) (unfused input here)
    (12, 6, 20, 512, 256, 3, 1, 1, 1),      # decoder (4,0)
]


def timeit(fn, n=7, inner=25):
    """Median over n samples of (time of `inner` back-to-back launches) / inner.

    Several launches per sample keep the GPU queue full: a single ~50 us kernel between two events would mostly measure
    the host's enqueue latency (torch.empty + ctypes), not the kernel."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()                                  # something in the queue before the first event
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    tiles = [0]
    for a in sys.argv[2:]:
        if a.startswith("--tiles"):
            tiles = [int(t, 0) for t in a.split("=")[1].split(",")]
    dev = "cuda"
    batch = int(os.environ.get("CONV_BENCH_BATCH", "0"))      # 24 = the stacked tgt/ref0 passes of the step
    only = [int(t) for t in os.environ.get("CONV_BENCH_SHAPES", "").split(",") if t]      # indices into SHAPES (PMC runs: one shape)
    for si, (B, H, W, Cin, Cout, k, s, p, pm) in enumerate(SHAPES):
        if only and si not in only:
            continue
        B = batch or B
        w = torch.nn.Parameter(torch.randn(Cout, Cin, k, k, device=dev) * 0.05)
        b = torch.nn.Parameter(torch.zeros(Cout, device=dev))
        spec = N.ConvSpec(w, b, s, p, pm)
        spec.mma = int(os.environ.get("CONV_BENCH_MMA", "0"))      # 1: bf16 MFMA tiles, 2: fp32 on split operands (where the launch qualifies)
        x = torch.randn(B, H, W, Cin, device=dev)
        Ho, Wo = N.out_size(H, k, s, p), N.out_size(W, k, s, p)
        dy = torch.randn(B, Ho, Wo, Cout, device=dev)
        fl = 2.0 * B * Ho * Wo * Cout * Cin * k * k
        tag = "B%d %dx%d %d->%d k%d s%d" % (B, H, W, Cin, Cout, k, s)
        for t in tiles:
            try:
                if what in ("fwd", "all"):
                    ms = timeit(lambda: N.conv_fwd(spec, x, act=N.ACT_RELU, tile=t))
                    print("fwd   %-32s tile %d  %8.3f ms  %7.2f TF/s" % (tag, t, ms, fl / ms / 1e9))
                if what in ("dgrad", "all"):
                    ms = timeit(lambda: N.conv_dgrad(spec, dy, (H, W), tile=t))
                    print("dgrad %-32s tile %d  %8.3f ms  %7.2f TF/s" % (tag, t, ms, fl / ms / 1e9))
                if what in ("wgrad", "all") and t in (0, 1, 2, 3, 4, 6):
                    ms = timeit(lambda: N.conv_wgrad(spec, x, dy, tile=t))
                    print("wgrad %-32s tile %d  %8.3f ms  %7.2f TF/s" % (tag, t, ms, fl / ms / 1e9))
            except Exception as e:      # an unsupported tile for this shape
                print("      %-32s tile %d  skipped: %s" % (tag, t, e))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
