#!/usr/bin/env python3
"""Per-phase shader cycles of the table-driven conv kernel, averaged over workgroups (needs the MCAV_DIAG=4 build:
make -C unsupervised-pseuso-lidar_amd/csrc diag D=4; MCAV_LIB_PATH=.../libmcav_depth_diag4.so python tools/diag_phases.py)."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "unsupervised-pseuso-lidar_amd"))
import torch  # noqa: E402
from mcav import lib as L  # noqa: E402
from mcav import nn as N  # noqa: E402

SHAPES = [(24, 48, 160, 64, 64, 3, 1, 1, 0), (24, 24, 80, 128, 128, 3, 1, 1, 0), (24, 12, 40, 256, 256, 3, 1, 1, 0), (24, 6, 20, 512, 512, 3, 1, 1, 0)]


def stamps(reset):
    buf = (ctypes.c_ulonglong * 8)()
    h = L.lib()
    h.mcav_diag_stamps.restype = ctypes.c_int
    h.mcav_diag_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    torch.cuda.synchronize()
    assert h.mcav_diag_stamps(buf, int(reset)) == 0
    return list(buf)


def main():
    for (B, H, W, Cin, Cout, k, s, p, pm) in SHAPES:
        w = torch.nn.Parameter(torch.randn(Cout, Cin, k, k, device="cuda") * 0.05)
        spec = N.ConvSpec(w, torch.nn.Parameter(torch.zeros(Cout, device="cuda")), s, p, pm)
        x = torch.randn(B, H, W, Cin, device="cuda")
        dy = torch.randn(B, H, W, Cout, device="cuda")
        for kind, fn in (("fwd", lambda t: N.conv_fwd(spec, x, act=N.ACT_RELU, tile=t)), ("dgrad", lambda t: N.conv_dgrad(spec, dy, (H, W), tile=t))):
            for tile in (1, 10, 12):
                for _ in range(30):
                    fn(tile)
                stamps(True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn(tile)
                e1.record()
                st = stamps(True)
                n = max(1, st[0])
                ms = e0.elapsed_time(e1) / 20
                print("%-5s %dx%d %d->%d tile %2d  %.3f ms  WGs/launch %5d  cycles/WG: tables %6d  first loads %6d  loop %7d  tail %6d  epilogue %6d" %
                      (kind, H, W, Cin, Cout, tile, ms, n // 20, st[1] // n, st[2] // n, st[3] // n, st[4] // n, st[5] // n))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
