"""CPU: the device math header (csrc/warp_math.h), compiled for the host, against the reference golden vectors.

This exercises the exact per-pixel functions the HIP kernels call (projection, bilinear taps and their
derivatives, smoothness stencil, Rodrigues backward) without a GPU.  It is a test of the header, not a
product path: the library built here lives under tests/ and is never loaded by the package.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import PKG, REPO

SRC = os.path.join(REPO, "tests", "hostcheck", "hostcheck.cpp")
OUT = os.path.join(REPO, "tests", "hostcheck", "_build")


@pytest.fixture(scope="module")
def host():
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, "libhostcheck.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-I", os.path.join(PKG, "csrc"), SRC, "-o", so])
    return ctypes.CDLL(so)


def fptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("up,tag", [((1.0, 1.0), ""), ((1.0, 0.0), "_mam"), ((0.0, 1.0), "_smooth")])
def test_warp_math_vs_reference(host, golden, up, tag):
    g = golden("loss_small.npz")
    B, _, H, W = g["tgt"].shape
    f32 = lambda k: np.ascontiguousarray(g[k], dtype=np.float32)
    tgt, r0, r1, dt, dr, poses = (f32(k) for k in ("tgt", "ref0", "ref1", "disp_t", "disp_r", "poses"))
    K = np.ascontiguousarray(g["K"], dtype=np.float64)
    upstream = np.array(up, np.float32)
    losses = np.zeros(2, np.float32)
    gdt, gdr, gp = np.zeros_like(dt), np.zeros_like(dr), np.zeros_like(poses)
    rc = host.hostcheck_warp_loss(fptr(tgt), fptr(r0), fptr(r1), fptr(dt), fptr(dr), fptr(poses), fptr(K), B, H, W, fptr(upstream),
                                  fptr(losses), fptr(gdt), fptr(gdr), fptr(gp))
    assert rc == 0
    assert np.allclose(losses, g["loss"], rtol=2e-6)
    want_dt = g["g_disp_t" + tag]
    want_p = g["g_poses" + tag]
    # L1 sign flips at |residual| ~ 1e-7 make a handful of pixels differ; everything else agrees to fp32 rounding
    bad = np.abs(gdt - want_dt) > 1e-4 * np.abs(want_dt).max()
    assert bad.mean() < 1e-3
    assert np.linalg.norm(gdt - want_dt) / np.linalg.norm(want_dt) < 1e-3
    assert np.abs(gp - want_p).max() <= 1e-3 * np.abs(want_p).max() + 1e-12
    if tag == "":
        assert np.linalg.norm(gdr - g["g_disp_r"]) / np.linalg.norm(g["g_disp_r"]) < 1e-3
