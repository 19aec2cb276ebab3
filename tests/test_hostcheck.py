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


def step_case_loss_inputs():
    """The loss inputs of test_step_gpu.py's 2 x 64 x 128 cases as the CPU oracle's networks produce them (same seeds, same batch)."""
    import torch
    from models.depth.resnet_dispnet import DispResNet
    from models.pose.pose_net import PoseNet
    from oracle import nets as on
    from oracle.step import synthetic_batch
    from seeding import reinit_by_name
    ref_d, ref_p = on.DispResNet(18), on.PoseNet()
    ref_d.load_state_dict(reinit_by_name(DispResNet(18), 141).state_dict())
    ref_p.load_state_dict(reinit_by_name(PoseNet(), 121).state_dict())
    with torch.no_grad():
        ref_p.pose_pred.weight.mul_(0.1)
        ref_p.pose_pred.bias.mul_(0.1)
        ref_d.train()
        ref_p.train()
        s = synthetic_batch(2, 64, 128, seed=5)
        return s, ref_d(s["tgt"])[0], ref_d(s["ref_imgs"][0])[0], ref_p(s["tgt"], s["ref_imgs"])


def test_pose_gradient_gaps_are_named_tie_pixels():
    """d loss / d poses of an fp32 evaluation sits 1e-4 .. 1e-3 from float64 whenever ONE pixel's bilinear cell or L1 sign is decided the
    other way ("name the pixel").  On the step test's inputs the stock-PyTorch fp32 oracle differs from float64 in two
    pixels; tests/flip_finder.py names them, their float64 margins are a few 1e-6 (a cell edge 4.8e-6 px away, a residual of 1e-6), and with
    float64 taking the same side at exactly those pixels the gap closes to rounding.  csrc/warp_math.h compiled for the host (the arithmetic
    of the HIP kernels without fma contraction) decides every pixel as float64 does on these inputs.  The GPU twin of this test
    (test_loss_gpu.py) runs the HIP kernels' own dump."""
    import torch
    import flip_finder as ff
    s, dt, dr, p = step_case_loss_inputs()
    tgt, refs, K = s["tgt"], s["ref_imgs"], s["intrinsics"]
    o64 = ff.oracle_taps(tgt, refs, dt, dr, p, K, torch.float64)
    o32 = ff.oracle_taps(tgt, refs, dt, dr, p, K, torch.float32)
    taps = torch.zeros(2, 3, ff.NPLANES, 64, 128)
    for w in range(3):
        taps[:, w, 0], taps[:, w, 1], taps[:, w, 2], taps[:, w, 3] = o32["ix"][w], o32["iy"][w], o32["gix"][w], o32["giy"][w]
        taps[:, w, 4:7] = o32["res"][w]
    flips, gap, after, bad = ff.report("stock PyTorch fp32", taps, o32["dposes"], o64)
    assert not bad, bad
    assert len(flips) >= 1 and gap > 1e-4 and after < 2e-5           # the whole gap is those pixels
    assert {f["kind"] for f in flips} <= {"cell-x", "cell-y", "l1-sign"}
    ht, hp = ff.host_taps(tgt, refs, dt, dr, p, K)
    flips, gap, after, bad = ff.report("csrc/warp_math.h on the host", ht, hp, o64)
    assert not bad, bad
    assert after < 2e-5
