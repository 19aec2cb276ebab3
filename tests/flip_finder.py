"""Names the pixels at which an fp32 evaluation of the loss takes another branch than float64 does (test infrastructure).

The photometric loss (reference losses.py:183-240; the SSIM mix of losses.py:12-54,77) is only piecewise smooth: the bilinear cell
(floor of the sampling position, pose_geometry.py:227), the L1 sign and the SSIM clamp are decisions.  A pixel whose float64 value sits
within rounding of such a kink is decided by the rounding of whichever fp32 evaluation looks at it, and d loss / d poses then moves by
that pixel's whole contribution (1e-4 .. 1e-3 of the gradient's norm at 2 x 64 x 128).  This module makes that statement checkable:

  oracle_taps()     per warp and pixel, from the oracle's own code path in any dtype: sampling position (ix, iy), d loss / d (ix, iy),
                    residuals, the unclamped SSIM distance -- and d loss / d poses
  hip_taps() / host_taps()   the same per-pixel quantities from the HIP kernels (mcav_warp_loss_debug_taps) / from csrc/warp_math.h compiled
                    for the host (tests/hostcheck)
  find_flips()      every pixel whose d loss / d (ix, iy) differs from float64's by more than rounding, with the decision that differs and its
                    float64 MARGIN (distance of the sampling position from the cell edge in pixels; |residual|; distance of the SSIM
                    distance from its clamp) -- a pixel without such a decision is reported as 'unexplained'
  pose_gradient_with() float64's d loss / d poses with the named pixels' d loss / d (ix, iy) replaced by the fp32 evaluation's: what float64
                    gives when it takes the SAME side at exactly those pixels.  The fp32 pose gradient must equal it to rounding.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

from oracle import geometry as og
from oracle.losses import ssim_distance

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TW = (0.25, 0.25, 0.5)          # weight of each warp's mean in loss_mam (losses.py:227-240)
NPLANES = 7


def _plan(tgt, refs, Dt, Dr, poses):
    return [(refs[0], tgt, Dt, poses[:, 0], False), (refs[1], tgt, Dt, poses[:, 1], False), (tgt, refs[1], Dr, poses[:, 0], True)]


def oracle_taps(tgt, refs, disp_t, disp_r, poses, K, dtype=torch.float64, ssim_weight=0.0):
    """-> dict(ix, iy, gix, giy [3][B,H,W]; res [3][B,3,H,W]; v (unclamped SSIM distance) [3][B,3,H,W] or None; grids, poses (graph kept),
    dposes [B,2,6])."""
    B, _, H, W = tgt.shape
    tgt, refs = tgt.to(dtype), [r.to(dtype) for r in refs]
    p = poses.detach().to(dtype).clone().requires_grad_()
    Dt, Dr = 1 / (10 * disp_t.detach().to(dtype) + 0.01), 1 / (10 * disp_r.detach().to(dtype) + 0.01)
    out = dict(ix=[], iy=[], gix=[], giy=[], res=[], v=[], grids=[], poses=p)
    total = 0
    for w, (src, tar, D, pose, inv) in enumerate(_plan(tgt, refs, Dt, Dr, p)):
        grid = og.project(og.reconstruct(D[:, 0], K), K, og.pose_to_matrix(pose, invert=inv))
        grid.retain_grad()
        warped = F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        res = warped - tar
        if ssim_weight:
            xp, yp = F.pad(warped, (1, 1, 1, 1), mode="reflect"), F.pad(tar, (1, 1, 1, 1), mode="reflect")
            mx, my = F.avg_pool2d(xp, 3, 1), F.avg_pool2d(yp, 3, 1)
            sx, sy, sxy = F.avg_pool2d(xp * xp, 3, 1) - mx * mx, F.avg_pool2d(yp * yp, 3, 1) - my * my, F.avg_pool2d(xp * yp, 3, 1) - mx * my
            v = (1 - ((2 * mx * my + 1e-4) * (2 * sxy + 9e-4)) / ((mx * mx + my * my + 1e-4) * (sx + sy + 9e-4))) / 2
            out["v"].append(v.detach())
            term = (ssim_weight * ssim_distance(warped, tar) + (1 - ssim_weight) * res.abs()).mean()
        else:
            out["v"].append(None)
            term = res.abs().mean()
        total = total + TW[w] * term
        out["grids"].append(grid)
        out["res"].append(res.detach())
    total.backward(retain_graph=True)
    for grid in out["grids"]:
        g = grid.detach()
        out["ix"].append((g[..., 0] + 1) / 2 * (W - 1))
        out["iy"].append((g[..., 1] + 1) / 2 * (H - 1))
        out["gix"].append(grid.grad[..., 0] * 2 / (W - 1))           # ix = (gx + 1) / 2 * (W - 1)
        out["giy"].append(grid.grad[..., 1] * 2 / (H - 1))
    out["dposes"] = p.grad.detach().clone()
    return out


def hip_taps(tgt, refs, disp_t, disp_r, poses, K, ssim=False):
    """The HIP kernels' per-pixel dump (C ABI: mcav_warp_loss_debug_taps).  -> (taps [B,3,7,H,W] on the CPU, d_poses [B,2,6], losses [2])."""
    from mcav import lib as L
    B, _, H, W = tgt.shape
    dev = tgt.device
    h = L.lib()
    ws = L.workspace(h.mcav_warp_loss_workspace_bytes(B, H, W), dev, "warp_loss", zero=True)
    taps = torch.zeros(B, 3, NPLANES, H, W, dtype=torch.float32, device=dev)
    losses = torch.empty(2, dtype=torch.float32, device=dev)
    g_dt, g_dr, g_p = torch.empty_like(disp_t), torch.empty_like(disp_r), torch.empty_like(poses)
    flags = (L.WL_K_F64 if K.dtype == torch.float64 else 0) | (L.WL_SSIM if ssim else 0)
    args = [t.contiguous() for t in (tgt, refs[0], refs[1], disp_t, disp_r, poses, K)]
    L.check(h.mcav_warp_loss_debug_taps(*[L.ptr(t) for t in args], B, H, W, flags, None, L.ptr(losses), L.ptr(g_dt), L.ptr(g_dr), L.ptr(g_p),
                                        L.ptr(ws), ws.numel(), L.ptr(taps), taps.numel(), L.stream()), "mcav_warp_loss_debug_taps")
    torch.cuda.synchronize()
    return taps.cpu(), g_p.cpu(), losses.cpu()


_HOST = None


def host_lib():
    global _HOST
    if _HOST is None:
        out = os.path.join(REPO, "tests", "hostcheck", "_build")
        os.makedirs(out, exist_ok=True)
        so = os.path.join(out, "libhostcheck.so")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-I", os.path.join(REPO, "unsupervised-pseuso-lidar_amd", "csrc"),
                               os.path.join(REPO, "tests", "hostcheck", "hostcheck.cpp"), "-o", so])
        _HOST = ctypes.CDLL(so)
    return _HOST


def host_taps(tgt, refs, disp_t, disp_r, poses, K):
    """csrc/warp_math.h compiled for the host (L1 photometric): -> (taps [B,3,7,H,W], d_poses [B,2,6])."""
    h = host_lib()
    B, _, H, W = tgt.shape
    f32 = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
    a = [f32(t) for t in (tgt, refs[0], refs[1], disp_t, disp_r, poses)]
    Kd = np.ascontiguousarray(K.detach().cpu().numpy(), dtype=np.float64)
    fp = lambda v: v.ctypes.data_as(ctypes.c_void_p)
    taps = np.zeros((B, 3, NPLANES, H, W), np.float32)
    assert h.hostcheck_warp_taps(*[fp(v) for v in a], fp(Kd), B, H, W, fp(taps)) == 0
    up, losses = np.ones(2, np.float32), np.zeros(2, np.float32)
    gdt, gdr, gp = np.zeros_like(a[3]), np.zeros_like(a[4]), np.zeros_like(a[5])
    assert h.hostcheck_warp_loss(*[fp(v) for v in a], fp(Kd), B, H, W, fp(up), fp(losses), fp(gdt), fp(gdr), fp(gp)) == 0
    return torch.from_numpy(taps), torch.from_numpy(gp)


def find_flips(taps, o64, rel=1e-3):
    """taps [B,3,7,H,W] of an fp32 evaluation, o64 = oracle_taps(..., float64).  -> list of dicts, one per pixel whose d loss / d (ix, iy)
    is further than rel * max|d loss / d (ix, iy)| from float64's."""
    flips = []
    for w in range(3):
        gix, giy = taps[:, w, 2].double(), taps[:, w, 3].double()
        diff = (gix - o64["gix"][w]).abs() + (giy - o64["giy"][w]).abs()
        scale = float((o64["gix"][w].abs() + o64["giy"][w].abs()).max())
        for b, y, x in (diff > rel * scale).nonzero().tolist():
            ix64, iy64 = float(o64["ix"][w][b, y, x]), float(o64["iy"][w][b, y, x])
            ixf, iyf = float(taps[b, w, 0, y, x]), float(taps[b, w, 1, y, x])
            f = dict(b=b, warp=w, y=y, x=x, ix64=ix64, iy64=iy64, ix=ixf, iy=iyf, diff=float(diff[b, y, x]) / scale, kind="unexplained", margin=float("inf"))
            cands = []
            if np.floor(ixf) != np.floor(ix64):
                cands.append(("cell-x", abs(ix64 - round(ix64))))
            if np.floor(iyf) != np.floor(iy64):
                cands.append(("cell-y", abs(iy64 - round(iy64))))
            r64 = o64["res"][w][b, :, y, x]
            cands.append(("l1-sign", float(r64.abs().min())))                      # the residual closest to its kink
            if o64["v"][w] is not None:                                            # the clamp of every 3x3 window the pixel takes part in
                v = o64["v"][w][b, :, max(0, y - 1):y + 2, max(0, x - 1):x + 2]
                cands.append(("ssim-clamp", float(torch.minimum(v.abs(), (v - 1).abs()).min())))
            kind, margin = min(cands, key=lambda c: c[1])
            f["kind"], f["margin"] = kind, margin
            flips.append(f)
    return flips


def pose_gradient_with(o64, taps, flips):
    """float64's d loss / d poses when the pixels in `flips` take the fp32 evaluation's d loss / d (ix, iy) (o64 keeps its graph)."""
    B, H, W = o64["ix"][0].shape
    grads = []
    for w, grid in enumerate(o64["grids"]):
        g = grid.grad.detach().clone()
        for f in flips:
            if f["warp"] == w:
                g[f["b"], f["y"], f["x"], 0] = float(taps[f["b"], w, 2, f["y"], f["x"]]) * (W - 1) / 2
                g[f["b"], f["y"], f["x"], 1] = float(taps[f["b"], w, 3, f["y"], f["x"]]) * (H - 1) / 2
        grads.append(g)
    (dp,) = torch.autograd.grad(o64["grids"], o64["poses"], grad_outputs=grads, retain_graph=True)
    return dp


def l2(a, b):
    a, b = torch.as_tensor(a).double().reshape(-1), torch.as_tensor(b).double().reshape(-1)
    return float((a - b).norm() / b.norm())


def report(name, taps, dposes, o64, cell_margin=5e-5, value_margin=1e-5):
    """Prints the named pixels and returns (flips, gap, gap_after): |dposes - float64| and |dposes - float64 taking the named pixels' side|."""
    flips = find_flips(taps, o64)
    gap = l2(dposes, o64["dposes"])
    after = l2(dposes, pose_gradient_with(o64, taps, flips)) if flips else gap
    print("%s: |d poses - fp64| %.3e; %d pixel(s) decided differently; with float64 taking their side: %.3e" % (name, gap, len(flips), after))
    for f in flips:
        print("   sample %d warp %d pixel (y %d, x %d): %s, float64 margin %.2e   [ix %.6f / fp64 %.6f, iy %.6f / fp64 %.6f, |d g| %.2f of max]"
              % (f["b"], f["warp"], f["y"], f["x"], f["kind"], f["margin"], f["ix"], f["ix64"], f["iy"], f["iy64"], f["diff"]))
    bad = [f for f in flips if f["margin"] > (cell_margin if f["kind"].startswith("cell") else value_margin)]
    return flips, gap, after, bad


def flip_correction(hip_inputs, taps, K, ssim_weight, cell_margin=5e-5, value_margin=1e-5):
    """For the whole-step tests.  hip_inputs = (tgt, refs, disp_t, disp_r, poses) as the HIP path produced them (CPU tensors).  -> (delta, flips,
    bad): delta [B,2,6] = float64's d loss / d poses with the named tie pixels taking the HIP kernel's side MINUS float64's own, on identical
    inputs -- what those pixels (and nothing else) add to every gradient downstream of the poses."""
    tgt, refs, dt, dr, p = hip_inputs
    o64 = oracle_taps(tgt, refs, dt, dr, p, K, torch.float64, ssim_weight)
    flips = find_flips(taps, o64)
    bad = [f for f in flips if f["margin"] > (cell_margin if f["kind"].startswith("cell") else value_margin)]
    if not flips:
        return torch.zeros_like(o64["dposes"]), flips, bad
    return pose_gradient_with(o64, taps, flips) - o64["dposes"], flips, bad
