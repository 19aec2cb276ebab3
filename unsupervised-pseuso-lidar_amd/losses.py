"""losses.py -- drop-in for the reference's losses.py on MI355X.

Same call surface (reference losses.py:12-54, 56-271): ``Losses().forward(tgt, ref_imgs, disparity, poses,
intrinsics, gt) -> [loss_mam, loss_smooth]`` with ``sum(loss).backward()`` working, and ``SSIM().standard_loss``.
The arithmetic runs in ONE fused HIP kernel (csrc/warp_loss.hip, mcav_warp_loss_fwd_bwd): disp->depth, the
three inverse warps per triplet (incl. the tgt->refs[1] warp with depth(ref0) and the inverted pose[0],
reference losses.py:203-207), bilinear sampling, L1 means, second-order smoothness, and the analytic backward
to both disparity maps and to the poses.  The kernel evaluates forward and backward together assuming unit
upstream gradients (what ``sum(loss).backward()`` supplies); a different upstream re-runs it with the real
weights, decided on the device without a host sync.
"""
import ctypes

import torch

from mcav import lib as L


class _WarpLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp_t, disp_r, poses, tgt, ref0, ref1, K, flags, term_weights):
        B, _, H, W = tgt.shape
        for n, t in (("tgt", tgt), ("ref0", ref0), ("ref1", ref1), ("disp_t", disp_t), ("disp_r", disp_r), ("poses", poses)):
            L.dev(t, n)
        if K.dtype == torch.float64:
            flags |= L.WL_K_F64
        L.dev(K, "intrinsics", K.dtype)
        h = L.lib()
        ws = L.workspace(h.mcav_warp_loss_workspace_bytes(B, H, W), tgt.device, "warp_loss", zero=True)
        losses = torch.empty(2, dtype=torch.float32, device=tgt.device)
        g_dt = torch.empty_like(disp_t)
        g_dr = torch.empty_like(disp_r)
        g_p = torch.empty_like(poses)
        tw = (ctypes.c_float * 3)(*term_weights)
        args = [L.ptr(tgt), L.ptr(ref0), L.ptr(ref1), L.ptr(disp_t), L.ptr(disp_r), L.ptr(poses), L.ptr(K), B, H, W]
        tail = [tw, L.ptr(losses), L.ptr(g_dt), L.ptr(g_dr), L.ptr(g_p), L.ptr(ws), ws.numel(), L.stream()]
        from mcav import nn as N
        i0 = h.mcav_kernel_timer_count() if N.PROFILE_LOSS is not None else 0
        L.check(h.mcav_warp_loss_fwd_bwd(*args, flags, None, *tail), "mcav_warp_loss_fwd_bwd")
        if N.PROFILE_LOSS is not None:
            N.PROFILE_LOSS.append(("warp_loss", i0, h.mcav_kernel_timer_count()))       # ONE launch since round 3 (prepare / finalize folded in)
        ctx.rerun = (args, tail, flags, (tgt, ref0, ref1, disp_t, disp_r, poses, K, ws, losses))
        ctx.grads = (g_dt, g_dr, g_p)
        l0, l1 = losses.unbind(0)
        return l0, l1

    @staticmethod
    def backward(ctx, g0, g1):
        args, tail, flags, keep = ctx.rerun
        g_dt, g_dr, g_p = ctx.grads
        dev = g_dt.device
        if g0 is not None and g1 is not None:          # the usual case (sum(loss).backward()): ONE launch instead of a fill and two copies
            up = torch.stack([g0.reshape(()), g1.reshape(())]).to(dtype=torch.float32, device=dev)
        else:
            up = torch.zeros(2, dtype=torch.float32, device=dev)
            if g0 is not None:
                up[0:1].copy_(g0.reshape(1))
            if g1 is not None:
                up[1:2].copy_(g1.reshape(1))
        scratch = torch.empty(2, dtype=torch.float32, device=dev)
        tail = list(tail)
        tail[1] = L.ptr(scratch)        # loss values are not needed again
        # no-op on the device when upstream == (1, 1); otherwise recomputes the gradients with the real weights
        L.check(L.lib().mcav_warp_loss_fwd_bwd(*args, flags | L.WL_SKIP_IF_UNIT, L.ptr(up), *tail), "mcav_warp_loss_fwd_bwd(bwd)")
        return g_dt, g_dr, g_p, None, None, None, None, None, None


class SSIM:
    """SSIM.standard_loss (reference losses.py:12-54) as a HIP kernel (3x3 box over reflection padding)."""

    def standard_loss(self, x, y, C1=1e-4, C2=9e-4, kernel_size=3, stride=1):
        if kernel_size != 3 or stride != 1:
            raise L.MCAVError("SSIM: only kernel_size=3, stride=1 (the reference's defaults) are implemented")
        x = L.dev(x.contiguous(), "x")
        y = L.dev(y.contiguous(), "y")
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        L.check(L.lib().mcav_ssim_fwd(L.ptr(x), L.ptr(y), B * C, H, W, C1, C2, L.ptr(out), L.stream()), "mcav_ssim_fwd")
        return out


class Losses:
    """`Losses()` is the reference's live loss (L1 photometric, losses.py:183-240).  `Losses(ssim=True)` -- or setting `.ssim` on an
    instance, which is what `loss: {ssim: true}` in a trainer config does -- switches the photometric term of every warp to
    0.85 * SSIM.standard_loss(warped, target) + 0.15 * |target - warped| (the mix of the reference's dormant
    compute_photometric_loss, losses.py:66-77, without its mean + 0.5 std clip), evaluated by the same fused kernel (MCAV_WL_SSIM)."""

    def __init__(self, ssim=False):
        self.clip_loss = 0.5
        self.ssim = bool(ssim)

    def _flags(self, n_scales):
        return L.WL_SSIM if self.ssim else 0

    def forward(self, tgt_img, ref_imgs, disparity, poses, intrinsics, gt=None):
        """-> [loss_mam, loss_smooth].  disparity = [disps(tgt), disps(ref0)], each a list over scales."""
        disp_t, disp_r = disparity[0], disparity[1]
        n = len(disp_t)
        ssim_flag = self._flags(max(n, len(disp_r)))
        if n != 1 or len(disp_r) != 1:
            from mcav.multiscale import multiscale_losses
            return multiscale_losses(tgt_img, ref_imgs, disparity, poses, intrinsics, ssim=self.ssim)
        tw = (0.25, 0.25, 0.5)     # mean of the two tgt-view L1 terms and the third term, averaged (losses.py:227-240)
        l0, l1 = _WarpLossFn.apply(disp_t[0].contiguous(), disp_r[0].contiguous(), poses.contiguous(), tgt_img.contiguous(),
                                   ref_imgs[0].contiguous(), ref_imgs[1].contiguous(), intrinsics.contiguous(), ssim_flag, tw)
        return [l0, l1]

    def reprojection_loss(self, tgt, refs, depths, poses, intrinsics, mode='min'):
        """Reference signature (losses.py:183): takes DEPTHS (nested [time][scale])."""
        if mode != 'min':
            raise L.MCAVError("reprojection_loss: only mode='min' (the reference's live path) is implemented")
        ssim_flag = self._flags(len(depths[0]))
        if len(depths[0]) != 1:
            from mcav.multiscale import multiscale_losses
            return multiscale_losses(tgt, refs, depths, poses, intrinsics, inputs_are_depth=True, ssim=self.ssim)[0]
        tw = (0.25, 0.25, 0.5)
        l0, _ = _WarpLossFn.apply(depths[0][0].contiguous(), depths[1][0].contiguous(), poses.contiguous(), tgt.contiguous(),
                                  refs[0].contiguous(), refs[1].contiguous(), intrinsics.contiguous(),
                                  L.WL_INPUT_DEPTH | L.WL_NO_SMOOTH | ssim_flag, tw)
        return l0

    def smooth_loss(self, pred_map):
        from mcav.multiscale import smooth_loss
        return smooth_loss(pred_map)
