"""Loss-stage pieces that are not the fused single-scale kernel: standalone smoothness, multi-scale nets.

smooth_loss: Losses.smooth_loss (reference losses.py:242-260) for a list of depth scales, weight /2.3 per scale.
multiscale_losses: Losses.forward for depth nets that return several scales (DispNetS): every coarser depth is
bilinearly resized to full resolution before warping (losses.py:212-216); smoothness stays at native resolution.
"""
import torch

from . import lib as L


class _SmoothFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *maps):
        h = L.lib()
        dev = maps[0].device
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        weight = 1.0
        ctx.maps = maps
        for D in maps:
            L.dev(D, "depth")
            B, C, H, W = D.shape
            if C != 1:
                raise L.MCAVError("smooth_loss expects [B,1,H,W] maps")
            ws = L.workspace(h.mcav_smooth_workspace_bytes(B, H, W), dev, "smooth")
            scratch = torch.empty_like(D)
            L.check(h.mcav_smooth_loss_fwd_bwd(L.ptr(D), B, H, W, weight, None, L.ptr(loss), L.ptr(scratch), 0, L.ptr(ws), ws.numel(),
                                               L.stream()), "mcav_smooth_loss_fwd_bwd")
            weight /= 2.3
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        h = L.lib()
        dev = g.device
        up = g.reshape(1).to(torch.float32).contiguous()
        dummy = torch.zeros(1, dtype=torch.float32, device=dev)
        grads = []
        weight = 1.0
        for D in ctx.maps:
            B, _, H, W = D.shape
            ws = L.workspace(h.mcav_smooth_workspace_bytes(B, H, W), dev, "smooth")
            gD = torch.empty_like(D)
            L.check(h.mcav_smooth_loss_fwd_bwd(L.ptr(D), B, H, W, weight, L.ptr(up), L.ptr(dummy), L.ptr(gD), 0, L.ptr(ws), ws.numel(),
                                               L.stream()), "mcav_smooth_loss_fwd_bwd")
            grads.append(gD)
            weight /= 2.3
        return tuple(grads)


def smooth_loss(pred_map):
    if not isinstance(pred_map, (tuple, list)):
        pred_map = [pred_map]
    return _SmoothFn.apply(*[m.contiguous() for m in pred_map])


class _ResizeFn(torch.autograd.Function):
    """F.interpolate(D, [H, W], mode='bilinear', align_corners=False) on [B,1,h,w] maps (losses.py:214-215)."""

    @staticmethod
    def forward(ctx, D, H, W):
        D = L.dev(D.contiguous(), "depth")
        B, C, h, w = D.shape
        out = torch.empty((B, 1, H, W), dtype=torch.float32, device=D.device)
        L.check(L.lib().mcav_resize_bilinear_fwd(L.ptr(D), B, h, w, L.ptr(out), H, W, 0.0, 0.0, L.stream()), "mcav_resize_bilinear_fwd")
        ctx.dims = (B, h, w, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        B, h, w, H, W = ctx.dims
        g = L.dev(g.contiguous(), "grad")
        out = torch.empty((B, 1, h, w), dtype=torch.float32, device=g.device)
        L.check(L.lib().mcav_resize_bilinear_bwd(L.ptr(g), B, h, w, L.ptr(out), H, W, 0.0, 0.0, 0, L.stream()), "mcav_resize_bilinear_bwd")
        return out, None, None


def multiscale_losses(tgt, refs, disparity, poses, K, inputs_are_depth=False, ssim=False):
    """Losses.forward for depth nets that return several scales (DispNetS).  Per scale: depth (from disparity), bilinear resize
    to the image size, the fused 3-warp kernel (L1, or the 0.85 SSIM + 0.15 L1 mix when ssim: the reference composes its photometric
    term per scale, losses.py:209-221) on the resized depths; smoothness on the native-resolution depths of tgt."""
    import losses as LS                      # the fused kernel's autograd node
    from geometry.pose_geometry import disp_to_depth
    from mcav import tape  # noqa: F401  (registers the resize entry points)
    depths = disparity if inputs_are_depth else disp_to_depth(disparity)
    n = len(depths[0])
    H, W = tgt.shape[-2:]
    tw = (0.5 / (2 * n), 0.5 / (2 * n), 1.0 / (2 * n))      # mean of the two tgt-view terms; every term / (2 n)  (losses.py:227-240)
    total = None
    for s in range(n):
        Dt, Dr = depths[0][s], depths[1][s]
        if Dt.shape[-1] != W:
            Dt, Dr = _ResizeFn.apply(Dt, H, W), _ResizeFn.apply(Dr, H, W)
        l0, _ = LS._WarpLossFn.apply(Dt.contiguous(), Dr.contiguous(), poses.contiguous(), tgt.contiguous(), refs[0].contiguous(),
                                     refs[1].contiguous(), K.contiguous(), L.WL_INPUT_DEPTH | L.WL_NO_SMOOTH | (L.WL_SSIM if ssim else 0), tw)
        total = l0 if total is None else total + l0
    return [total, smooth_loss(depths[0])]
