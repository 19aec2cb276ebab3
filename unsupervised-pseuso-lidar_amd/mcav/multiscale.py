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


def multiscale_losses(tgt, refs, disparity, poses, K, inputs_are_depth=False):
    raise L.MCAVError("multi-scale depth lists are not wired yet (single-scale DispResNet path only)")
