"""models/depth/layers.py -- the reference's decoder building blocks (models/depth/layers.py:22-58) as parameter holders.

ConvBlock = reflection-padded 3x3 conv + ELU, Conv3x3 = reflection-padded 3x3 conv.  The modules own the parameters
under the reference's attribute names (`.conv.conv.weight`, `.conv.weight`); standalone calls run the HIP conv kernel.
"""
import torch
import torch.nn as nn

from mcav import lib as L
from mcav import nn as N
from mcav.holders import ConvParams
from mcav.depthnet import dec_spec


def disp_to_depth(disp, min_depth, max_depth):
    """scaled_disp, depth as monodepth2's helper (reference layers.py:10-19; unused on the hot path). Forward only."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    return scaled_disp, 1 / scaled_disp


class Conv3x3(nn.Module):
    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        if not use_refl:
            raise NotImplementedError("only the reflection-padded variant (the reference's only use) is built")
        self.conv = ConvParams(int(in_channels), int(out_channels), 3)

    def forward(self, x):
        """x NCHW -> NCHW (forward only; the differentiable path is DepthDecoder / DispResNet)."""
        return N.nhwc_to_nchw(N.conv_fwd(dec_spec(self.conv), N.nchw_to_nhwc(x, N.up16(x.shape[1]) if x.shape[1] % 4 else x.shape[1])))


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)

    def forward(self, x):
        xin = N.nchw_to_nhwc(x, x.shape[1])
        return N.nhwc_to_nchw(N.conv_fwd(dec_spec(self.conv.conv), xin, act=N.ACT_ELU))


class _Upsample2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = L.dev(x.contiguous(), "x")
        B, C, h, w = x.shape
        out = torch.empty((B, C, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
        L.check(L.lib().mcav_upsample_nearest2x(L.ptr(x), B * C, h, w, L.ptr(out), L.stream()), "mcav_upsample_nearest2x")
        return out

    @staticmethod
    def backward(ctx, g):
        g = L.dev(g.contiguous(), "grad")
        B, C, H, W = g.shape
        out = torch.empty((B, C, H // 2, W // 2), dtype=torch.float32, device=g.device)
        L.check(L.lib().mcav_upsample_nearest2x_bwd(L.ptr(g), B * C, H // 2, W // 2, L.ptr(out), L.stream()), "mcav_upsample_nearest2x_bwd")
        return out


def upsample(x):
    """Upsample input tensor by a factor of 2 (reference layers.py:55-58: F.interpolate(x, scale_factor=2, mode="nearest")), NCHW.
    Inside DepthDecoder the upsample is fused into the next conv's gather (mcav/depthnet.py); this is the standalone op."""
    return _Upsample2xFn.apply(x)
