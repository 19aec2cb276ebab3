"""models/depth/resnet_dispnet.py -- drop-in for the reference module (models/depth/resnet_dispnet.py:12-107) on MI355X.

ResnetEncoder (torchvision resnet18/34 trunk restated with torchvision's parameter names), DepthDecoder (monodepth2
decoder, positional ModuleList keys 0..13 as the reference registers them) and DispResNet (no-arg constructor,
`__call__(img[B,3,H,W]) -> [disp[B,1,H,W]]`).  Parameters have the reference's names and shapes; the arithmetic is
the HIP engine in mcav/depthnet.py.  `pretrained=True` cannot fetch ImageNet weights offline: weights are torchvision's
random init (kaiming_normal fan_out) until a state_dict is loaded.
"""
import numpy as np
import torch
import torch.nn as nn

from mcav import lib as L
from mcav import nn as N
from mcav import depthnet as E
from mcav.holders import BNParams, ConvParams, LinearParams
from .layers import ConvBlock, Conv3x3

BLOCK_COUNTS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = ConvParams(inplanes, planes, 3, bias=False)
        self.bn1 = BNParams(planes)
        self.conv2 = ConvParams(planes, planes, 3, bias=False)
        self.bn2 = BNParams(planes)
        self.downsample = downsample
        self.stride = stride


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = ConvParams(inplanes, planes, 1, bias=False)
        self.bn1 = BNParams(planes)
        self.conv2 = ConvParams(planes, planes, 3, bias=False)      # carries the stride (torchvision's v1.5 layout)
        self.bn2 = BNParams(planes)
        self.conv3 = ConvParams(planes, planes * 4, 1, bias=False)
        self.bn3 = BNParams(planes * 4)
        self.downsample = downsample
        self.stride = stride


class ResNet(nn.Module):
    """torchvision.models.resnet{18,34,50,101,152} parameter layout (conv1, bn1, layer1-4, fc)."""

    def __init__(self, num_layers=18):
        super().__init__()
        if num_layers not in BLOCK_COUNTS:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        block = BasicBlock if num_layers <= 34 else Bottleneck
        self.conv1 = ConvParams(3, 64, 7, bias=False)
        self.bn1 = BNParams(64)
        inplanes = 64
        for si, (planes, n) in enumerate(zip((64, 128, 256, 512), BLOCK_COUNTS[num_layers])):
            blocks = []
            for bi in range(n):
                stride = 2 if (si > 0 and bi == 0) else 1
                ds = None
                if stride != 1 or inplanes != planes * block.expansion:
                    ds = nn.Sequential(ConvParams(inplanes, planes * block.expansion, 1, bias=False), BNParams(planes * block.expansion))
                blocks.append(block(inplanes, planes, stride, ds))
                inplanes = planes * block.expansion
            setattr(self, "layer%d" % (si + 1), nn.Sequential(*blocks))
        self.fc = LinearParams(512 * block.expansion, 1000)          # unused by the depth net; kept for state_dict compatibility
        for m in self.modules():
            if isinstance(m, ConvParams):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


def _to_nhwc4(x):
    x = L.dev(x.contiguous(), "image")
    if x.shape[1] != 3:
        raise L.MCAVError("expected a 3-channel image batch [B,3,H,W]")
    return N.nchw_to_nhwc(x, 4)


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, *params):
        feats, sv = E.encoder_forward(mod.encoder, _to_nhwc4(x), mod.training)
        ctx.mod, ctx.sv = mod, sv
        return tuple(N.nhwc_to_nchw(f) for f in feats)

    @staticmethod
    def backward(ctx, *gfeats):
        dfe = [N.nchw_to_nhwc(L.dev(g.contiguous(), "grad"), g.shape[1]) for g in gfeats]
        E.encoder_backward(ctx.mod.encoder, ctx.sv, dfe)
        return (None, None) + (None,) * len(list(ctx.mod.parameters()))


class ResnetEncoder(nn.Module):
    def __init__(self, num_layers=18, pretrained=False):
        super().__init__()
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNet(num_layers)
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4
        if pretrained:
            # The reference starts from torchvision's ImageNet weights (resnet_dispnet.py:30: a network fetch).  Offline they can only come
            # from a local file: MCAV_RESNET_WEIGHTS = a torchvision resnet state_dict (.pth; the key names are the same).  Without it the
            # trunk keeps torchvision's random init, and says so once.
            import os
            import warnings
            path = os.environ.get("MCAV_RESNET_WEIGHTS", "")
            if path and os.path.exists(path):
                missing = self.encoder.load_state_dict(torch.load(path, map_location="cpu"), strict=False)
                if missing.missing_keys:
                    warnings.warn("ResnetEncoder: %s lacks %d keys (e.g. %s)" % (path, len(missing.missing_keys), missing.missing_keys[0]))
            elif not getattr(ResnetEncoder, "_warned", False):
                ResnetEncoder._warned = True
                warnings.warn("ResnetEncoder(pretrained=True): no ImageNet weights offline (set MCAV_RESNET_WEIGHTS to a torchvision resnet%d "
                              "state_dict); the trunk starts from random init, unlike the reference" % num_layers)

    def forward(self, x):
        """img [B,3,H,W] -> 5 feature maps (NCHW), differentiable w.r.t. the parameters."""
        self.features = list(_EncoderFn.apply(x, self, *self.parameters()))
        return self.features


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, nparams, *args):
        feats_nchw = args[nparams:]
        feats = [N.nchw_to_nhwc(L.dev(f.contiguous(), "feature"), f.shape[1]) for f in feats_nchw]
        disps, sv = E.decoder_forward(mod, feats, mod.scales)
        ctx.mod, ctx.sv, ctx.nparams = mod, sv, nparams
        return tuple(disps[s].view(disps[s].shape[0], 1, disps[s].shape[1], disps[s].shape[2]) for s in mod.scales)

    @staticmethod
    def backward(ctx, *gd):
        mod = ctx.mod
        dd = {}
        for s, g in zip(mod.scales, gd):
            if g is not None:
                g = L.dev(g.contiguous(), "grad")
                dd[s] = g.view(g.shape[0], g.shape[2], g.shape[3], 1)
        dfe = E.decoder_backward(mod, ctx.sv, dd)
        return (None, None) + (None,) * ctx.nparams + tuple(N.nhwc_to_nchw(d) for d in dfe)


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        if num_output_channels != 1 or not use_skips:
            raise NotImplementedError("DepthDecoder: the reference's configuration (1 output channel, skips) only")
        self.scales = list(scales)
        self.num_ch_enc = np.array(num_ch_enc)
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.index = {}
        mods = []
        for i in range(4, -1, -1):
            cin = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.index[("upconv", i, 0)] = len(mods)
            mods.append(ConvBlock(cin, self.num_ch_dec[i]))
            cin = self.num_ch_dec[i] + (self.num_ch_enc[i - 1] if i > 0 else 0)
            self.index[("upconv", i, 1)] = len(mods)
            mods.append(ConvBlock(cin, self.num_ch_dec[i]))
        for s in range(4) if set(self.scales) <= set(range(4)) else self.scales:
            self.index[("dispconv", s)] = len(mods)
            mods.append(Conv3x3(self.num_ch_dec[s], 1))
        self.decoder = nn.ModuleList(mods)

    def conv(self, *key):
        return self.decoder[self.index[key]]

    def forward(self, input_features):
        """5 NCHW feature maps -> {("disp", s): [B,1,h,w]} for s in self.scales (differentiable: features and parameters)."""
        params = list(self.parameters())
        outs = _DecoderFn.apply(self, len(params), *params, *input_features)
        self.outputs = {("disp", s): o for s, o in zip(self.scales, outs)}
        return self.outputs


class _DispResNetFn(torch.autograd.Function):
    """Whole depth net as one autograd node: image -> disparity (scale 0)."""

    @staticmethod
    def forward(ctx, x, mod, *params):
        feats, esv = E.encoder_forward(mod.encoder.encoder, _to_nhwc4(x), mod.training)
        disps, dsv = E.decoder_forward(mod.decoder, feats, (0,))
        ctx.mod, ctx.esv, ctx.dsv = mod, esv, dsv
        d = disps[0]
        return d.view(d.shape[0], 1, d.shape[1], d.shape[2])       # NHWC with C = 1 is NCHW

    @staticmethod
    def backward(ctx, g):
        mod = ctx.mod
        g = L.dev(g.contiguous(), "grad")
        dfe = E.decoder_backward(mod.decoder, ctx.dsv, {0: g.view(g.shape[0], g.shape[2], g.shape[3], 1)})
        E.encoder_backward(mod.encoder.encoder, ctx.esv, dfe)
        ctx.esv = ctx.dsv = None
        return (None, None) + (None,) * len(list(mod.parameters()))


class _DispResNetPairFn(torch.autograd.Function):
    """Two independent passes (e.g. tgt and ref0, trainer.py:296-299) as ONE set of launches over the stacked batch.

    BatchNorm statistics stay per pass (the conv epilogue's statistic tiles never straddle a pass; running statistics are
    updated pass 0 first, then pass 1), so the result equals two separate calls, at half the launches and twice the rows per GEMM."""

    @staticmethod
    def forward(ctx, xa, xb, mod, *params):
        xa, xb = L.dev(xa.contiguous(), "image"), L.dev(xb.contiguous(), "image")
        if xa.shape != xb.shape or xa.shape[1] != 3:
            raise L.MCAVError("forward_pair: two [B,3,H,W] image batches of one shape")
        B = xa.shape[0]
        x4 = torch.zeros((2 * B, xa.shape[2], xa.shape[3], 4), dtype=torch.float32, device=xa.device)      # (the kernel keeps channel 3 as it finds it)
        N.nchw_to_nhwc(xa, 4, x4[:B])                  # both passes into ONE stacked NHWC4 buffer (no concatenation pass)
        N.nchw_to_nhwc(xb, 4, x4[B:])
        feats, esv = E.encoder_forward(mod.encoder.encoder, x4, mod.training, groups=2)
        disps, dsv = E.decoder_forward(mod.decoder, feats, (0,))
        ctx.mod, ctx.esv, ctx.dsv = mod, esv, dsv
        d = disps[0]
        d = d.view(d.shape[0], 1, d.shape[1], d.shape[2])
        B = xa.shape[0]
        return d[:B], d[B:]

    @staticmethod
    def backward(ctx, ga, gb):
        mod = ctx.mod
        if ga is None or gb is None:
            ref = ga if ga is not None else gb
            ga = torch.zeros_like(ref) if ga is None else ga
            gb = torch.zeros_like(ref) if gb is None else gb
        g = torch.cat([L.dev(ga.contiguous(), "grad"), L.dev(gb.contiguous(), "grad")], 0)
        dfe = E.decoder_backward(mod.decoder, ctx.dsv, {0: g.view(g.shape[0], g.shape[2], g.shape[3], 1)})
        if N.GRADS_READY is not None:                   # both passes are in this one backward: the decoder's gradients are final
            from mcav.dist import announced_stages
            if "decoder" in announced_stages():
                N.grads_ready(mod.decoder.parameters())
        E.encoder_backward(mod.encoder.encoder, ctx.esv, dfe, complete=True)
        ctx.esv = ctx.dsv = None
        return (None, None, None) + (None,) * len(list(mod.parameters()))


class DispResNet(nn.Module):
    def __init__(self, num_layers=18, dtype=None):
        """The reference hard-codes ResNet-18 (resnet_dispnet.py:101); num_layers=50 is BASELINE.json's configs[3] extension.
        dtype=torch.bfloat16 opts into the bf16 MFMA conv tiles of configs[2] / [4] (mcav.nn.set_compute_dtype); default fp32 as the reference."""
        super().__init__()
        self.encoder = ResnetEncoder(num_layers, True)
        self.decoder = DepthDecoder(self.encoder.num_ch_enc)
        if dtype is not None:
            N.set_compute_dtype(self, dtype)

    def forward(self, x):
        return [_DispResNetFn.apply(x, self, *self.parameters())]

    def forward_pair(self, xa, xb):
        """== (self(xa), self(xb)) evaluated in that order, as one stacked launch set (see _DispResNetPairFn)."""
        da, db = _DispResNetPairFn.apply(xa, xb, self, *self.parameters())
        return [da], [db]


class DispResNet50(DispResNet):
    """No-argument ResNet-50 variant so a config file can name it (model.depth.name: DispResNet50)."""

    def __init__(self):
        super().__init__(50)
