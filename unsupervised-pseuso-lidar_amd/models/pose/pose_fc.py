"""models/pose/pose_fc.py -- drop-in for the reference PoseFc (models/pose/pose_fc.py:21-84) on MI355X.

PoseNet's trunk, then the reference's 360 -> 128 -> 32 -> 12 regressor (only valid at 384x1280 inputs, where the trunk's
output is [B,12,3,10]) and the in-place rotation zeroing `pose[:, :, :3] = 0`.  The fully connected layers run as 1x1
gather-GEMMs on [B,1,1,C] tensors; the flatten follows the reference's NCHW order.
"""
import torch
import torch.nn as nn

from mcav import lib as L
from mcav import nn as N
from mcav import posenet as E
from mcav import tape as T
from mcav.depthnet import spec_of
from mcav.holders import ConvParams, LinearParams
from .pose_net import conv_gn


def _fc_spec(lin):
    """A LinearParams seen as a 1x1 conv: weight [out, in] -> a detached [out, in, 1, 1] view of the same storage whose
    .grad is a view of the parameter's gradient buffer, so wgrad accumulates straight into lin.weight.grad."""
    s = getattr(lin, "_mcav_spec", None)
    if s is None or s._lin_weight is not lin.weight or s.weight.data_ptr() != lin.weight.data_ptr():
        w4 = lin.weight.detach().view(lin.weight.shape[0], lin.weight.shape[1], 1, 1)
        s = N.ConvSpec(w4, lin.bias, 1, 0, N.PAD_ZERO)
        s._lin_weight = lin.weight
        lin._mcav_spec = s
    g = N.grad_buffer(lin.weight)
    if s.weight.grad is None or s.weight.grad.data_ptr() != g.data_ptr():
        s.weight.grad = g.view_as(s.weight)
    return s


class _PoseFcFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, record, tgt, ref0, ref1, *params):
        imgs = [L.dev(t.contiguous(), "image") for t in (tgt, ref0, ref1)]
        acts, p = E.trunk_forward(mod, E.pack_inputs(imgs[0], imgs[1:]))
        B, h, w, c = p.shape
        if c * h * w != 12 * 3 * 10:
            raise L.MCAVError("PoseFc is hard-wired to 384x1280 inputs (12*3*10 features), got %dx%dx%d" % (c, h, w))
        tape = T.Tape(enabled=record)
        flat = N.nhwc_to_nchw(p).view(B, 1, 1, 360)                       # the reference flattens NCHW
        fc = mod.fc_loc
        h1 = T.conv(tape, _fc_spec(getattr(fc, "0")), flat, act=N.ACT_RELU)
        h2 = T.conv(tape, _fc_spec(getattr(fc, "2")), h1, act=N.ACT_RELU)
        out = T.conv(tape, _fc_spec(getattr(fc, "4")), h2)
        mask = torch.ones(B, 1, 1, 12, device=out.device)
        mask.view(B, 2, 6)[:, :, :3] = 0
        out = T.mul_const(tape, out, mask)
        ctx.mod, ctx.acts, ctx.tape, ctx.flat, ctx.out, ctx.pshape = mod, acts, tape, flat, out, (B, h, w, c)
        return out.view(B, 2, 6)

    @staticmethod
    def backward(ctx, g):
        g = L.dev(g.contiguous(), "grad")
        B, h, w, c = ctx.pshape
        ctx.tape.backward([(ctx.out, g.view(B, 1, 1, 12))])
        dflat = ctx.tape.grad(ctx.flat)                                    # [B,1,1,360] in NCHW feature order
        dp = N.nchw_to_nhwc(dflat.view(B, c, h, w), c)
        E.trunk_backward(ctx.mod, ctx.acts, dp)
        ctx.tape = ctx.acts = None
        return (None,) * (5 + len(list(ctx.mod.parameters())))


class PoseFc(nn.Module):
    def __init__(self, nb_ref_imgs=2):
        super().__init__()
        if nb_ref_imgs != 2:
            raise NotImplementedError("PoseFc: two reference images (the reference's configuration) only")
        self.nb_ref_imgs = nb_ref_imgs
        ch = [16, 32, 64, 128, 256, 256, 256]
        cin = 3 * (1 + nb_ref_imgs)
        for i, (c, k) in enumerate(zip(ch, E.KS)):
            setattr(self, "conv%d" % (i + 1), conv_gn(cin, c, k))
            cin = c
        self.pose_pred = ConvParams(cin, 6 * nb_ref_imgs, 1)
        fc = nn.Sequential()
        fc.add_module("0", LinearParams(12 * 3 * 10, 128))
        fc.add_module("2", LinearParams(128, 32))
        fc.add_module("4", LinearParams(32, 12))
        self.fc_loc = fc
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, ConvParams):
                nn.init.xavier_uniform_(m.weight.data)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, LinearParams):
                nn.init.constant_(m.bias, 0)
        getattr(self.fc_loc, "4").weight.data.zero_()          # identity start: the reference zeroes the last layer's weight

    def forward(self, target_image, ref_imgs):
        assert len(ref_imgs) == self.nb_ref_imgs
        return _PoseFcFn.apply(self, torch.is_grad_enabled(), target_image, ref_imgs[0], ref_imgs[1], *self.parameters())
