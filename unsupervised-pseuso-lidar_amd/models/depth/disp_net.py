"""models/depth/disp_net.py -- drop-in for the reference DispNetS (models/depth/disp_net.py:51-141) on MI355X.

Same constructor (alpha, beta), `init_weights()`, parameter names (convN.{0,2,3}, upconvN.0, iconvN.0, predict_dispN.0) and
`__call__(img) -> (disp1, disp2, disp3, disp4)` with values alpha * sigmoid + beta.  Built op by op on mcav/tape.py:
stride-2 conv + ReLU (BatchNorm statistics in the conv epilogue) -> BatchNorm -> conv + ReLU; transposed convs run as the
stride-2 adjoint gather-GEMM computed directly at the cropped size (crop_like); bilinear x2 disparity up-feed; 3-way concats.
"""
import torch
import torch.nn as nn

from mcav import lib as L
from mcav import nn as N
from mcav import tape as T
from mcav.depthnet import spec_of
from mcav.holders import BNParams, ConvParams, DeconvParams

CONV_PLANES = [32, 64, 128, 256, 512, 512, 512]
UP_PLANES = [512, 512, 256, 128, 64, 32, 16]
KS = [7, 5, 3, 3, 3, 3, 3]


def downsample_conv(cin, cout, k):
    """conv(s2) - ReLU - BatchNorm - conv - ReLU with the reference's Sequential indices 0, 2, 3 (1 and 4 are the ReLUs)."""
    seq = nn.Sequential()
    seq.add_module("0", ConvParams(cin, cout, k))
    seq.add_module("2", BNParams(cout))
    seq.add_module("3", ConvParams(cout, cout, k))
    return seq


def _seq0(m):
    s = nn.Sequential()
    s.add_module("0", m)
    return s


def _deconv_spec(holder):
    s = getattr(holder, "_mcav_dspec", None)
    if s is None or s.weight is not holder.weight:
        s = T.DeconvSpec(holder.weight, holder.bias)
        holder._mcav_dspec = s
    return s


class _DispNetSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, record, *params):
        x = L.dev(x.contiguous(), "image")
        tape = T.Tape(enabled=record)              # grad mode is off inside Function.forward: the caller decides
        outs = mod._run(tape, N.nchw_to_nhwc(x, 4))
        ctx.tape, ctx.outs = tape, outs
        return tuple(o.view(o.shape[0], 1, o.shape[1], o.shape[2]) for o in outs)

    @staticmethod
    def backward(ctx, *gs):
        seeds = []
        for o, g in zip(ctx.outs, gs):
            if g is not None:
                g = L.dev(g.contiguous(), "grad")
                seeds.append((o, g.view(g.shape[0], g.shape[2], g.shape[3], 1)))
        ctx.tape.backward(seeds)
        ctx.tape = None
        return (None,) * len(ctx.needs_input_grad)


class DispNetS(nn.Module):
    def __init__(self, alpha=10, beta=0.01):
        super().__init__()
        self.alpha, self.beta = alpha, beta
        cin = 3
        for i in range(7):
            setattr(self, "conv%d" % (i + 1), downsample_conv(cin, CONV_PLANES[i], KS[i]))
            cin = CONV_PLANES[i]
        ins = [CONV_PLANES[6]] + UP_PLANES[:6]
        for i in range(7):
            setattr(self, "upconv%d" % (7 - i), _seq0(DeconvParams(ins[i], UP_PLANES[i], 3)))
        c, u = CONV_PLANES, UP_PLANES
        iin = [u[0] + c[5], u[1] + c[4], u[2] + c[3], u[3] + c[2], 1 + u[4] + c[1], 1 + u[5] + c[0], 1 + u[6]]
        for i in range(7):
            setattr(self, "iconv%d" % (7 - i), _seq0(ConvParams(iin[i], u[i], 3)))
        for i, cch in zip((4, 3, 2, 1), (u[3], u[4], u[5], u[6])):
            setattr(self, "predict_disp%d" % i, _seq0(ConvParams(cch, 1, 3)))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (ConvParams, DeconvParams)):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    # ------------------------------------------------------------------ engine
    def _run(self, tape, x4):
        train = self.training
        o = [None]
        h = x4
        for i in range(1, 8):
            seq = getattr(self, "conv%d" % i)
            k = KS[i - 1]
            first = spec_of(seq[0], 2, (k - 1) // 2, N.PAD_ZERO, smallc=(i == 1))
            a, slab = T.conv(tape, first, h, act=N.ACT_RELU, stats=True, x_needs_grad=(i > 1))
            b = T.batchnorm(tape, seq[1], a, slab, train)
            h = T.conv(tape, spec_of(seq[2], 1, (k - 1) // 2, N.PAD_ZERO), b, act=N.ACT_RELU)
            o.append(h)
        N.flush_bn_counters()
        hw = lambda t: (t.shape[1], t.shape[2])

        def up(name, x, ref):
            return T.deconv(tape, _deconv_spec(getattr(self, name)[0]), x, hw(ref))

        def iconv(name, parts):
            return T.conv(tape, spec_of(getattr(self, name)[0], 1, 1, N.PAD_ZERO), T.concat(tape, parts), act=N.ACT_RELU)

        def predict(name, x):
            s = T.conv(tape, spec_of(getattr(self, name)[0], 1, 1, N.PAD_ZERO), x, act=N.ACT_SIGMOID)
            return T.affine(tape, s, float(self.alpha), float(self.beta))

        def up_disp(d, ref):
            return T.resize_bilinear(tape, d, hw(ref), scale=0.5)      # F.interpolate(scale_factor=2) then crop_like

        i7 = iconv("iconv7", [up("upconv7", o[7], o[6]), o[6]])
        i6 = iconv("iconv6", [up("upconv6", i7, o[5]), o[5]])
        i5 = iconv("iconv5", [up("upconv5", i6, o[4]), o[4]])
        i4 = iconv("iconv4", [up("upconv4", i5, o[3]), o[3]])
        d4 = predict("predict_disp4", i4)
        i3 = iconv("iconv3", [up("upconv3", i4, o[2]), o[2], up_disp(d4, o[2])])
        d3 = predict("predict_disp3", i3)
        i2 = iconv("iconv2", [up("upconv2", i3, o[1]), o[1], up_disp(d3, o[1])])
        d2 = predict("predict_disp2", i2)
        i1 = iconv("iconv1", [up("upconv1", i2, x4), up_disp(d2, x4)])
        d1 = predict("predict_disp1", i1)
        return d1, d2, d3, d4

    def forward(self, x):
        return _DispNetSFn.apply(x, self, torch.is_grad_enabled(), *self.parameters())
