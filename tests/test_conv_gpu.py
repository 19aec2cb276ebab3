"""GPU: the implicit-GEMM conv kernels (forward, dgrad, wgrad) and the HBM-bound NN kernels against torch CPU ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def nhwc(t, cp=None):
    """NCHW cpu tensor -> NHWC cuda tensor (optionally zero-padded to cp channels)."""
    t = t.permute(0, 2, 3, 1).contiguous()
    if cp is not None and cp != t.shape[3]:
        t = torch.cat([t, torch.zeros(*t.shape[:3], cp - t.shape[3])], 3)
    return t.contiguous().to(DEV)


def nchw(t, c=None):
    t = t.detach().cpu()
    if c is not None:
        t = t[..., :c]
    return t.permute(0, 3, 1, 2).contiguous()


def ref_conv(x, w, b, stride, pad, pad_mode, act):
    if pad_mode == 1 and pad > 0:
        x = F.pad(x, (pad,) * 4, mode="reflect")
        pad = 0
    y = F.conv2d(x, w, b, stride=stride, padding=pad)
    return {0: lambda v: v, 1: F.relu, 2: F.elu, 3: torch.sigmoid}[act](y)


CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, pad_mode, act, tile
    (2, 12, 20, 16, 16, 3, 1, 1, 1, 2, 0),      # decoder (0,1)-like, N=16 -> MFMA16 tiles
    (2, 12, 20, 32, 32, 3, 1, 1, 1, 2, 0),      # N=32
    (2, 12, 20, 64, 64, 3, 1, 1, 0, 0, 1),      # resnet 3x3, tile 128x64
    (2, 12, 20, 64, 64, 3, 1, 1, 0, 0, 2),      # tile 64x64
    (1, 10, 18, 64, 128, 3, 2, 1, 0, 0, 0),     # stride 2
    (2, 9, 15, 64, 128, 1, 2, 0, 0, 0, 0),      # 1x1 stride-2 downsample, odd sizes
    (2, 12, 20, 128, 128, 3, 1, 1, 0, 1, 5),    # tile 128x128
    (2, 8, 12, 16, 1, 3, 1, 1, 1, 3, 0),        # dispconv: Cout = 1, sigmoid
    (2, 16, 24, 9, 16, 7, 2, 3, 0, 1, 0),       # pose conv1: Cin = 9 padded to 16
    (2, 7, 11, 16, 32, 5, 2, 2, 0, 1, 0),       # pose conv2, odd sizes
    (3, 3, 10, 256, 256, 3, 2, 1, 0, 1, 0),     # pose conv6/7 geometry
    (2, 6, 10, 96, 32, 3, 1, 1, 1, 2, 0),       # Cin = 96 (decoder (1,1))
    (2, 6, 10, 32, 16, 3, 1, 1, 1, 2, 6),       # tile 64x16
    (2, 6, 10, 32, 32, 3, 1, 1, 1, 2, 3),       # tile 256x32
    (2, 6, 10, 32, 32, 3, 1, 1, 1, 2, 7),       # tile 128x32
    (2, 12, 20, 64, 64, 3, 1, 1, 0, 1, 8),      # 32-deep K-tiles, 128x64
    (2, 12, 20, 128, 128, 3, 1, 1, 1, 2, 9),    # 32-deep K-tiles, 128x128, reflection pad
    (1, 10, 18, 64, 128, 3, 2, 1, 0, 0, 10),    # 32-deep K-tiles, 64x64, stride 2 (and its parity-class adjoint)
    (2, 6, 10, 96, 32, 3, 1, 1, 1, 2, 10),      # Cin = 96 with 32-deep K-tiles
    (2, 6, 10, 64, 64, 3, 1, 1, 0, 1, 12),      # 32x64 tiles (16x16x4 MFMA fragments)
    (1, 9, 11, 128, 64, 3, 2, 1, 0, 0, 12),     # 32x64 tiles, stride 2 and its parity-class adjoint
    (2, 6, 10, 64, 64, 3, 1, 1, 1, 2, 12),      # 32x64 tiles, reflection pad (adjoint through the table kernel's border path)
    (2, 6, 10, 64, 64, 3, 1, 1, 0, 1, 11),      # 64-deep K-tiles
    (2, 9, 35, 16, 32, 3, 1, 1, 1, 2, 0),       # halo-tile kernel: 16 -> 32, ragged tiles in both directions
    (1, 17, 66, 16, 16, 3, 1, 1, 0, 1, 0),      # halo-tile kernel with zero padding
    (2, 12, 20, 16, 16, 3, 1, 1, 1, 2, 0x200),  # the same shapes through the general kernels (bit 9)
    (2, 6, 10, 32, 16, 3, 1, 1, 1, 2, 0x200),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(case):
    from mcav import nn as N
    B, H, W, Cin, Cout, k, stride, pad, pad_mode, act, tile = case
    g = torch.Generator().manual_seed(hash(case) % 10000)
    x = torch.randn(B, Cin, H, W, generator=g).requires_grad_()
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).requires_grad_()
    b = (0.1 * torch.randn(Cout, generator=g)).requires_grad_()
    pre = ref_conv(x, w, b, stride, pad, pad_mode, 0)
    want = {0: lambda v: v, 1: F.relu, 2: F.elu, 3: torch.sigmoid}[act](pre)
    dy = torch.randn(pre.shape, generator=g)
    pre.backward(dy)

    wp = torch.nn.Parameter(w.detach().to(DEV))
    bp = torch.nn.Parameter(b.detach().to(DEV))
    spec = N.ConvSpec(wp, bp, stride, pad, pad_mode)
    xin = nhwc(x.detach(), N.up16(Cin) if Cin % 4 else None)
    got = N.conv_fwd(spec, xin, act=act, tile=tile)
    assert rel_err(nchw(got), want) < 2e-5
    # dgrad (gradient at the pre-activation output), wgrad + bias grad
    dyd = nhwc(dy)
    dx = N.conv_dgrad(spec, dyd, (H, W), tile=tile)
    assert rel_err(nchw(dx, Cin), x.grad) < 2e-5
    wt = tile if tile in (1, 2, 3, 4, 6, 7, 0x100) else 0
    N.conv_wgrad(spec, xin, dyd, tile=wt)
    assert rel_err(wp.grad, w.grad) < 5e-5
    assert rel_err(bp.grad, b.grad) < 5e-5
    N.conv_wgrad(spec, xin, dyd, tile=wt)            # accumulates
    assert rel_err(wp.grad, 2 * w.grad) < 5e-5


def test_conv_stats_and_epilogues():
    from mcav import nn as N
    g = torch.Generator().manual_seed(3)
    B, H, W, Cin, Cout = 2, 10, 14, 64, 64
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    spec = N.ConvSpec(torch.nn.Parameter(w.to(DEV)), None, 1, 1, 0)
    y, slab = N.conv_fwd(spec, nhwc(x), stats=True)
    want = F.conv2d(x, w, None, padding=1)
    assert rel_err(nchw(y), want) < 2e-5
    s = slab.sum(0).cpu()
    assert rel_err(s[0], want.sum((0, 2, 3))) < 1e-4
    assert rel_err(s[1], (want ** 2).sum((0, 2, 3))) < 1e-4
    # dgrad epilogue: * relu'(aux) + addend
    dy = torch.randn(B, Cout, H, W, generator=g)
    aux = torch.randn(B, Cin, H, W, generator=g)
    add = torch.randn(B, Cin, H, W, generator=g)
    xr = x.clone().requires_grad_()
    F.conv2d(xr, w, None, padding=1).backward(dy)
    want_dx = xr.grad * (aux > 0).float() + add
    got = N.conv_dgrad(spec, nhwc(dy), (H, W), dact_aux=nhwc(aux), dact=N.ACT_RELU, addend=nhwc(add))
    assert rel_err(nchw(got), want_dx) < 2e-5


@pytest.mark.parametrize("B,H,W,C,k,groups,addend,tile", [(4, 12, 20, 64, 3, 2, True, 0), (2, 9, 14, 64, 3, 1, False, 2), (4, 6, 10, 128, 1, 2, True, 0),
                                                          (6, 24, 40, 64, 3, 2, True, 1)])
def test_dgrad_epilogue_carries_batchnorm_backward_sums(B, H, W, C, k, groups, addend, tile):
    """Round 3: the data gradient that produces a BatchNorm's incoming gradient applies the ReLU mask -- after the residual addend where there
    is one -- and leaves the BatchNorm-backward partial sums (sum g, sum g * xhat per tile and channel, per group) in a slab
    (mcav_igemm_desc.stats_x); `mcav_bn_bwd_finalize` + `mcav_bn_bwd_apply` then give the same dx, dgamma, dbeta as the three-pass form."""
    from mcav import nn as N
    g = torch.Generator().manual_seed(7 * B + H)
    cout = C
    w = torch.randn(cout, C, k, k, generator=g) * (2.0 / (C * k * k)) ** 0.5
    spec = N.ConvSpec(torch.nn.Parameter(w.to(DEV)), None, 1, k // 2, N.PAD_ZERO)
    dy = torch.randn(B, cout, H, W, generator=g)
    x_raw = torch.randn(B, C, H, W, generator=g)                 # the raw conv output of the BatchNorm being differentiated
    y_act = F.relu(torch.randn(B, C, H, W, generator=g))         # its activated output (mask)
    add = torch.randn(B, C, H, W, generator=g) if addend else None
    st = N.BNState()
    st.groups = groups
    mean = 0.1 * torch.randn(groups, C, generator=g)
    invstd = 0.5 + torch.rand(groups, C, generator=g)
    st.mean, st.invstd = mean.to(DEV), invstd.to(DEV)
    st.scale = st.shift = None
    got, slab, mtg = N.conv_dgrad(spec, nhwc(dy), (H, W), dact_aux=nhwc(y_act), dact=N.ACT_RELU | (N.DACT_AFTER_ADDEND if addend else 0),
                                  addend=None if add is None else nhwc(add), bn_stats=(nhwc(x_raw), st), tile=tile)
    want = F.conv_transpose2d(dy.double(), w.double(), stride=1, padding=k // 2)
    if add is not None:
        want = want + add.double()
    want = want * (y_act > 0).double()
    assert rel_err(nchw(got), want) < 2e-5
    # the slab's sums per group = the reduce pass's: sum of g and of g * xhat
    per = B // groups
    sums = slab.view(groups, mtg, 2, C).double().sum(1).cpu()
    for gi in range(groups):
        gsl = want[gi * per:(gi + 1) * per]
        xh = (x_raw[gi * per:(gi + 1) * per].double() - mean[gi].double().view(1, C, 1, 1)) * invstd[gi].double().view(1, C, 1, 1)
        assert rel_err(sums[gi, 0], gsl.sum((0, 2, 3))) < 1e-4
        assert rel_err(sums[gi, 1], (gsl * xh).sum((0, 2, 3))) < 1e-4
    # ... and the finalize + apply that follow equal the three-pass BatchNorm backward on the same (masked) gradient
    bn = torch.nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * torch.randn(C, generator=g))
    bn.weight.grad = torch.zeros_like(bn.weight)
    bn.bias.grad = torch.zeros_like(bn.bias)
    dx_f = N.bn_backward(bn, st, got, None, nhwc(x_raw), True, fused=(slab, mtg))
    gw_f, gb_f = bn.weight.grad.clone(), bn.bias.grad.clone()
    bn.weight.grad.zero_()
    bn.bias.grad.zero_()
    dx_u = N.bn_backward(bn, st, got, None, nhwc(x_raw), False)
    assert rel_err(dx_f, dx_u) < 1e-5 and rel_err(gw_f, bn.weight.grad) < 1e-5 and rel_err(gb_f, bn.bias.grad) < 1e-5


@pytest.mark.parametrize("dims", [(2, 5, 7), (1, 128, 208)])
def test_decoder_level_fused_upsample_concat(dims):
    """conv(cat(up2(a), skip)) with reflection padding: forward, wgrad, and the split/pooled dgrad.
    The large case has 725 pixels per wgrad split x 3 taps x 2 sources: the offset table is consumed in chunks."""
    from mcav import nn as N
    g = torch.Generator().manual_seed(4)
    (B, h, w_), C1, C2, Cout = dims, 32, 64, 32
    a = torch.randn(B, C1, h, w_, generator=g).requires_grad_()
    skip = torch.randn(B, C2, 2 * h, 2 * w_, generator=g).requires_grad_()
    wt = (torch.randn(Cout, C1 + C2, 3, 3, generator=g) * 0.05).requires_grad_()
    bs = (0.1 * torch.randn(Cout, generator=g)).requires_grad_()
    xcat = torch.cat([F.interpolate(a, scale_factor=2, mode="nearest"), skip], 1)
    pre = F.conv2d(F.pad(xcat, (1, 1, 1, 1), mode="reflect"), wt, bs)
    want = F.elu(pre)
    dy = torch.randn(pre.shape, generator=g)
    pre.backward(dy)
    wp, bp = torch.nn.Parameter(wt.detach().to(DEV)), torch.nn.Parameter(bs.detach().to(DEV))
    spec = N.ConvSpec(wp, bp, 1, 1, 1)
    got = N.conv_fwd(spec, nhwc(a.detach()), nhwc(skip.detach()), up1=True, act=N.ACT_ELU)            # merged-tap upsample path
    assert rel_err(nchw(got), want) < 2e-5
    for tile in (0x800, 10, 0x800 | 10, 12, 1):       # the nine-tap path (bit 11), and both under 32-deep / 32-row / 128-row tiles
        got = N.conv_fwd(spec, nhwc(a.detach()), nhwc(skip.detach()), up1=True, act=N.ACT_ELU, tile=tile)
        assert rel_err(nchw(got), want) < 2e-5, hex(tile)
    dyd = nhwc(dy)
    N.conv_wgrad(spec, nhwc(a.detach()), dyd, x2=nhwc(skip.detach()), up1=True, tile=0x1000)   # skip half + merged-tap upsampled half (bit 12)
    assert rel_err(wp.grad, wt.grad) < 5e-5 and rel_err(bp.grad, bs.grad) < 5e-5
    assert rel_err(wp.grad[:, :C1], wt.grad[:, :C1]) < 5e-5 and rel_err(wp.grad[:, C1:], wt.grad[:, C1:]) < 5e-5
    wp.grad = None
    bp.grad = None
    N.conv_wgrad(spec, nhwc(a.detach()), dyd, x2=nhwc(skip.detach()), up1=True, tile=0x800)   # one ordinary launch (bit 11)
    assert rel_err(wp.grad, wt.grad) < 5e-5 and rel_err(bp.grad, bs.grad) < 5e-5
    da = N.conv_dgrad(spec, dyd, (2 * h, 2 * w_), n_begin=0, n_count=C1, pool=True)
    assert rel_err(nchw(da), a.grad) < 2e-5
    # the same gradient as a 4x4 stride-2 conv on the edge-replicated low-resolution domain + fold (used for >= 64 channels), with epilogue
    old_min, N.UPMERGE_ADJ_MIN_N = N.UPMERGE_ADJ_MIN_N, 16
    try:
        aux = torch.randn(B, C1, h, w_, generator=g)
        add = torch.randn(B, C1, h, w_, generator=g)
        for tile in (0, 10, 12):
            da = N.conv_dgrad(spec, dyd, (2 * h, 2 * w_), n_begin=0, n_count=C1, pool=True, tile=tile)
            assert rel_err(nchw(da), a.grad) < 2e-5, tile
        da = N.conv_dgrad(spec, dyd, (2 * h, 2 * w_), n_begin=0, n_count=C1, pool=True, dact_aux=nhwc(aux), dact=N.ACT_ELU, addend=nhwc(add))
        want_da = a.grad * torch.where(aux > 0, torch.ones_like(aux), aux + 1) + add
        assert rel_err(nchw(da), want_da) < 2e-5
    finally:
        N.UPMERGE_ADJ_MIN_N = old_min
    ds = N.conv_dgrad(spec, dyd, (2 * h, 2 * w_), n_begin=C1, n_count=C2)
    assert rel_err(nchw(ds), skip.grad) < 2e-5


def test_stem_smallc():
    from mcav import nn as N
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 20, 28, generator=g)
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).requires_grad_()
    y = F.conv2d(x, w, None, stride=2, padding=3)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    wp = torch.nn.Parameter(w.detach().to(DEV))
    spec = N.ConvSpec(wp, None, 2, 3, 0, smallc=True)
    x4 = N.nchw_to_nhwc(x.to(DEV), 4)
    got = N.conv_fwd(spec, x4)
    assert rel_err(nchw(got), y) < 2e-5
    N.conv_wgrad(spec, x4, nhwc(dy))
    assert rel_err(wp.grad, w.grad) < 5e-5


@pytest.mark.parametrize("B,H,W", [(4, 64, 128), (2, 45, 70), (3, 192, 640)])
def test_stem_patch_kernels_forward_stats_wgrad(B, H, W):
    """csrc/conv_stem.hip: the 7x7 stride-2 image stem with the input patch in LDS -- forward with the BatchNorm statistics of two stacked
    passes, and the persistent weight-gradient kernel -- against torch, on whole, ragged and full-size maps; and against the general
    kernels (desc.tile bit 9) it replaces."""
    from mcav import nn as N
    g = torch.Generator().manual_seed(15)
    x = torch.randn(B, 3, H, W, generator=g)
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).requires_grad_()
    y = F.conv2d(x, w, None, stride=2, padding=3)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    wp = torch.nn.Parameter(w.detach().to(DEV))
    spec = N.ConvSpec(wp, None, 2, 3, 0, smallc=True)
    x4 = N.nchw_to_nhwc(x.to(DEV), 4)
    groups = 2 if B % 2 == 0 else 1
    got, slab = N.conv_fwd(spec, x4, stats=True, groups=groups)
    assert getattr(spec, "_stem", None) is not None, "the stem launch did not take the patch kernel"
    assert rel_err(nchw(got), y) < 2e-5
    mt = slab.shape[0] // groups
    assert mt == (B // groups) * (-(-y.shape[2] // 8)) * (-(-y.shape[3] // 32))
    for grp in range(groups):
        s_ = slab[grp * mt:(grp + 1) * mt].sum(0).cpu()
        part = y.detach()[grp * (B // groups):(grp + 1) * (B // groups)]
        assert rel_err(s_[0], part.sum((0, 2, 3))) < 1e-4
        assert rel_err(s_[1], (part ** 2).sum((0, 2, 3))) < 1e-4
    old = N.conv_fwd(spec, x4, tile=0x200)
    assert rel_err(got, old) < 2e-5
    N.conv_wgrad(spec, x4, nhwc(dy))
    assert rel_err(wp.grad, w.grad) < 5e-5
    N.conv_wgrad(spec, x4, nhwc(dy))                  # accumulates
    assert rel_err(wp.grad, 2 * w.grad) < 5e-5


@pytest.mark.parametrize("B,H,W", [(2, 45, 70), (4, 192, 640)])
def test_pose_stem_patch_kernels(B, H, W):
    """csrc/conv_stem.hip on PoseNet's conv1 (9 of 16 stored channels -> 16, 7x7 stride 2, bias + ReLU): forward, weight and bias gradient
    against torch and against the general kernels (desc.tile bit 9)."""
    from mcav import nn as N
    g = torch.Generator().manual_seed(16)
    x = torch.randn(B, 9, H, W, generator=g)
    w = (torch.randn(16, 9, 7, 7, generator=g) * 0.05).requires_grad_()
    b = (0.1 * torch.randn(16, generator=g)).requires_grad_()
    pre = F.conv2d(x, w, b, stride=2, padding=3)
    dy = torch.randn(pre.shape, generator=g)
    pre.backward(dy)
    wp, bp = torch.nn.Parameter(w.detach().to(DEV)), torch.nn.Parameter(b.detach().to(DEV))
    spec = N.ConvSpec(wp, bp, 2, 3, 0)
    xin = nhwc(x, 16)
    got = N.conv_fwd(spec, xin, act=N.ACT_RELU)
    assert getattr(spec, "_stem", None) is not None, "the launch did not take the patch kernel"
    assert rel_err(nchw(got), F.relu(pre)) < 2e-5
    assert rel_err(got, N.conv_fwd(spec, xin, act=N.ACT_RELU, tile=0x200)) < 2e-5
    N.conv_wgrad(spec, xin, nhwc(dy))
    assert rel_err(wp.grad, w.grad) < 5e-5
    assert rel_err(bp.grad, b.grad) < 5e-5
    N.conv_wgrad(spec, xin, nhwc(dy))                  # accumulates
    assert rel_err(wp.grad, 2 * w.grad) < 5e-5 and rel_err(bp.grad, 2 * b.grad) < 5e-5


def test_batchnorm_train_and_backward():
    from mcav import nn as N
    from mcav.holders import BNParams
    g = torch.Generator().manual_seed(6)
    B, C, H, W = 3, 64, 9, 13
    x = (1.5 * torch.randn(B, C, H, W, generator=g) + 0.7).requires_grad_()
    res = torch.randn(B, C, H, W, generator=g).requires_grad_()
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * torch.randn(C, generator=g))
        bn.bias.copy_(0.1 * torch.randn(C, generator=g))
    out = F.relu(bn(x) + res)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy)
    hb = BNParams(C).to(DEV)
    with torch.no_grad():
        hb.weight.copy_(bn.weight.detach())
        hb.bias.copy_(bn.bias.detach())
    # statistics slab as the conv epilogue would write it: one "tile" holding the sums
    xd = nhwc(x.detach())
    slab = torch.stack([xd.sum((0, 1, 2)), (xd * xd).sum((0, 1, 2))]).view(1, 2, C).contiguous()
    st = N.bn_train_coeffs(hb, slab, B * H * W)
    y = N.bn_apply(xd, st, True, nhwc(res.detach()))
    assert rel_err(nchw(y), out) < 1e-5
    assert rel_err(hb.running_mean, bn.running_mean) < 1e-5 and rel_err(hb.running_var, bn.running_var) < 1e-5
    N.flush_bn_counters()                    # the counter increments of a forward pass are applied in one fused add (the nets flush themselves)
    assert int(hb.num_batches_tracked) == 1
    dx, dz = N.bn_backward(hb, st, nhwc(dy), y, xd, True, want_dres=True)
    assert rel_err(nchw(dx), x.grad) < 1e-4
    assert rel_err(nchw(dz), res.grad) < 1e-6
    assert rel_err(hb.weight.grad, bn.weight.grad) < 1e-4 and rel_err(hb.bias.grad, bn.bias.grad) < 1e-4


def test_maxpool_adam_misc():
    from mcav import nn as N
    import ctypes
    from mcav import lib as L
    g = torch.Generator().manual_seed(7)
    x = torch.relu(torch.randn(2, 64, 11, 14, generator=g)).requires_grad_()   # ReLU zeros -> ties
    y = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    yd, idx = N.maxpool_fwd(nhwc(x.detach()))
    assert rel_err(nchw(yd), y) == 0
    dx = N.maxpool_bwd(nhwc(dy), idx, (2, 11, 14, 64))
    assert rel_err(nchw(dx), x.grad) < 1e-6
    # Adam against torch.optim.Adam for 3 steps
    p = torch.randn(1000, generator=g)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref], 1e-3)
    pd, m, v = p.to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    for step in range(1, 4):
        gr = torch.randn(1000, generator=g)
        ref.grad = gr.clone()
        opt.step()
        L.check(L.lib().mcav_adam_step(N.P(pd), N.P(gr.to(DEV)), N.P(m), N.P(v), 1000, 1e-3, 0.9, 0.999, 1e-8, step, 1.0, L.stream()), "adam")
    assert rel_err(pd, ref) < 1e-6
    # spatial mean
    t = torch.randn(3, 2, 5, 12, generator=g)
    out = N.spatial_mean(t.to(DEV), 0.06)
    assert rel_err(out, 0.06 * t.mean((1, 2))) < 1e-5


@pytest.mark.parametrize("shape", [(2, 8, 12, 16), (1, 2, 2, 16), (2, 3, 5, 32), (1, 37, 70, 16), (1, 6, 9, 64), (1, 4, 4, 128),
                                   (2, 3, 130, 16), (1, 33, 65, 32), (3, 17, 16, 64),      # H = 3 (both border rules on one row), ragged tile grids
                                   (1, 256, 256, 16)])      # the last one is large enough for the pre-multiplied-gradient path
def test_one_channel_head_stencil(shape):
    """The disparity head (3x3 reflect conv to ONE channel + sigmoid): stencil forward and the fused one-pass backward
    (d input through ELU' + addend, weight and bias gradients) against torch, including the reflected border lines."""
    from mcav import nn as N
    B, H, W, C = shape
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + C)
    pre_x = torch.randn(B, C, H, W, generator=g).requires_grad_()
    x = F.elu(pre_x)                                          # the head's input is an ELU output (decoder conv (i, 1))
    w = (torch.randn(1, C, 3, 3, generator=g) * 0.2).requires_grad_()
    b = (0.1 * torch.randn(1, generator=g)).requires_grad_()
    disp = torch.sigmoid(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b))
    ddisp = torch.randn(disp.shape, generator=g)
    addend = torch.randn(B, C, H, W, generator=g)
    disp.backward(ddisp)
    want_dx = pre_x.grad + addend                             # gradient at the pre-activation of x, plus the other branch's

    wp, bp = torch.nn.Parameter(w.detach().to(DEV)), torch.nn.Parameter(b.detach().to(DEV))
    spec = N.ConvSpec(wp, bp, 1, 1, N.PAD_REFLECT)
    xd = nhwc(x.detach())
    assert N.narrow_ok(spec, xd)
    got = N.conv3x3r_c1_fwd(spec, xd, N.ACT_SIGMOID)
    assert rel_err(nchw(got), disp) < 2e-5
    dx = N.conv3x3r_c1_bwd(spec, xd, nhwc(ddisp), got, N.ACT_SIGMOID, N.ACT_ELU, addend=nhwc(addend))
    assert rel_err(nchw(dx), want_dx) < 2e-5
    assert rel_err(wp.grad, w.grad) < 5e-5 and rel_err(bp.grad, b.grad) < 5e-5
    N.conv3x3r_c1_bwd(spec, xd, nhwc(ddisp), got, N.ACT_SIGMOID, N.ACT_ELU)       # accumulates into .grad; no addend
    assert rel_err(wp.grad, 2 * w.grad) < 5e-5


def test_halo_kernel_fused_upsample():
    """Decoder conv (0,1): 3x3 reflect conv on the nearest-upsampled 16-channel map, no skip tensor (halo-tile kernels):
    forward, and the data gradient back through the upsample (2x2 sum) and the producer's ELU, plus an addend."""
    from mcav import nn as N
    g = torch.Generator().manual_seed(11)
    for (B, h, w_, C, Cout) in [(2, 7, 19, 16, 16), (1, 1, 1, 16, 32), (2, 4, 16, 32, 16), (1, 5, 40, 32, 32)]:
        pre_a = torch.randn(B, C, h, w_, generator=g).requires_grad_()
        a = F.elu(pre_a)
        wt = (torch.randn(Cout, C, 3, 3, generator=g) * 0.1).requires_grad_()
        bs = (0.1 * torch.randn(Cout, generator=g)).requires_grad_()
        pre = F.conv2d(F.pad(F.interpolate(a, scale_factor=2, mode="nearest"), (1, 1, 1, 1), mode="reflect"), wt, bs)
        want = F.elu(pre)
        dy = torch.randn(pre.shape, generator=g)
        addend = torch.randn(B, C, h, w_, generator=g)
        pre.backward(dy)
        spec = N.ConvSpec(torch.nn.Parameter(wt.detach().to(DEV)), torch.nn.Parameter(bs.detach().to(DEV)), 1, 1, N.PAD_REFLECT)
        ad = nhwc(a.detach())
        for tile in (0, 0x400, 0x200):                  # merged-tap halo kernel (forward), the 9-tap halo kernel, the general kernels
            got = N.conv_fwd(spec, ad, None, up1=True, act=N.ACT_ELU, tile=tile)
            assert rel_err(nchw(got), want) < 2e-5
            da = N.conv_dgrad(spec, nhwc(dy), (2 * h, 2 * w_), n_begin=0, n_count=C, dact_aux=ad, dact=N.ACT_ELU, addend=nhwc(addend),
                              pool=True, tile=tile)
            assert rel_err(nchw(da), pre_a.grad + addend) < 2e-5
            spec.weight.grad = None                       # weight / bias gradient: merged-tap halo kernel, plain halo kernel, general kernel
            spec.bias.grad = None
            N.conv_wgrad(spec, ad, nhwc(dy), up1=True, tile=tile)
            assert rel_err(spec.weight.grad, wt.grad) < 5e-5 and rel_err(spec.bias.grad, bs.grad) < 5e-5, hex(tile)


def test_multi_pack_matches_single_pack():
    """One-launch re-packing of every registered filter copy (LDS-tiled) == the per-filter pack kernel, bit for bit."""
    from mcav import nn as N
    g = torch.Generator().manual_seed(21)
    shapes = [(64, 3, 7, True), (64, 64, 3, False), (128, 64, 1, False), (32, 96, 3, False), (16, 9, 7, False), (32, 16, 5, False),
              (1, 16, 3, False), (12, 256, 1, False), (256, 128, 3, False)]
    specs = []
    for cout, cin, k, smallc in shapes:
        w = torch.nn.Parameter(torch.randn(cout, cin, k, k, generator=g).to(DEV))
        specs.append(N.ConvSpec(w, None, 1, k // 2, N.PAD_ZERO, smallc=smallc))
    first = []
    for s in specs:                                   # first request: the single-filter kernel
        f = s.packed_fwd().clone()
        b = None if s.smallc else s.packed_bwd().clone()
        first.append((f, b))
    with torch.no_grad():
        for s in specs:
            s.weight.add_(0.0)                        # same values, new version: every copy is now stale
    for s, (f, b) in zip(specs, first):               # the first request re-derives ALL registered copies in one launch
        assert torch.equal(s.packed_fwd(), f)
        if b is not None:
            assert torch.equal(s.packed_bwd(), b)


def test_pose_input_pack_is_cat_in_nhwc():
    """mcav_nchw3_to_nhwc == torch.cat([tgt, ref0, ref1], 1) laid out NHWC in 16 channels, zeros past the ninth (pose_net.py:59-61)."""
    from mcav import nn as N
    g = torch.Generator().manual_seed(9)
    for B, H, W in ((2, 5, 7), (3, 64, 96)):
        imgs = [torch.randn(B, 3, H, W, generator=g).to(DEV) for _ in range(3)]
        got = N.nchw3_to_nhwc(*imgs, 16)
        want = torch.zeros(B, H, W, 16, device=DEV)
        want[..., :9] = torch.cat(imgs, 1).permute(0, 2, 3, 1)
        assert torch.equal(got, want)
    with pytest.raises(Exception):
        N.nchw3_to_nhwc(imgs[0], imgs[1], imgs[2][:, :, :-1], 16)


def test_kernel_timer_counts_conv_dispatches():
    """bench.py's roofline hook: between begin() and end() every conv kernel is timed per dispatch; results are unchanged and the
    timer is off again afterwards."""
    from mcav import nn as N
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 32, 24, 40, generator=g)
    wt = torch.randn(32, 32, 3, 3, generator=g) * 0.05
    conv = torch.nn.Conv2d(32, 32, 3, padding=1, bias=True)
    with torch.no_grad():
        conv.weight.copy_(wt)
    conv = conv.to(DEV)
    spec = N.ConvSpec(conv.weight, conv.bias, 1, 1, N.PAD_ZERO)
    xd = nhwc(x)
    y0 = N.conv_fwd(spec, xd)
    assert N.L.lib().mcav_kernel_timer_count() == 0
    N.kernel_timer_begin()
    y1 = N.conv_fwd(spec, xd)
    k = N.L.lib().mcav_kernel_timer_count()           # dispatches of one convolution (the halo kernel runs in blocks of 16 channels)
    y2 = N.conv_fwd(spec, xd)
    assert k >= 1 and N.L.lib().mcav_kernel_timer_count() == 2 * k
    durs = N.kernel_timer_end()
    assert len(durs) == 2 * k and all(0.0 < d < 50.0 for d in durs)
    assert torch.equal(y0, y1) and torch.equal(y0, y2)
    N.conv_fwd(spec, xd)
    assert N.L.lib().mcav_kernel_timer_count() == 0


@pytest.mark.parametrize("case", [(2, 12, 20, 64, 64), (2, 24, 48, 64, 128), (3, 6, 20, 128, 96), (2, 8, 16, 32, 64)])
def test_fp32_patch_in_lds_kernel_matches_the_table_driven_one(case):
    """conv3x3_patch_f32_kernel (csrc/conv_bf16.hip; opt-in: MCAV_PATCH_F32=1 or bit 13 of the descriptor's tile word): the 3x3 stride-1
    zero-padded convolution and its data gradient from a source patch kept in LDS, fp32 MFMA -- forward with bias + ReLU and with the
    BatchNorm statistics epilogue (two groups), data gradient with the ReLU-derivative factor and an addend, against float64 and against the
    table-driven kernel (reference seam: torchvision BasicBlock conv1 / conv2, resnet_dispnet.py:38-44)."""
    from mcav import nn as N
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = 0.1 * torch.randn(Cout, generator=g)
    spec = N.ConvSpec(torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(b.to(DEV)), 1, 1, N.PAD_ZERO)
    PATCH = 1 << 13
    want = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    got = N.conv_fwd(spec, nhwc(x), act=N.ACT_RELU, tile=PATCH)
    ref = N.conv_fwd(spec, nhwc(x), act=N.ACT_RELU)
    assert rel_err(nchw(got), want) < 2e-6 and rel_err(nchw(got), nchw(ref)) < 2e-6
    nob = N.ConvSpec(spec.weight, None, 1, 1, N.PAD_ZERO)
    G = 2 if B % 2 == 0 else 1
    y, slab = N.conv_fwd(nob, nhwc(x), stats=True, groups=G, tile=PATCH)
    raw = F.conv2d(x.double(), w.double(), None, padding=1)
    assert rel_err(nchw(y), raw) < 2e-6
    mt = slab.shape[0] // G
    for grp in range(G):
        part = raw[grp * (B // G):(grp + 1) * (B // G)]
        s = slab[grp * mt:(grp + 1) * mt].double().sum(0).cpu()
        assert rel_err(s[0], part.sum((0, 2, 3))) < 2e-5 and rel_err(s[1], (part ** 2).sum((0, 2, 3))) < 2e-6
    dy = torch.randn(B, Cout, H, W, generator=g)
    aux = torch.randn(B, Cin, H, W, generator=g)
    add = torch.randn(B, Cin, H, W, generator=g)
    dxw = F.conv_transpose2d(dy.double(), w.double(), padding=1) * (aux.double() > 0) + add.double()
    dx = N.conv_dgrad(spec, nhwc(dy), (H, W), dact_aux=nhwc(aux), dact=N.ACT_RELU, addend=nhwc(add), tile=PATCH)
    assert rel_err(nchw(dx), dxw) < 2e-6
