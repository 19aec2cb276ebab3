"""GPU: fp32 contractions carried by the bf16 MFMA on split operands (csrc/conv_bf16.hip, mcav_igemm_desc.mma = 2; `set_compute_dtype(m, "fp32-split")`).

Every operand element is split into three bf16 planes a = h + m + l (round to nearest; the remainder is below 2^-26 |a|) and the six plane
products hh, hm, mh, hl, lh, mm are accumulated in fp32.  That is an fp32 computation, not a reduced-precision one, and the tests hold it to the
fp32 kernels' own bar: against a float64 reference on the UNROUNDED operands the split result must be as close as the fp32-MFMA kernel's
(tools/mfma_split_test.hip measures the bare instruction sequences the same way: profiles/r03_mfma_split_exactness.txt).  Reference seam:
models/depth/resnet_dispnet.py, models/pose/pose_net.py (every nn.Conv2d of the two networks), trainer.py:261-266 (the step).
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from seeding import reinit_by_name
from test_bf16_gpu import CASES, depth_of, ref_conv64
from test_conv_gpu import nchw, nhwc

pytestmark = pytest.mark.gpu
DEV = "cuda"
SPLIT = "fp32-split"


def spec_of(w, b, stride, pad, pad_mode, mma):
    from mcav import nn as N
    spec = N.ConvSpec(torch.nn.Parameter(w.detach().to(DEV)), None if b is None else torch.nn.Parameter(b.detach().to(DEV)), stride, pad, pad_mode)
    spec.mma = mma
    return spec


def rms_err(got, ref):
    e = got.detach().double().cpu() - ref.detach().double().cpu()
    return float(e.pow(2).mean().sqrt() / ref.detach().double().pow(2).mean().sqrt())


def no_worse(err_split, err_fp32, floor=5e-7):
    """The split result is as exact as the fp32 MFMA's: within 1.5x of its error (or under a floor of a few ulp where both are tiny)."""
    return err_split <= max(1.5 * err_fp32, floor)


# + reflection padding and its adjoint on the patch kernel (round 4): several blocks per map (rows 1 / H - 2 and columns 1 / W - 2 in different
# blocks), a ragged map whose H - 2 row sits in a block of its own, 128-pixel blocks (6 x 96 x 320)
REFLECT_CASES = [(2, 24, 80, 64, 128, 3, 1, 1, 1), (1, 9, 33, 64, 32, 3, 1, 1, 1), (6, 96, 320, 64, 32, 3, 1, 1, 1), (3, 6, 20, 256, 128, 3, 1, 1, 1)]


@pytest.mark.parametrize("case", CASES + [(2, 24, 40, 64, 64, 3, 1, 1, 0), (1, 6, 20, 256, 512, 3, 2, 1, 0)] + REFLECT_CASES)
def test_split_conv_fwd_dgrad_wgrad_match_float64_as_the_fp32_kernels_do(case):
    from mcav import nn as N
    B, H, W, Cin, Cout, k, stride, pad, pad_mode = case
    g = torch.Generator().manual_seed(abs(hash(case)) % 10000 + 1)
    # operands with a wide spread of magnitudes (a log-normal scale per channel): every plane of the split carries weight
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(1.5 * torch.randn(1, Cin, 1, 1, generator=g))
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5 * torch.exp(torch.randn(Cout, 1, 1, 1, generator=g))
    b = 0.1 * torch.randn(Cout, generator=g)
    xin = nhwc(x)
    want = ref_conv64(x, w, b, stride, pad, pad_mode)
    dy = torch.randn(want.shape, generator=g) * torch.exp(1.5 * torch.randn(1, Cout, 1, 1, generator=g))
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    ref_conv64(xr, wr, None, stride, pad, pad_mode).backward(dy.double())
    errs, rms = {}, {}
    for name, mma in (("fp32", N.MMA_FP32), ("split", N.MMA_SPLIT_ALL)):
        spec = spec_of(w, b, stride, pad, pad_mode, mma)
        y = N.conv_fwd(spec, xin)
        dx = N.conv_dgrad(spec, nhwc(dy), (H, W))
        N.conv_wgrad(spec, xin, nhwc(dy))
        g1 = spec.weight.grad.clone()
        N.conv_wgrad(spec, xin, nhwc(dy))                   # accumulates
        errs[name] = (rel_err(nchw(y), want), rel_err(nchw(dx), xr.grad), rel_err(g1, wr.grad), rel_err(spec.weight.grad, 2 * wr.grad),
                      rel_err(spec.bias.grad, 2 * dy.double().sum((0, 2, 3))))
        rms[name] = (rms_err(nchw(y), want), rms_err(nchw(dx), xr.grad), rms_err(g1, wr.grad))
        if mma == N.MMA_SPLIT_ALL:
            assert "f16s" in spec._packs and "b16s" in spec._packs, "the launches did not take the split kernels"
            assert spec._packs["f16s"].dtype == torch.bfloat16 and spec._packs["f16s"].shape[0] == 3 * spec.np
    print("split vs fp32 kernels against float64 %s: fwd %.2e / %.2e, dgrad %.2e / %.2e, wgrad %.2e / %.2e" %
          (case, errs["split"][0], errs["fp32"][0], errs["split"][1], errs["fp32"][1], errs["split"][2], errs["fp32"][2]))
    # fwd / dgrad / wgrad: the LARGEST element error no worse than the fp32 kernel's -- or, where one element of a small deep-K case decides
    # that maximum either way (5 x 7 map, K = 2304: over eight draws the maximum is 2.7-8.7e-7 for the split form and 2.6-6.1e-7 for the fp32
    # kernel with either padding, tools/split_seed_scan.py), the rms error no worse.  The accumulated weight gradient and the bias gradient (a
    # plain fp32 sum over pixels in a different order) are held to the absolute bar.
    for i, (es, ef) in enumerate(zip(errs["split"], errs["fp32"])):
        assert es < 3e-6, (errs["split"], errs["fp32"])
        if i < 3:
            assert no_worse(es, ef) or rms["split"][i] <= 1.25 * rms["fp32"][i], (i, errs["split"], errs["fp32"], rms)


def test_split_filter_planes_add_up_to_the_filter():
    """The packed planes (first use: mcav_f32_to_bf16_planes; after an optimiser step: mcav_pack_weights_multi with the planes flag) hold
    h + m + l = the fp32 packed filter to 2^-24 of each element, in both the forward and the transposed (data-gradient) copy."""
    from mcav import nn as N
    g = torch.Generator().manual_seed(3)
    w = torch.randn(96, 80, 3, 3, generator=g) * torch.exp(2.0 * torch.randn(96, 80, 1, 1, generator=g))
    spec = spec_of(w, None, 1, 1, 0, N.MMA_SPLIT_ALL)
    for kind32, kind16 in (("f", "f16s"), ("b", "b16s")):
        for rnd in range(2):
            f32 = spec._packed(kind32).double()
            pl = spec._packed(kind16).double()
            rows = f32.shape[0]
            total = pl[:rows] + pl[rows:2 * rows] + pl[2 * rows:]
            assert float((total - f32).abs().max()) <= 2.0 ** -24 * float(f32.abs().max())
            assert float(((total - f32).abs() / f32.abs().clamp_min(1e-30))[f32 != 0].max()) <= 2.0 ** -23
            assert float(pl[rows:2 * rows].abs().max()) <= 2.0 ** -8 * float(f32.abs().max())
            with torch.no_grad():                            # a new weight version: both copies are re-derived by the registry's one launch
                spec.weight.mul_(1.25).add_(0.001)


@pytest.mark.parametrize("B,H,W", [(4, 10, 14), (12, 48, 160)])
def test_split_batchnorm_statistics_epilogue_and_groups(B, H, W):
    from mcav import nn as N
    g = torch.Generator().manual_seed(5)
    C = 64
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.05
    spec = spec_of(w, None, 1, 1, 0, N.MMA_SPLIT)
    y, slab = N.conv_fwd(spec, nhwc(x), stats=True, groups=2)
    want = ref_conv64(x, w, None, 1, 1, 0)
    assert rel_err(nchw(y), want) < 2e-6
    mt = slab.shape[0] // 2
    for grp in range(2):
        s = slab[grp * mt:(grp + 1) * mt].double().sum(0).cpu()
        part = want[grp * (B // 2):(grp + 1) * (B // 2)]
        assert rel_err(s[0], part.sum((0, 2, 3))) < 2e-5
        assert rel_err(s[1], (part ** 2).sum((0, 2, 3))) < 2e-6


@pytest.mark.parametrize("mode,Cout", [("all", 64), ("default", 64), ("default", 32)])
def test_split_decoder_level_fused_upsample_concat(mode, Cout):
    """A decoder level conv(cat(up2(a), skip)).  "all": every launch on the split kernels; "default": mma = 2 as the networks run it -- the weight
    gradient as the skip half on the patch kernel (32 outputs: its narrow form) + the upsampled half in merged-tap form on the fp32 MFMA."""
    from mcav import nn as N
    g = torch.Generator().manual_seed(6)
    B, h, w_, C1, C2 = 2, 6, 10, 64, 64
    a = torch.randn(B, C1, h, w_, generator=g)
    skip = torch.randn(B, C2, 2 * h, 2 * w_, generator=g)
    wt = torch.randn(Cout, C1 + C2, 3, 3, generator=g) * 0.05
    bs = 0.1 * torch.randn(Cout, generator=g)
    spec = spec_of(wt, bs, 1, 1, 1, N.MMA_SPLIT_ALL if mode == "all" else N.MMA_SPLIT)
    ar, sr, wr = a.double().requires_grad_(), skip.double().requires_grad_(), wt.double().requires_grad_()
    xcat = torch.cat([F.interpolate(ar, scale_factor=2, mode="nearest"), sr], 1)
    pre = F.conv2d(F.pad(xcat, (1, 1, 1, 1), mode="reflect"), wr, bs.double())
    dy = torch.randn(pre.shape, generator=g)
    pre.backward(dy.double())
    got = N.conv_fwd(spec, nhwc(a), nhwc(skip), up1=True, act=N.ACT_ELU)
    assert rel_err(nchw(got), F.elu(pre)) < 2e-6
    N.conv_wgrad(spec, nhwc(a), nhwc(dy), x2=nhwc(skip), up1=True)
    assert rel_err(spec.weight.grad, wr.grad) < 2e-6
    assert rel_err(spec.bias.grad, dy.double().sum((0, 2, 3))) < 2e-6
    dskip = N.conv_dgrad(spec, nhwc(dy), (2 * h, 2 * w_), n_begin=C1, n_count=C2)
    assert rel_err(nchw(dskip), sr.grad) < 2e-6


@pytest.mark.parametrize("B,H,W", [(2, 64, 128), (4, 192, 640), (12, 192, 640)])
def test_split_depth_maps_within_north_star_bound_of_cpu_oracle(B, H, W):
    """north_star: depth maps match the reference PyTorch-CPU path on identical inputs within 1e-3 relative -- the fp32 path's bound, held by
    the split path with the same margin (the fp32-MFMA path's own error is printed beside it)."""
    from mcav import nn as N
    from models.depth.resnet_dispnet import DispResNet
    from oracle import nets as on
    hip = reinit_by_name(DispResNet(dtype=SPLIT if B != 12 else None), 141)          # (batch 12 = configs[1]'s own shape, on the DEFAULT mode)
    assert getattr(hip.encoder.encoder.layer1[0].conv1, "_mcav_mma", N.DEFAULT_MMA) == N.MMA_SPLIT
    ref = on.DispResNet()
    ref.load_state_dict(hip.state_dict())
    hip.to(DEV).train()
    ref.train()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 3, H, W, generator=g)
    x = F.avg_pool2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), 3, 1)
    with torch.no_grad():
        want = ref(x)[0]
    dw = depth_of(want)
    rel = (depth_of(hip(x.to(DEV))[0]) - dw).abs() / dw
    N.set_compute_dtype(hip, "fp32-mfma")
    rel32 = (depth_of(hip(x.to(DEV))[0]) - dw).abs() / dw
    print("depth net %dx%dx%d vs CPU oracle: split max-rel %.3e mean %.3e | fp32 MFMA max-rel %.3e mean %.3e" %
          (B, H, W, float(rel.max()), float(rel.mean()), float(rel32.max()), float(rel32.mean())))
    assert float(rel.max()) < 1e-3
    assert float(rel.max()) < 3 * float(rel32.max()) + 1e-5


def test_split_train_step_equals_fp32_step_to_fp32_rounding():
    """One whole training step (both networks on the split kernels, fused loss, backward): losses and every gradient agree with the fp32-MFMA
    step to the level two fp32 evaluation orders differ by; bit-reproducible."""
    from losses import Losses
    from mcav.optim import FusedAdam
    from models.depth.resnet_dispnet import DispResNet
    from models.pose.pose_net import PoseNet
    from oracle.step import synthetic_batch
    s = synthetic_batch(4, 96, 320, seed=9)
    tgt, refs, K = s["tgt"].to(DEV), [r.to(DEV) for r in s["ref_imgs"]], s["intrinsics"].to(DEV)
    out = {}
    for name, dt in (("fp32", "fp32-mfma"), ("split", None), ("split_again", SPLIT)):          # (None = the default = the split form, round 4)
        d = reinit_by_name(DispResNet(dtype=dt), 141).to(DEV).train()
        p = reinit_by_name(PoseNet(dtype=dt), 121).to(DEV).train()
        with torch.no_grad():
            p.pose_pred.weight.mul_(0.1)
            p.pose_pred.bias.mul_(0.1)
        opt = FusedAdam(list(d.parameters()) + list(p.parameters()), 1e-4)
        opt.zero_grad()
        disps = list(d.forward_pair(tgt, refs[0]))
        loss = Losses().forward(tgt, refs, disps, p(tgt, refs), K, None)
        sum(loss).backward()
        torch.cuda.synchronize()
        out[name] = ([float(l.detach()) for l in loss], opt.arena().gflat.clone(),
                     dict((n, q.grad.clone()) for n, q in list(d.named_parameters()) + list(p.named_parameters()) if q.grad is not None))
    (l32, g32, n32), (ls, gs, ns), (ls2, gs2, _) = out["fp32"], out["split"], out["split_again"]
    assert ls == ls2 and torch.equal(gs, gs2)
    assert torch.isfinite(gs).all()
    for a, b in zip(ls, l32):
        assert abs(a - b) < 2e-5 * abs(b), (ls, l32)
    worst = 0.0
    for n in n32:
        if float(n32[n].norm()) == 0.0:                      # (parameters the step does not use: the encoder's fc)
            assert float(ns[n].norm()) == 0.0
            continue
        e = float((ns[n] - n32[n]).norm() / n32[n].norm().clamp_min(1e-30))
        worst = max(worst, e)
        cos = float(F.cosine_similarity(ns[n].flatten().double(), n32[n].flatten().double(), dim=0))
        assert cos > 0.999, (n, cos, e)
    print("split vs fp32-MFMA step: losses %s vs %s, worst per-tensor relative L2 gradient difference %.2e" % (ls, l32, worst))
    assert float((gs - g32).norm() / g32.norm()) < 5e-3


def _run_both(x, w, dy, H, W):
    """fwd / dgrad of one 3x3 stride-1 zero-padded convolution on the split patch kernel and on the fp32 MFMA kernel -> {name: (y, dx)} NCHW on the CPU."""
    from mcav import nn as N
    out = {}
    for name, mma in (("fp32", N.MMA_FP32), ("split", N.MMA_SPLIT)):
        spec = spec_of(w, None, 1, 1, 0, mma)
        y = N.conv_fwd(spec, nhwc(x))
        dx = N.conv_dgrad(spec, nhwc(dy), (H, W))
        if mma == N.MMA_SPLIT:
            assert "f16s" in spec._packs and "b16s" in spec._packs, "the launches did not take the split patch kernel"
        out[name] = (nchw(y).cpu(), nchw(dx).cpu())
    return out


@pytest.mark.parametrize("sx,sw", [(1e-30, 1e30), (1e30, 1e-30), (1e-18, 1.0), (1e18, 1e-18)])
def test_split_extreme_magnitudes_match_float64_as_the_fp32_kernel_does(sx, sw):
    """the m / l planes of a tiny operand sit 2^-9 / 2^-18 below it.  bf16 has fp32's exponent range, so all three planes of
    |a| >= 2^-126 x 2^18 = 3e-33 are normal numbers and the split stays exact: activations of 1e-30 against weights of 1e30 (and the mirror
    image) give the fp32 kernel's error against float64, not worse.  Below 3e-33 the l plane, below 6e-36 the m plane leave the normal range:
    those elements degrade towards 2^-17 / 2^-8 relative (documented in DESIGN.md section 4c; the fp32 MFMA treats them as fp32 denormals'
    neighbours too) -- no activation or gradient of this path is that small (Adam's own epsilon is 1e-8)."""
    B, C, H, W = 2, 64, 12, 20
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, C, H, W, generator=g) * sx
    w = torch.randn(C, C, 3, 3, generator=g) * (0.05 * sw)
    dy = torch.randn(B, C, H, W, generator=g) * sx
    want = ref_conv64(x, w, None, 1, 1, 0)
    xr = x.double().requires_grad_()
    ref_conv64(xr, w.double(), None, 1, 1, 0).backward(dy.double())
    got = _run_both(x, w, dy, H, W)
    e = {k: (rel_err(v[0], want), rel_err(v[1], xr.grad)) for k, v in got.items()}
    print("scales x %.0e w %.0e: split fwd %.2e dgrad %.2e | fp32 MFMA fwd %.2e dgrad %.2e" % (sx, sw, e["split"][0], e["split"][1], e["fp32"][0], e["fp32"][1]))
    for es, ef in zip(e["split"], e["fp32"]):
        assert torch.isfinite(torch.tensor(es)) and es < 3e-6 and no_worse(es, ef), e


def test_split_non_finite_operands_give_non_finite_results_where_the_fp32_kernel_does():
    """An inf or NaN in an activation or a gradient must not disappear: every output the fp32 MFMA kernel leaves non-finite is non-finite on the
    split kernel too, and every other output is the same finite number as without the bad element.  (One difference, by construction: h = bf16(inf)
    = inf and m = bf16(inf - inf) = NaN, so where the fp32 kernel answers +-inf the split form answers NaN.  Both are caught by the same
    isfinite test a training loop applies; fp32 values above bf16's largest finite number, 3.39e38, also round to inf in the h plane.)"""
    B, C, H, W = 1, 64, 8, 16
    g = torch.Generator().manual_seed(19)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.05
    dy = torch.randn(B, C, H, W, generator=g)
    clean = _run_both(x, w, dy, H, W)
    xb, dyb = x.clone(), dy.clone()
    xb[0, 3, 2, 5] = float("inf")
    xb[0, 40, 6, 11] = float("nan")
    dyb[0, 7, 4, 4] = float("-inf")
    bad = _run_both(xb, w, dyb, H, W)
    for i, what in ((0, "forward"), (1, "data gradient")):
        nf32, nfs = ~torch.isfinite(bad["fp32"][i]), ~torch.isfinite(bad["split"][i])
        assert bool(nf32.any()), what
        assert torch.equal(nf32, nfs), "%s: the split kernel's non-finite outputs differ from the fp32 kernel's (%d vs %d)" % (what, int(nfs.sum()), int(nf32.sum()))
        same = ~nfs
        assert torch.equal(bad["split"][i][same], clean["split"][i][same]), what          # untouched outputs are bit-identical
    # the 3x3 neighbourhood of each bad pixel, every output channel
    assert int((~torch.isfinite(bad["split"][0])).sum()) == 2 * 9 * C


@pytest.mark.parametrize("tile_bits", [1 << 14, 3 << 14])
@pytest.mark.parametrize("case", [(4, 12, 20, 64, 64), (4, 24, 48, 64, 128), (6, 6, 20, 128, 96), (2, 48, 32, 32, 64)])
def test_second_patch_kernel_matches_float64_as_the_first_one_does(case, tile_bits):
    """conv3x3_patch2_kernel (csrc/conv_bf16.hip; opt-in through mcav_igemm_desc.tile bits 14 / 15: measured level with the first patch kernel, so
    it is off by default): several 64-pixel sub-blocks -- of different images on the small maps -- per workgroup, filter tiles by LDS-DMA into a
    two-stage ring whose LDS image is permuted on the SOURCE side.  Forward with bias + ReLU + grouped BatchNorm statistics, and the data
    gradient, against float64: the same error as the first kernel's (both are the six-product split form)."""
    from mcav import nn as N
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case) + tile_bits % 97)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(1, Cin, 1, 1, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = 0.1 * torch.randn(Cout, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    want = torch.relu(ref_conv64(x, w, b, 1, 1, 0))
    raw = ref_conv64(x, w, None, 1, 1, 0)
    xr = x.double().requires_grad_()
    ref_conv64(xr, w.double(), None, 1, 1, 0).backward(dy.double())
    errs = {}
    for name, tile in (("first", 0), ("second", tile_bits)):
        spec = spec_of(w, b, 1, 1, 0, N.MMA_SPLIT)
        y = N.conv_fwd(spec, nhwc(x), act=N.ACT_RELU, tile=tile)
        dx = N.conv_dgrad(spec, nhwc(dy), (H, W), tile=tile)
        spec0 = spec_of(w, None, 1, 1, 0, N.MMA_SPLIT)
        yraw, slab = N.conv_fwd(spec0, nhwc(x), stats=True, groups=2, tile=tile)
        mt = slab.shape[0] // 2
        s0 = torch.stack([slab[grp * mt:(grp + 1) * mt].double().sum(0).cpu() for grp in range(2)])      # [group][sum | sum of squares][channel]
        part = [raw[grp * (B // 2):(grp + 1) * (B // 2)] for grp in range(2)]
        errs[name] = (rel_err(nchw(y), want), rel_err(nchw(dx), xr.grad), rel_err(nchw(yraw), raw),
                      max(rel_err(s0[grp][0], part[grp].sum((0, 2, 3))) for grp in range(2)),
                      max(rel_err(s0[grp][1], (part[grp] ** 2).sum((0, 2, 3))) for grp in range(2)), slab.shape[0])
    print("patch kernels %s bits %x: first %s | second %s" % (case, tile_bits, ["%.1e" % e for e in errs["first"][:5]], ["%.1e" % e for e in errs["second"][:5]]))
    assert errs["second"][5] < errs["first"][5], "the second kernel did not run (same statistics rows as the first)"
    for es, ef in zip(errs["second"][:3], errs["first"][:3]):
        assert es < 3e-6 and no_worse(es, ef), errs
    assert errs["second"][3] < 3e-5 and errs["second"][4] < 3e-6, errs


@pytest.mark.parametrize("case", [(6, 12, 20, 64, 64, 0), (6, 24, 48, 64, 128, 0), (6, 6, 20, 128, 96, 0), (2, 48, 32, 32, 64, 0), (4, 24, 48, 64, 64, 1),
                                  (3, 9, 33, 64, 64, 1), (5, 12, 40, 64, 64, 0)])
def test_twelve_wavefront_patch_kernel_matches_float64_as_the_first_one_does(case):
    """conv3x3_patch3_kernel (low byte 0x33 of mcav_igemm_desc.tile): three 64-pixel blocks per workgroup, twelve wavefronts, filter tiles in a
    two-stage ring.  Forward with bias + ReLU, grouped BatchNorm statistics (zero padding), the data gradient -- with reflection padding its
    adjoint (border extras per block) -- against float64: the same error as the first kernel's; block counts that are no multiple of three."""
    from mcav import nn as N
    B, H, W, Cin, Cout, pm = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(1, Cin, 1, 1, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = 0.1 * torch.randn(Cout, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    want = torch.relu(ref_conv64(x, w, b, 1, 1, pm))
    raw = ref_conv64(x, w, None, 1, 1, pm)
    xr = x.double().requires_grad_()
    ref_conv64(xr, w.double(), None, 1, 1, pm).backward(dy.double())
    grouped = pm == 0 and B % 2 == 0
    errs = {}
    for name, tile in (("first", 0), ("twelve", 0x33)):
        spec = spec_of(w, b, 1, 1, pm, N.MMA_SPLIT_ALL)
        y = N.conv_fwd(spec, nhwc(x), act=N.ACT_RELU, tile=tile)
        dx = N.conv_dgrad(spec, nhwc(dy), (H, W), tile=tile)
        e = [rel_err(nchw(y), want), rel_err(nchw(dx), xr.grad)]
        rows = 0
        if grouped:
            spec0 = spec_of(w, None, 1, 1, pm, N.MMA_SPLIT_ALL)
            yraw, slab = N.conv_fwd(spec0, nhwc(x), stats=True, groups=2, tile=tile)
            mt = slab.shape[0] // 2
            rows = slab.shape[0]
            s0 = torch.stack([slab[grp * mt:(grp + 1) * mt].double().sum(0).cpu() for grp in range(2)])
            part = [raw[grp * (B // 2):(grp + 1) * (B // 2)] for grp in range(2)]
            e += [rel_err(nchw(yraw), raw), max(rel_err(s0[grp][0], part[grp].sum((0, 2, 3))) for grp in range(2)),
                  max(rel_err(s0[grp][1], (part[grp] ** 2).sum((0, 2, 3))) for grp in range(2))]
        errs[name] = (e, rows)
    print("patch kernels %s: first %s | twelve-wavefront %s" % (case, ["%.1e" % v for v in errs["first"][0]], ["%.1e" % v for v in errs["twelve"][0]]))
    if grouped:
        assert errs["twelve"][1] < errs["first"][1], "the twelve-wavefront kernel did not run (same statistics rows as the first)"
        assert errs["twelve"][0][3] < 3e-5 and errs["twelve"][0][4] < 3e-6, errs
    for es, ef in zip(errs["twelve"][0][:3], errs["first"][0][:3]):
        assert es < 3e-6 and no_worse(es, ef), errs


@pytest.mark.parametrize("case", [(4, 48, 160, 64, 64, 0), (2, 24, 80, 128, 128, 1), (3, 7, 21, 64, 128, 1), (24, 6, 20, 128, 64, 0), (1, 12, 40, 64, 72, 0),
                                  (4, 24, 80, 64, 32, 1)])
def test_default_mode_weight_gradient_takes_the_patch_kernel_and_matches_float64(case):
    """Round 4: under the DEFAULT mode (mcav_wgrad_desc.mma = 2) the weight gradients of the single-source 3x3 stride-1 layers with 64-channel
    multiples run wgrad3x3_patch_kernel (transposing LDS reads over the staged patch, six plane products in one fp32 accumulator); everything
    else stays on the fp32 MFMA kernels.  Against float64 on the unrounded operands the result is as close as the fp32 kernel's: zero and
    reflection padding, blocks of several shapes (4 x 16, 3 x 20 with a ragged map, 6 x 10), 24 images, a Cout that is no multiple of 64, 32
    outputs (the narrow form: wavefront pairs share a block's pixel steps), bias gradient (the MFMA against ones) and accumulation into an existing gradient."""
    from mcav import nn as N
    B, H, W, Cin, Cout, pad_mode = case
    g = torch.Generator().manual_seed(1234 + H)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(1.5 * torch.randn(1, Cin, 1, 1, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = 0.1 * torch.randn(Cout, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g) * torch.exp(1.5 * torch.randn(1, Cout, 1, 1, generator=g))
    wr = w.double().requires_grad_()
    ref_conv64(x.double(), wr, None, 1, 1, pad_mode).backward(dy.double())
    want_b = dy.double().sum((0, 2, 3))
    errs, rms = {}, {}
    for name, mma in (("fp32", N.MMA_FP32), ("split", N.MMA_SPLIT)):
        spec = spec_of(w, b, 1, 1, pad_mode, mma)
        N.conv_wgrad(spec, nhwc(x), nhwc(dy))
        assert N.LAST_WGRAD_PLANES == (6 if mma == N.MMA_SPLIT else 0), "the launch did not take the expected kernel"
        g1 = spec.weight.grad.clone()
        N.conv_wgrad(spec, nhwc(x), nhwc(dy))                # accumulates
        errs[name] = (rel_err(g1, wr.grad), rel_err(spec.weight.grad, 2 * wr.grad), rel_err(spec.bias.grad, 2 * want_b))
        rms[name] = rms_err(g1, wr.grad)
    print("wgrad %s: split %.2e (rms %.2e) / fp32 %.2e (rms %.2e); bias %.2e / %.2e" %
          (case, errs["split"][0], rms["split"], errs["fp32"][0], rms["fp32"], errs["split"][2], errs["fp32"][2]))
    assert all(e < 3e-6 for e in errs["split"]), errs
    assert no_worse(errs["split"][0], errs["fp32"][0]) or rms["split"] <= 1.25 * rms["fp32"], (errs, rms)
    # ... and a shape the patch kernel does not cover (32 input channels) stays on the fp32 MFMA kernel under the same mode
    spec = spec_of(torch.randn(64, 32, 3, 3, generator=g) * 0.05, None, 1, 1, 0, N.MMA_SPLIT)
    N.conv_wgrad(spec, torch.randn(2, 12, 20, 32, generator=g).to(DEV), torch.randn(2, 12, 20, 64, generator=g).to(DEV))
    assert N.LAST_WGRAD_PLANES == 0


@pytest.mark.parametrize("B,H,W", [(4, 64, 128), (2, 45, 70), (6, 192, 640)])
def test_split_depth_stem_matches_float64_as_the_fp32_stem_kernel_does(B, H, W):
    """The depth net's image stem (conv 7x7 stride 2, 3 -> 64 on the NHWC4 image; reference resnet_dispnet.py: torchvision conv1) in the split
    form (mma = 3; measured level with the fp32 stem kernel, so the default mode keeps that one): stem7x7s2_split_fwd_kernel -- 14 k-steps of the bf16 MFMA on three planes per operand, every A fragment one aligned 16-byte
    read of the staged image patch -- against float64, no worse than the fp32 stem kernel; BatchNorm statistics of two stacked passes; whole,
    ragged and full-size maps."""
    from mcav import nn as N
    g = torch.Generator().manual_seed(15 + H)
    x = torch.randn(B, 3, H, W, generator=g) * torch.tensor([1.0, 0.05, 4.0]).view(1, 3, 1, 1)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1 * torch.exp(torch.randn(64, 1, 1, 1, generator=g))
    want = F.conv2d(x.double(), w.double(), None, stride=2, padding=3)
    x4 = N.nchw_to_nhwc(x.to(DEV), 4)
    groups = 2 if B % 2 == 0 else 1
    errs = {}
    for name, mma in (("fp32", N.MMA_FP32), ("split", N.MMA_SPLIT_ALL)):
        spec = N.ConvSpec(torch.nn.Parameter(w.to(DEV)), None, 2, 3, 0, smallc=True)
        spec.mma = mma
        got, slab = N.conv_fwd(spec, x4, stats=True, groups=groups)
        assert getattr(spec, "_stem", None) is not None
        mt = slab.shape[0] // groups
        e = [rel_err(nchw(got), want), rms_err(nchw(got), want)]
        for grp in range(groups):
            s_ = slab[grp * mt:(grp + 1) * mt].double().sum(0).cpu()
            part = want[grp * (B // groups):(grp + 1) * (B // groups)]
            e += [rel_err(s_[0], part.sum((0, 2, 3))), rel_err(s_[1], (part ** 2).sum((0, 2, 3)))]
        errs[name] = e
    print("depth stem %s: split %s | fp32 %s" % ((B, H, W), ["%.1e" % v for v in errs["split"]], ["%.1e" % v for v in errs["fp32"]]))
    assert errs["split"][0] < 3e-6 and (no_worse(errs["split"][0], errs["fp32"][0]) or errs["split"][1] <= 1.25 * errs["fp32"][1]), errs
    assert all(v < 1e-4 for v in errs["split"][2:]), errs
