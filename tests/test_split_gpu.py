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


def no_worse(err_split, err_fp32, floor=5e-7):
    """The split result is as exact as the fp32 MFMA's: within 1.5x of its error (or under a floor of a few ulp where both are tiny)."""
    return err_split <= max(1.5 * err_fp32, floor)


@pytest.mark.parametrize("case", CASES + [(2, 24, 40, 64, 64, 3, 1, 1, 0), (1, 6, 20, 256, 512, 3, 2, 1, 0)])
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
    errs = {}
    for name, mma in (("fp32", N.MMA_FP32), ("split", N.MMA_SPLIT_ALL)):
        spec = spec_of(w, b, stride, pad, pad_mode, mma)
        y = N.conv_fwd(spec, xin)
        dx = N.conv_dgrad(spec, nhwc(dy), (H, W))
        N.conv_wgrad(spec, xin, nhwc(dy))
        g1 = spec.weight.grad.clone()
        N.conv_wgrad(spec, xin, nhwc(dy))                   # accumulates
        errs[name] = (rel_err(nchw(y), want), rel_err(nchw(dx), xr.grad), rel_err(g1, wr.grad), rel_err(spec.weight.grad, 2 * wr.grad),
                      rel_err(spec.bias.grad, 2 * dy.double().sum((0, 2, 3))))
        if mma == N.MMA_SPLIT_ALL:
            assert "f16s" in spec._packs and "b16s" in spec._packs, "the launches did not take the split kernels"
            assert spec._packs["f16s"].dtype == torch.bfloat16 and spec._packs["f16s"].shape[0] == 3 * spec.np
    print("split vs fp32 kernels against float64 %s: fwd %.2e / %.2e, dgrad %.2e / %.2e, wgrad %.2e / %.2e" %
          (case, errs["split"][0], errs["fp32"][0], errs["split"][1], errs["fp32"][1], errs["split"][2], errs["fp32"][2]))
    for es, ef in zip(errs["split"], errs["fp32"]):
        assert es < 3e-6 and no_worse(es, ef), (errs["split"], errs["fp32"])


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


def test_split_decoder_level_fused_upsample_concat():
    from mcav import nn as N
    g = torch.Generator().manual_seed(6)
    B, h, w_, C1, C2, Cout = 2, 6, 10, 64, 64, 64
    a = torch.randn(B, C1, h, w_, generator=g)
    skip = torch.randn(B, C2, 2 * h, 2 * w_, generator=g)
    wt = torch.randn(Cout, C1 + C2, 3, 3, generator=g) * 0.05
    bs = 0.1 * torch.randn(Cout, generator=g)
    spec = spec_of(wt, bs, 1, 1, 1, N.MMA_SPLIT_ALL)
    ar, sr, wr = a.double().requires_grad_(), skip.double().requires_grad_(), wt.double().requires_grad_()
    xcat = torch.cat([F.interpolate(ar, scale_factor=2, mode="nearest"), sr], 1)
    pre = F.conv2d(F.pad(xcat, (1, 1, 1, 1), mode="reflect"), wr, bs.double())
    dy = torch.randn(pre.shape, generator=g)
    pre.backward(dy.double())
    got = N.conv_fwd(spec, nhwc(a), nhwc(skip), up1=True, act=N.ACT_ELU)
    assert rel_err(nchw(got), F.elu(pre)) < 2e-6
    N.conv_wgrad(spec, nhwc(a), nhwc(dy), x2=nhwc(skip), up1=True)
    assert rel_err(spec.weight.grad, wr.grad) < 2e-6
    dskip = N.conv_dgrad(spec, nhwc(dy), (2 * h, 2 * w_), n_begin=C1, n_count=C2)
    assert rel_err(nchw(dskip), sr.grad) < 2e-6


@pytest.mark.parametrize("B,H,W", [(2, 64, 128), (4, 192, 640)])
def test_split_depth_maps_within_north_star_bound_of_cpu_oracle(B, H, W):
    """north_star: depth maps match the reference PyTorch-CPU path on identical inputs within 1e-3 relative -- the fp32 path's bound, held by
    the split path with the same margin (the fp32-MFMA path's own error is printed beside it)."""
    from mcav import nn as N
    from models.depth.resnet_dispnet import DispResNet
    from oracle import nets as on
    hip = reinit_by_name(DispResNet(dtype=SPLIT), 141)
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
    N.set_compute_dtype(hip, torch.float32)
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
    for name, dt in (("fp32", None), ("split", SPLIT), ("split_again", SPLIT)):
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
