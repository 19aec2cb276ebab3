"""GPU: the HIP networks (reference call surface) against golden vectors from the reference modules."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from seeding import reinit_by_name

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def grads_match(model, g, names, tol, l2=False):
    sd = dict(model.named_parameters())
    for n in names:
        gr = sd[n].grad
        assert gr is not None, n
        if gr.numel() > 65536:
            gr = gr[:8, :8]
        want = torch.from_numpy(g["g_" + n.replace(".", "_")])
        if l2:      # deep ReLU/BatchNorm stacks: a few activation-mask flips move single entries; compare in the L2 sense
            assert float((gr.cpu() - want).norm() / want.norm()) < tol, n
        else:
            assert rel_err(gr, want) < tol, n


def test_posenet_vs_reference(golden):
    from models.pose.pose_net import PoseNet
    g = golden("posenet.npz")
    m = reinit_by_name(PoseNet(), 21).to(DEV)
    out = m(T(g["tgt"]), [T(g["ref0"]), T(g["ref1"])])
    assert out.shape == (2, 2, 6)
    assert rel_err(out, g["out"]) < 1e-4
    (out * T(g["coef"])).sum().backward()
    grads_match(m, g, ["conv1.0.weight", "conv1.0.bias", "conv4.0.weight", "conv7.0.bias", "pose_pred.weight", "pose_pred.bias"], 1e-3)


def test_decoder_vs_reference(golden):
    from models.depth.resnet_dispnet import DepthDecoder
    g = golden("decoder.npz")
    dec = reinit_by_name(DepthDecoder(np.array([64, 64, 128, 256, 512])), 31).to(DEV)
    feats = [T(g["f%d" % i]).requires_grad_() for i in range(5)]
    out = dec(feats)
    for s in range(4):
        assert rel_err(out[("disp", s)], g["disp%d" % s]) < 1e-4
    (out[("disp", 0)] * T(g["coef"])).sum().backward()
    for i in range(5):
        assert rel_err(feats[i].grad, g["g_f%d" % i]) < 1e-3, i
    grads_match(dec, g, ["decoder.0.conv.conv.weight", "decoder.0.conv.conv.bias", "decoder.7.conv.conv.weight",
                         "decoder.9.conv.conv.weight", "decoder.9.conv.conv.bias", "decoder.10.conv.weight", "decoder.10.conv.bias"], 1e-3)


def test_dispresnet_vs_reference(golden):
    from models.depth.resnet_dispnet import DispResNet
    g = golden("dispresnet.npz")
    m = reinit_by_name(DispResNet(), 41).to(DEV)
    m.train()
    out = m(T(g["x"]))[0]
    assert out.shape == (2, 1, 64, 128)
    # north_star parity criterion: depth maps within 1e-3 relative
    assert rel_err(out, g["disp"]) < 1e-3
    depth_got, depth_want = 1 / (10 * out.detach().cpu() + 0.01), 1 / (10 * torch.from_numpy(g["disp"]) + 0.01)
    abs_rel = float(((depth_got - depth_want).abs() / depth_want).mean())
    assert abs_rel < 1e-4, abs_rel
    bn1 = m.encoder.encoder.bn1
    assert rel_err(bn1.running_mean, g["running_mean_bn1"]) < 1e-4 and rel_err(bn1.running_var, g["running_var_bn1"]) < 1e-4
    (out * T(g["coef"])).sum().backward()
    grads_match(m, g, ["encoder.encoder.conv1.weight", "encoder.encoder.bn1.weight", "encoder.encoder.bn1.bias",
                       "encoder.encoder.layer2.0.downsample.0.weight", "encoder.encoder.layer4.1.bn2.weight",
                       "decoder.decoder.0.conv.conv.bias", "decoder.decoder.10.conv.weight"], 5e-3)


def test_dispnets_vs_reference(golden):
    from models.depth.disp_net import DispNetS
    g = golden("dispnets.npz")
    m = reinit_by_name(DispNetS(), 51).to(DEV)
    m.train()
    outs = m(T(g["x"]))
    assert len(outs) == 4
    for i, o in enumerate(outs):
        assert rel_err(o, g["disp%d" % (i + 1)]) < 1e-3, i
    sum((o * T(g["coef%d" % (i + 1)])).sum() for i, o in enumerate(outs)).backward()
    grads_match(m, g, ["conv1.0.weight", "conv1.2.weight", "upconv7.0.weight", "upconv1.0.bias", "iconv3.0.weight", "predict_disp1.0.weight"], 1e-2, l2=True)


def test_dispnets_gradients_vs_fp64_arbiter():
    """Every DispNetS parameter gradient under the float64 arbiter (tests/arbiter.py), as the ResNet nets have it: the golden test above
    compares six tensors with the reference's fp32 run at 1e-2 L2; here HIP must be as close to float64 as the CPU fp32 oracle or the
    1e-6-perturbation envelope allow (factor 2), and within 2e-3 absolutely."""
    from arbiter import Verdicts, double_copy, perturb_, perturb_tensor
    from models.depth.disp_net import DispNetS
    from oracle import nets as on
    hip = reinit_by_name(DispNetS(), 51)
    ref = on.DispNetS()
    ref.load_state_dict(hip.state_dict())
    hip.to(DEV).train()
    ref.train()
    g = torch.Generator().manual_seed(53)
    x = torch.randn(2, 3, 64, 128, generator=g)
    want = ref(x)
    coefs = [torch.randn(o.shape, generator=g) for o in want]
    sum((o * c).sum() for o, c in zip(want, coefs)).backward()
    got = hip(x.to(DEV))
    sum((o * c.to(DEV)).sum() for o, c in zip(got, coefs)).backward()

    def run64(net, xin):
        net.zero_grad()
        outs = net(xin)
        sum((o * c.double()).sum() for o, c in zip(outs, coefs)).backward()
        return outs, dict(net.named_parameters())
    out64, r64 = run64(double_copy(ref), x.double())
    envs = [run64(perturb_(double_copy(ref), 1e-6, 910 + e), perturb_tensor(x.double(), 1e-6, 960 + e))[1] for e in range(2)]
    v = Verdicts(floor=2.5e-4)
    for i, (a, b, c) in enumerate(zip(got, want, out64)):
        v.add("disp%d" % (i + 1), a, b, c)
    rp = dict(ref.named_parameters())
    for n, q in hip.named_parameters():
        if rp[n].grad is not None:
            assert q.grad is not None, n
            v.add(n, q.grad, rp[n].grad, r64[n].grad, [env[n].grad for env in envs])
    v.check("test_dispnets_gradients_vs_fp64_arbiter", hip_abs=2e-3)


def test_posefc_vs_reference(golden):
    from models.pose.pose_fc import PoseFc
    g = golden("posefc.npz")
    m = reinit_by_name(PoseFc(), 61).to(DEV)
    gen = torch.Generator().manual_seed(62)
    tgt, r0, r1 = (torch.randn(1, 3, 384, 1280, generator=gen) for _ in range(3))
    out = m(tgt.to(DEV), [r0.to(DEV), r1.to(DEV)])
    assert rel_err(out, g["out"]) < 1e-3
    assert float(out[:, :, :3].abs().max()) == 0.0
    (out * T(g["coef"])).sum().backward()
    grads_match(m, g, ["fc_loc.0.weight", "fc_loc.4.weight", "pose_pred.bias", "conv7.0.bias"], 5e-3)
    with pytest.raises(Exception):
        m(tgt[:, :, :192, :640].contiguous().to(DEV), [r0[:, :, :192, :640].contiguous().to(DEV), r1[:, :, :192, :640].contiguous().to(DEV)])


@pytest.mark.parametrize("B,H,W,gamma3,hip_abs", [(4, 64, 128, None, 1e-1)])
def test_dispresnet50_vs_oracle(B, H, W, gamma3, hip_abs):
    """ResNet-50 encoder (Bottleneck blocks) + decoder with the x4 channel widths (BASELINE.json configs[3]) against the CPU oracle.
    The absolute 2e-3 pin of the ResNet-50 backward lives in test_step_gpu.py (the damped batch-8 whole step, where the CPU fp32 oracle sits
    3e-6 from float64 on the median tensor).  This network-only problem -- a white-noise upstream on the disparity -- stays ill-conditioned
    damped or not: stock PyTorch fp32 itself is 1e-2 from float64 on its worst tensor and 2.5e-3 on the median one (measured on the CPU
    for batch 8, 96x160, gamma3 0.2, three upstream / input variants), so only the relative rule can decide here."""
    from models.depth.resnet_dispnet import DispResNet50
    from oracle import nets as on
    from test_step_gpu import damp_residual_branches
    hip = reinit_by_name(DispResNet50(), 77)
    if gamma3 is not None:
        damp_residual_branches(hip, gamma3)
    ref = on.DispResNet(50)
    ref.load_state_dict(hip.state_dict())
    hip.to(DEV).train()
    ref.train()
    g = torch.Generator().manual_seed(78)
    x = torch.randn(B, 3, H, W, generator=g)          # batch 4: 32 samples per channel in layer4's BatchNorm (2x4 maps)
    coef = torch.randn(B, 1, H, W, generator=g)
    want = ref(x)[0]
    (want * coef).sum().backward()
    got = hip(x.to(DEV))[0]
    assert rel_err(got, want) < 1e-3
    (got * coef.to(DEV)).sum().backward()
    rp = dict(ref.named_parameters())
    # gradients: the fp64 arbiter (tests/arbiter.py) -- the same network in float64, as it is and on 1e-6-perturbed weights / input
    from arbiter import Verdicts, double_copy, perturb_, perturb_tensor
    ref64 = double_copy(ref)
    ref64.zero_grad()
    out64 = ref64(x.double())[0]
    (out64 * coef.double()).sum().backward()
    r64 = dict(ref64.named_parameters())
    envs = []
    for e in range(2):
        re_ = perturb_(double_copy(ref), 1e-6, 900 + e)
        re_.zero_grad()
        (re_(perturb_tensor(x.double(), 1e-6, 950 + e))[0] * coef.double()).sum().backward()
        envs.append(dict(re_.named_parameters()))
    v = Verdicts(floor=2.5e-4)
    v.add("disparity", got, want, out64)
    for n, p in hip.named_parameters():
        if rp[n].grad is not None:
            v.add(n, p.grad, rp[n].grad, r64[n].grad, [env[n].grad for env in envs])
    v.check("test_dispresnet50_vs_oracle[gamma3=%s]" % gamma3, hip_abs=hip_abs)      # gamma ~ 1: ill-conditioned (envelope up to 2e-2), the relative rule decides
    # the stacked two-pass form gives the same disparities
    hip2 = reinit_by_name(DispResNet50(), 77)
    if gamma3 is not None:
        damp_residual_branches(hip2, gamma3)
    hip2.to(DEV).train()
    a, b = hip2.forward_pair(x.to(DEV), x.flip(0).contiguous().to(DEV))
    assert rel_err(a[0], want) < 1e-3


def test_standalone_upsample():
    """`upsample` of the reference's call surface (models/depth/layers.py:55-58: nearest x2 on NCHW), forward and backward."""
    import torch.nn.functional as F
    from models.depth.layers import upsample
    g = torch.Generator().manual_seed(21)
    for shape in ((2, 5, 7, 9), (1, 16, 24, 80)):
        x = torch.randn(*shape, generator=g)
        coef = torch.randn(shape[0], shape[1], 2 * shape[2], 2 * shape[3], generator=g)
        xr = x.clone().requires_grad_()
        want = F.interpolate(xr, scale_factor=2, mode="nearest")
        (want * coef).sum().backward()
        xd = x.to(DEV).requires_grad_()
        got = upsample(xd)
        assert torch.equal(got.detach().cpu(), want.detach())
        (got * coef.to(DEV)).sum().backward()
        assert rel_err(xd.grad, xr.grad) < 1e-6
