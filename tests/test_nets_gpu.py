"""GPU: the HIP networks (reference call surface) against golden vectors from the reference modules."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from seeding import reinit_by_name

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def grads_match(model, g, names, tol):
    sd = dict(model.named_parameters())
    for n in names:
        gr = sd[n].grad
        assert gr is not None, n
        if gr.numel() > 65536:
            gr = gr[:8, :8]
        assert rel_err(gr, g["g_" + n.replace(".", "_")]) < tol, n


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
