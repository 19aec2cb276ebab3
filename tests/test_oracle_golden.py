"""CPU: the oracle (oracle/) against golden vectors captured from the reference's own modules."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from seeding import reinit_by_name
from oracle import geometry as og
from oracle import losses as ol
from oracle import nets as on

T = torch.from_numpy


def digest(model):
    s = a = 0.0
    for p in model.state_dict().values():
        if p.dtype.is_floating_point:
            s += float(p.double().sum())
            a += float(p.double().abs().sum())
    return np.array([s, a])


def check_digest(model, want):
    got = digest(model)
    assert np.allclose(got, want, rtol=1e-9, atol=1e-9), "weights differ from the ones that generated the fixture"


def test_loss_small(golden):
    g = golden("loss_small.npz")
    disp_t = T(g["disp_t"]).requires_grad_()
    disp_r = T(g["disp_r"]).requires_grad_()
    poses = T(g["poses"]).requires_grad_()
    out = ol.losses_forward(T(g["tgt"]), [T(g["ref0"]), T(g["ref1"])], [[disp_t], [disp_r]], poses, T(g["K"]))
    assert abs(float(out[0]) - g["loss"][0]) < 1e-6 * abs(g["loss"][0])
    assert abs(float(out[1]) - g["loss"][1]) < 1e-6 * abs(g["loss"][1])
    sum(out).backward()
    assert rel_err(disp_t.grad, g["g_disp_t"]) < 1e-5
    assert rel_err(disp_r.grad, g["g_disp_r"]) < 1e-5
    assert rel_err(poses.grad, g["g_poses"]) < 1e-5


def test_warp_pieces(golden):
    g = golden("loss_small.npz")
    K = T(g["K"])
    Dt = T(g["depth_t"])
    p = T(g["poses"])
    assert rel_err(og.disp_to_depth([[T(g["disp_t"])]])[0][0][:, 0], Dt) < 1e-7
    assert rel_err(og.reconstruct(Dt, K), g["cam_points"]) < 1e-6
    assert rel_err(og.pose_to_matrix(p[:, 0]), g["Tcw0"]) < 1e-6
    assert rel_err(og.pose_to_matrix(p[:, 0], invert=True), g["Tcw0_inv"]) < 1e-6
    assert rel_err(og.project(og.reconstruct(Dt, K), K, og.pose_to_matrix(p[:, 0])), g["grid0"]) < 1e-6
    Dr = og.disp_to_depth([[T(g["disp_r"])]])[0][0][:, 0]
    assert rel_err(og.inverse_warp(T(g["ref0"]), Dt, p[:, 0], K, False), g["warp0"]) < 1e-5
    assert rel_err(og.inverse_warp(T(g["ref1"]), Dt, p[:, 1], K, False), g["warp1"]) < 1e-5
    assert rel_err(og.inverse_warp(T(g["tgt"]), Dr, p[:, 0], K, True), g["warp2"]) < 1e-5


def test_loss_ka1(golden):
    g = golden("loss_ka1.npz")
    torch.manual_seed(0)
    B, H, W = 4, 64, 128
    K = torch.tensor([[0.58 * W, 0, 0.5 * W], [0, 1.92 * H, 0.5 * H], [0, 0, 1]], dtype=torch.float64).repeat(B, 1, 1)
    tgt = torch.randn(B, 3, H, W)
    refs = [torch.randn(B, 3, H, W), torch.randn(B, 3, H, W)]
    disp_t = torch.rand(B, 1, H, W).requires_grad_()
    disp_r = torch.rand(B, 1, H, W).requires_grad_()
    poses = (0.01 * torch.randn(B, 2, 6)).requires_grad_()
    dig = np.array([float(tgt.double().sum()), float(refs[1].double().sum()), float(disp_r.double().sum()),
                    float(poses.double().sum())])
    if not np.allclose(dig, g["input_digest"], rtol=1e-9):
        pytest.skip("torch RNG stream differs from the one that generated the fixture")
    out = ol.losses_forward(tgt, refs, [[disp_t], [disp_r]], poses, K)
    # [0.9413443804, 6.7181606293]
    assert abs(float(out[0]) - 0.9413443804) < 2e-6 and abs(float(out[1]) - 6.7181606293) < 2e-5
    assert np.allclose([float(out[0]), float(out[1])], g["loss"], rtol=1e-6)
    sum(out).backward()
    norms = [float(disp_t.grad.norm()), float(disp_r.grad.norm()), float(poses.grad.norm())]
    assert np.allclose(norms, g["grad_norms"], rtol=1e-4)
    assert rel_err(poses.grad, g["g_poses"]) < 1e-4
    assert rel_err(disp_t.grad[1, 0, 17], g["g_disp_t_row"]) < 1e-4


def test_loss_ssim(golden):
    """SSIM + L1 photometric mix: the oracle against the composition of the reference's own inverse_warp / SSIM / smooth_loss."""
    g = golden("loss_ssim.npz")
    disp_t, disp_r, poses = T(g["disp_t"]).requires_grad_(), T(g["disp_r"]).requires_grad_(), T(g["poses"]).requires_grad_()
    out = ol.losses_forward(T(g["tgt"]), [T(g["ref0"]), T(g["ref1"])], [[disp_t], [disp_r]], poses, T(g["K"]), ssim_weight=0.85)
    assert np.allclose([float(out[0]), float(out[1])], g["loss"], rtol=1e-6)
    sum(out).backward()
    assert rel_err(disp_t.grad, g["g_disp_t"]) < 1e-5 and rel_err(disp_r.grad, g["g_disp_r"]) < 1e-5
    assert rel_err(poses.grad, g["g_poses"]) < 1e-5


def test_warp_edge(golden):
    g = golden("warp_edge.npz")
    img, K = T(g["img"]), T(g["K"])
    B, _, H, W = img.shape
    w_id = og.inverse_warp(img, torch.full((B, H, W), 5.0), torch.zeros(B, 6), K, False)
    assert rel_err(w_id, g["warp_identity"]) < 1e-6
    # identity pose is identity only up to the +1e-5/+1e-7 guards
    assert float((w_id - img).abs().max()) < 5e-3
    depth, big = T(g["depth"]), T(g["pose_big"])
    assert rel_err(og.inverse_warp(img, depth, big, K, False), g["warp_big"]) < 1e-4
    assert rel_err(og.inverse_warp(img, depth, big, K, True), g["warp_big_inv"]) < 1e-4


def test_ssim(golden):
    g = golden("ssim.npz")
    x, y = T(g["x"]), T(g["y"])
    assert rel_err(ol.ssim_distance(x, y), g["ssim"]) < 1e-6
    assert float(ol.ssim_distance(x, x).max()) == float(g["ssim_self_max"]) == 0.0
    torch.manual_seed(1)
    x3, y3 = torch.rand(4, 3, 64, 128), torch.rand(4, 3, 64, 128)
    s3 = ol.ssim_distance(x3, y3)
    # mean 0.4965941, min 0.0145527, max 0.9818020
    assert np.allclose([float(s3.mean()), float(s3.min()), float(s3.max())], g["ka3"], rtol=1e-5)
    assert abs(float(s3.mean()) - 0.4965941) < 1e-5


def test_smooth(golden):
    g = golden("smooth.npz")
    d0 = T(g["disp0"]).requires_grad_()
    d1 = T(g["disp1"]).requires_grad_()
    depth = og.disp_to_depth([[d0, d1]])[0]
    assert rel_err(depth[0], g["depth0"]) < 1e-7
    loss = ol.smooth_loss(depth)
    assert abs(float(loss) - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    loss.backward()
    assert rel_err(d0.grad, g["g_disp0"]) < 1e-5
    assert rel_err(d1.grad, g["g_disp1"]) < 1e-5


def grads_match(model, g, names, tol=2e-4):
    sd = dict(model.named_parameters())
    for n in names:
        gr = sd[n].grad
        if gr.numel() > 65536:
            gr = gr[:8, :8]
        assert rel_err(gr, g["g_" + n.replace(".", "_")]) < tol, n


def test_posenet(golden):
    g = golden("posenet.npz")
    m = reinit_by_name(on.PoseNet(), 21)
    check_digest(m, g["digest"])
    out = m(T(g["tgt"]), [T(g["ref0"]), T(g["ref1"])])
    assert rel_err(out, g["out"]) < 1e-5
    (out * T(g["coef"])).sum().backward()
    grads_match(m, g, ["conv1.0.weight", "conv1.0.bias", "conv4.0.weight", "conv7.0.bias", "pose_pred.weight", "pose_pred.bias"])


def test_decoder(golden):
    g = golden("decoder.npz")
    dec = reinit_by_name(on.DepthDecoder(np.array([64, 64, 128, 256, 512])), 31)
    check_digest(dec, g["digest"])
    feats = [T(g["f%d" % i]).requires_grad_() for i in range(5)]
    out = dec(feats)
    for s in range(4):
        assert rel_err(out[("disp", s)], g["disp%d" % s]) < 1e-5
    (out[("disp", 0)] * T(g["coef"])).sum().backward()
    for i in range(5):
        assert rel_err(feats[i].grad, g["g_f%d" % i]) < 2e-4
    grads_match(dec, g, ["decoder.0.conv.conv.weight", "decoder.0.conv.conv.bias", "decoder.7.conv.conv.weight",
                         "decoder.9.conv.conv.weight", "decoder.9.conv.conv.bias", "decoder.10.conv.weight",
                         "decoder.10.conv.bias"])


def test_dispresnet(golden):
    g = golden("dispresnet.npz")
    m = reinit_by_name(on.DispResNet(), 41)
    m.train()
    check_digest(m, g["digest"])
    out = m(T(g["x"]))[0]
    assert rel_err(out, g["disp"]) < 1e-5
    assert rel_err(m.encoder.encoder.bn1.running_mean, g["running_mean_bn1"]) < 1e-5
    assert rel_err(m.encoder.encoder.bn1.running_var, g["running_var_bn1"]) < 1e-5
    (out * T(g["coef"])).sum().backward()
    grads_match(m, g, ["encoder.encoder.conv1.weight", "encoder.encoder.bn1.weight", "encoder.encoder.bn1.bias",
                       "encoder.encoder.layer2.0.downsample.0.weight", "encoder.encoder.layer4.1.bn2.weight",
                       "decoder.decoder.0.conv.conv.bias", "decoder.decoder.10.conv.weight"], tol=1e-3)


def test_dispnets(golden):
    g = golden("dispnets.npz")
    m = reinit_by_name(on.DispNetS(), 51)
    m.train()
    check_digest(m, g["digest"])
    outs = m(T(g["x"]))
    for i, o in enumerate(outs):
        assert rel_err(o, g["disp%d" % (i + 1)]) < 1e-5
    sum((o * T(g["coef%d" % (i + 1)])).sum() for i, o in enumerate(outs)).backward()
    grads_match(m, g, ["conv1.0.weight", "conv1.2.weight", "upconv7.0.weight", "upconv1.0.bias", "iconv3.0.weight",
                       "predict_disp1.0.weight"], tol=1e-3)


def test_posefc(golden):
    g = golden("posefc.npz")
    m = reinit_by_name(on.PoseFc(), 61)
    gen = torch.Generator().manual_seed(62)
    check_digest(m, g["digest"])
    tgt, r0, r1 = (torch.randn(1, 3, 384, 1280, generator=gen) for _ in range(3))
    out = m(tgt, [r0, r1])
    assert rel_err(out, g["out"]) < 1e-5
    assert float(out[:, :, :3].abs().max()) == 0.0
    coef = torch.randn(1, 2, 6, generator=gen)
    assert rel_err(coef, g["coef"]) == 0
    (out * coef).sum().backward()
    grads_match(m, g, ["fc_loc.0.weight", "fc_loc.4.weight", "pose_pred.bias", "conv7.0.bias"], tol=1e-3)


def test_state_dict_keys():
    want = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    models = {"PoseNet": on.PoseNet(), "DepthDecoder": on.DepthDecoder(np.array([64, 64, 128, 256, 512])),
              "DispResNet": on.DispResNet(), "DispNetS": on.DispNetS(), "PoseFc": on.PoseFc()}
    for name, m in models.items():
        got = {k: list(v.shape) for k, v in m.state_dict().items()}
        assert got == want[name], name


def test_metrics_oracle_vs_reference():
    """oracle.evaluate.compute_errors == the reference's compute_errors (evaluate.py:6-39, quirk 'sq_rel' = rms included)."""
    from oracle import evaluate as oe
    g = np.load(os.path.join(GOLDEN, "metrics.npz"))
    got = oe.compute_errors(g["gt"], g["disp"])
    for k in ("silog", "abs_rel", "log10", "rms", "sq_rel", "log_rms", "d1", "d2", "d3"):
        assert abs(float(got[k]) - float(g[k])) <= 1e-6 * max(1.0, abs(float(g[k]))), k
    assert got["sq_rel"] == got["rms"] and got["sq_rel_fixed"] != got["rms"]


def test_pseudo_lidar_oracle_vs_reference():
    """oracle.pseudo_lidar.project_PL == PseudoLiDAR.project_PL (pseudo-lidar/utils/PseudoLiDAR.py:69-110), bit for bit (same numpy ops)."""
    from oracle import pseudo_lidar as op
    g = np.load(os.path.join(GOLDEN, "pseudo_lidar.npz"))
    for name in "abc":
        got = op.project_PL(g["depth_" + name], g["T"], g["P_" + name], int(g["sparsity_" + name]))
        assert got.shape == g["cloud_" + name].shape and got.dtype == np.float64
        assert np.array_equal(got, g["cloud_" + name])
        assert np.all(got[:, 3] == 0.0)          # the reference's inverse_rigid_trans leaves the homogeneous row zero


def test_preprocess_oracle_vs_pillow_golden():
    """oracle.preprocess == the reference's transform chain run with the real Pillow (tests/golden/preprocess.npz), bit for bit; and the
    C ABI's host-side coefficient tables (mcav_resample_coeffs) == the oracle's (Pillow's precompute_coeffs + normalize_coeffs_8bpc)."""
    import ctypes
    from oracle import preprocess as op
    g = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    for name in ("down", "up", "mixed", "kitti"):
        img, want_small, want = g["img_" + name], g["resized_" + name], g["out_" + name]
        h, w = want_small.shape[:2]
        assert np.array_equal(op.byte_quirk(img), img)                       # float32(v)/255*255 truncates back to v for every byte
        assert np.array_equal(op.pil_resize_bilinear_u8(img, h, w), want_small)
        assert np.array_equal(op.load_transform(img, h, w), want)
    import mcav.lib as L
    import dataloaders  # noqa: F401  (registers the signatures)
    hnd = L.lib()
    for in_size, out_size in ((248, 128), (33, 80), (37, 37), (1242, 640), (375, 192), (7, 2)):
        ks, bounds, kk = op.resample_coeffs(in_size, out_size)
        c_ks = ctypes.c_int(0)
        cap = hnd.mcav_resample_coeffs(in_size, out_size, ctypes.byref(c_ks), None, None, 0)
        assert c_ks.value == ks and cap == out_size * ks
        b = np.zeros(out_size * 2, np.int32)
        k = np.zeros(cap, np.int32)
        assert hnd.mcav_resample_coeffs(in_size, out_size, ctypes.byref(c_ks), b.ctypes.data_as(ctypes.c_void_p), k.ctypes.data_as(ctypes.c_void_p), cap) == 0
        assert np.array_equal(b.reshape(-1, 2), bounds) and np.array_equal(k.reshape(out_size, ks), kk)


def test_preprocess_oracle_vs_installed_pillow():
    """Live check against the Pillow in this environment (skipped where it is absent): random sizes, both directions."""
    Image = pytest.importorskip("PIL.Image")
    from oracle import preprocess as op
    rng = np.random.RandomState(4)
    for (h0, w0, h, w) in [(61, 97, 24, 80), (30, 30, 64, 31), (375, 1242, 192, 640), (9, 5, 9, 11)]:
        img = rng.randint(0, 256, (h0, w0, 3)).astype(np.uint8)
        want = np.asarray(Image.fromarray(img).resize((w, h), Image.BILINEAR))
        assert np.array_equal(op.pil_resize_bilinear_u8(img, h, w), want)
