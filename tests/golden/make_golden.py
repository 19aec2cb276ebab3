#!/usr/bin/env python3
"""Generate golden vectors by importing the REFERENCE's own modules (build container only).

Run from the repo root:   python tests/golden/make_golden.py
Reads  /root/reference (never copied, never shipped).  Writes small .npz/.json fixtures next to this
file.  The fixtures are data only: seeded inputs and the reference's outputs / gradients.

Harness-side shims (the reference files are untouched):
  * sys.modules['cv2']            empty module  (cv2 is imported but unused: losses.py:6, transform.py:9)
  * torch.Tensor.cuda             identity      (transform.py:134 hard-codes .cuda())
  * sys.modules['torchvision(.models)']  stub whose resnetNN attributes build oracle.nets.ResNet
    (torchvision is not installed; only dereferenced in resnet_dispnet.py:20-30)
The reference's Transform.k_hom repeats K 4 times (transform.py:110), so every loss/warp case uses B=4.
"""
import contextlib
import io
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

warnings.filterwarnings("ignore")
sys.path.insert(0, REPO)
from oracle import nets as onets  # noqa: E402  (only used as the torchvision-resnet stand-in)
sys.path.insert(0, HERE)
from seeding import reinit_by_name  # noqa: E402

sys.modules["cv2"] = types.ModuleType("cv2")
torch.Tensor.cuda = lambda self, *a, **k: self
tv = types.ModuleType("torchvision")
tvm = types.ModuleType("torchvision.models")
for n in (18, 34, 50, 101, 152):
    setattr(tvm, "resnet%d" % n, onets.resnet_factory(n))
tv.models = tvm
sys.modules["torchvision"] = tv
sys.modules["torchvision.models"] = tvm
sys.path.insert(0, REF)

import losses as ref_losses  # noqa: E402
from geometry import pose_geometry as ref_pg  # noqa: E402
from geometry.transform import Transform as RefTransform  # noqa: E402
from models.depth import resnet_dispnet as ref_rd  # noqa: E402
from models.depth import disp_net as ref_dn  # noqa: E402
from models.pose import pose_net as ref_pn  # noqa: E402
from models.pose import pose_fc as ref_pf  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def npy(t):
    return t.detach().cpu().numpy()


def kmat(B, H, W):
    K = torch.tensor([[0.58 * W, 0.0, 0.5 * W], [0.0, 1.92 * H, 0.5 * H], [0.0, 0.0, 1.0]], dtype=torch.float64)
    return K.repeat(B, 1, 1)


def smooth_images(g, B, H, W):
    x = torch.randn(B, 3, H, W, generator=g)
    return torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect"), 3, 1).contiguous()


def weight_digest(model):
    s = 0.0
    a = 0.0
    for p in model.state_dict().values():
        if p.dtype.is_floating_point:
            s += float(p.double().sum())
            a += float(p.double().abs().sum())
    return np.array([s, a])


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024))


# ------------------------------------------------------------------ 1. loss, explicit small case
def case_loss_small():
    g = torch.Generator().manual_seed(11)
    B, H, W = 4, 24, 48
    K = kmat(B, H, W)
    tgt, r0, r1 = (smooth_images(g, B, H, W) for _ in range(3))
    disp_t = torch.rand(B, 1, H, W, generator=g).requires_grad_()
    disp_r = torch.rand(B, 1, H, W, generator=g).requires_grad_()
    poses = (0.02 * torch.randn(B, 2, 6, generator=g)).requires_grad_()
    L = ref_losses.Losses()
    out = quiet(L.forward, tgt, [r0, r1], [[disp_t], [disp_r]], poses, K, None)
    sum(out).backward()
    arrays = dict(tgt=npy(tgt), ref0=npy(r0), ref1=npy(r1), disp_t=npy(disp_t), disp_r=npy(disp_r), poses=npy(poses),
                  K=npy(K), loss=np.array([float(out[0]), float(out[1])]), g_disp_t=npy(disp_t.grad),
                  g_disp_r=npy(disp_r.grad), g_poses=npy(poses.grad))
    # per-loss gradients (upstream (1,0) and (0,1)) so the HIP path's general-weight branch is pinned too
    for tag, w in (("mam", (1.0, 0.0)), ("smooth", (0.0, 1.0))):
        for t in (disp_t, disp_r, poses):
            t.grad = None
        out = quiet(L.forward, tgt, [r0, r1], [[disp_t], [disp_r]], poses, K, None)
        (w[0] * out[0] + w[1] * out[1]).backward()
        arrays["g_disp_t_" + tag] = npy(disp_t.grad)
        arrays["g_poses_" + tag] = npy(poses.grad) if poses.grad is not None else np.zeros((B, 2, 6), np.float32)
    # the three warped images + sampling grids, for the standalone inverse_warp / Transform API
    depths = ref_pg.disp_to_depth([[disp_t.detach()], [disp_r.detach()]])
    Dt, Dr = depths[0][0].squeeze(), depths[1][0].squeeze()
    p = poses.detach()
    arrays["depth_t"] = npy(Dt)
    arrays["warp0"] = npy(ref_pg.inverse_warp(r0, Dt, p[:, 0], K, False))
    arrays["warp1"] = npy(ref_pg.inverse_warp(r1, Dt, p[:, 1], K, False))
    arrays["warp2"] = npy(ref_pg.inverse_warp(tgt, Dr, p[:, 0], K, True))
    tr = RefTransform()
    Xc = tr.reconstruct(Dt, K)
    Tcw = ref_pg.transformation_from_parameters(p[:, 0, :3].unsqueeze(1), p[:, 0, 3:].unsqueeze(1))
    arrays["cam_points"] = npy(Xc)
    arrays["Tcw0"] = npy(Tcw)
    arrays["Tcw0_inv"] = npy(ref_pg.invert_pose(Tcw))
    arrays["grid0"] = npy(tr.project(Xc, K, Tcw))
    save("loss_small.npz", **arrays)


# ------------------------------------------------------------------ 2. loss KA1 (seed-regenerated inputs)
def case_loss_ka1():
    torch.manual_seed(0)
    B, H, W = 4, 64, 128
    K = kmat(B, H, W)
    tgt = torch.randn(B, 3, H, W)
    refs = [torch.randn(B, 3, H, W), torch.randn(B, 3, H, W)]
    disp_t = torch.rand(B, 1, H, W).requires_grad_()
    disp_r = torch.rand(B, 1, H, W).requires_grad_()
    poses = (0.01 * torch.randn(B, 2, 6)).requires_grad_()
    out = quiet(ref_losses.Losses().forward, tgt, refs, [[disp_t], [disp_r]], poses, K, None)
    sum(out).backward()
    save("loss_ka1.npz", loss=np.array([float(out[0]), float(out[1])]),
         grad_norms=np.array([float(disp_t.grad.norm()), float(disp_r.grad.norm()), float(poses.grad.norm())]),
         g_poses=npy(poses.grad), g_disp_t_row=npy(disp_t.grad[1, 0, 17]), g_disp_r_row=npy(disp_r.grad[2, 0, 40]),
         input_digest=np.array([float(tgt.double().sum()), float(refs[1].double().sum()), float(disp_r.double().sum()),
                                float(poses.double().sum())]))


# ------------------------------------------------------------------ 3. identity-pose warp, large-motion warp
def case_warp_edge():
    g = torch.Generator().manual_seed(5)
    B, H, W = 4, 20, 40
    K = kmat(B, H, W)
    img = torch.randn(B, 3, H, W, generator=g)
    depth5 = torch.full((B, H, W), 5.0)
    ident = torch.zeros(B, 6)
    w_id = ref_pg.inverse_warp(img, depth5, ident, K, False)
    # big motion: many samples fall outside the image (zeros padding) and behind-camera-ish geometry
    depth = 0.5 + 20 * torch.rand(B, H, W, generator=g)
    big = torch.tensor([[0.3, -0.2, 0.1, 2.0, -1.0, 0.5], [-0.4, 0.5, 0.2, -3.0, 0.5, -0.2],
                        [0.05, 0.02, -0.6, 0.1, 4.0, 0.3], [0.0, 0.0, 0.0, 0.0, 0.0, -0.45]])
    w_big = ref_pg.inverse_warp(img, depth, big, K, False)
    w_big_inv = ref_pg.inverse_warp(img, depth, big, K, True)
    save("warp_edge.npz", img=npy(img), K=npy(K), depth=npy(depth), pose_big=npy(big), warp_identity=npy(w_id),
         warp_big=npy(w_big), warp_big_inv=npy(w_big_inv))


# ------------------------------------------------------------------ 4. SSIM (+ KA3 scalars)
def case_ssim():
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 16, 24, generator=g)
    y = torch.rand(2, 3, 16, 24, generator=g)
    s = ref_losses.SSIM().standard_loss(x, y)
    torch.manual_seed(1)
    x3, y3 = torch.rand(4, 3, 64, 128), torch.rand(4, 3, 64, 128)
    s3 = ref_losses.SSIM().standard_loss(x3, y3)
    save("ssim.npz", x=npy(x), y=npy(y), ssim=npy(s), ssim_self_max=np.array(float(ref_losses.SSIM().standard_loss(x, x).max())),
         ka3=np.array([float(s3.mean()), float(s3.min()), float(s3.max())]))


# ------------------------------------------------------------------ 4b. SSIM + L1 photometric mix over the live path's three warps
def case_loss_ssim():
    """The reference cannot run its SSIM photometric loss end to end (self.SSIM is commented out, losses.py:59), so this case composes
    the reference's OWN pieces the way its live path composes the L1 terms: ref inverse_warp (pose_geometry.py:201-229), ref
    SSIM.standard_loss (losses.py:12-54), the 0.85 / 0.15 weights of losses.py:77, the three warps and means of losses.py:183-240."""
    g = torch.Generator().manual_seed(21)
    B, H, W = 4, 40, 72                        # ragged against the kernel's 32 x 32 tiles
    K = kmat(B, H, W)
    tgt, r0, r1 = (smooth_images(g, B, H, W) for _ in range(3))
    disp_t = torch.rand(B, 1, H, W, generator=g).requires_grad_()
    disp_r = torch.rand(B, 1, H, W, generator=g).requires_grad_()
    poses = (0.02 * torch.randn(B, 2, 6, generator=g)).requires_grad_()
    depths = ref_pg.disp_to_depth([[disp_t], [disp_r]])
    ssim = ref_losses.SSIM()

    def photo(pred, target):
        return (0.85 * ssim.standard_loss(pred, target) + 0.15 * torch.abs(target - pred)).mean()

    Dt, Dr = depths[0][0][:, 0], depths[1][0][:, 0]
    w0 = ref_pg.inverse_warp(r0, Dt, poses[:, 0], K, False)
    w1 = ref_pg.inverse_warp(r1, Dt, poses[:, 1], K, False)
    w2 = ref_pg.inverse_warp(tgt, Dr, poses[:, 0], K, True)
    loss_mam = (torch.stack([photo(w0, tgt), photo(w1, tgt)]).mean() + torch.stack([photo(w2, r1)]).mean()) / 2
    loss_smooth = ref_losses.Losses().smooth_loss(depths[0])
    (loss_mam + loss_smooth).backward()
    save("loss_ssim.npz", tgt=npy(tgt), ref0=npy(r0), ref1=npy(r1), disp_t=npy(disp_t), disp_r=npy(disp_r), poses=npy(poses), K=npy(K),
         loss=np.array([float(loss_mam), float(loss_smooth)]), g_disp_t=npy(disp_t.grad), g_disp_r=npy(disp_r.grad), g_poses=npy(poses.grad))


# ------------------------------------------------------------------ 5. smoothness + disp_to_depth
def case_smooth():
    g = torch.Generator().manual_seed(8)
    d = [torch.rand(3, 1, 16, 20, generator=g).requires_grad_(), torch.rand(3, 1, 8, 10, generator=g).requires_grad_()]
    depth = ref_pg.disp_to_depth([d])[0]
    loss = ref_losses.Losses().smooth_loss(depth)
    loss.backward()
    save("smooth.npz", disp0=npy(d[0]), disp1=npy(d[1]), depth0=npy(depth[0]), loss=np.array(float(loss)),
         g_disp0=npy(d[0].grad), g_disp1=npy(d[1].grad))


# ------------------------------------------------------------------ 6. networks (seed-regenerated weights)
def grads_of(model, names):
    """Selected parameter gradients; tensors above 64K elements are stored as their [:8, :8] corner block."""
    sd = dict(model.named_parameters())
    out = {}
    for n in names:
        gr = sd[n].grad
        if gr.numel() > 65536:
            gr = gr[:8, :8]
        out["g_" + n.replace(".", "_")] = npy(gr)
    return out


def case_posenet():
    m = reinit_by_name(ref_pn.PoseNet(), 21)
    g = torch.Generator().manual_seed(22)
    tgt, r0, r1 = (torch.randn(2, 3, 64, 128, generator=g) for _ in range(3))
    out = m(tgt, [r0, r1])
    coef = torch.randn(2, 2, 6, generator=g)
    (out * coef).sum().backward()
    save("posenet.npz", tgt=npy(tgt), ref0=npy(r0), ref1=npy(r1), coef=npy(coef), out=npy(out), digest=weight_digest(m),
         **grads_of(m, ["conv1.0.weight", "conv1.0.bias", "conv4.0.weight", "conv7.0.bias", "pose_pred.weight", "pose_pred.bias"]))
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def case_decoder():
    """Reference DepthDecoder alone, on seeded feature maps (R18 channel set), 64x128 image geometry."""
    dec = reinit_by_name(ref_rd.DepthDecoder(np.array([64, 64, 128, 256, 512])), 31)
    g = torch.Generator().manual_seed(32)
    B = 2
    shapes = [(64, 32, 64), (64, 16, 32), (128, 8, 16), (256, 4, 8), (512, 2, 4)]
    feats = [torch.randn(B, c, h, w, generator=g).requires_grad_() for c, h, w in shapes]
    out = dec(feats)
    coef = torch.randn(B, 1, 64, 128, generator=g)
    (out[("disp", 0)] * coef).sum().backward()
    arrays = {"f%d" % i: npy(f) for i, f in enumerate(feats)}
    arrays.update({"g_f%d" % i: npy(f.grad) for i, f in enumerate(feats)})
    arrays.update({"disp%d" % s: npy(out[("disp", s)]) for s in range(4)})
    arrays.update(grads_of(dec, ["decoder.0.conv.conv.weight", "decoder.0.conv.conv.bias", "decoder.7.conv.conv.weight",
                                 "decoder.9.conv.conv.weight", "decoder.9.conv.conv.bias", "decoder.10.conv.weight",
                                 "decoder.10.conv.bias"]))
    save("decoder.npz", coef=npy(coef), digest=weight_digest(dec), **arrays)
    return {k: list(v.shape) for k, v in dec.state_dict().items()}


def case_dispresnet():
    """Reference DispResNet wiring (encoder glue + decoder) around the oracle's ResNet-18 trunk (hybrid)."""
    m = reinit_by_name(ref_rd.DispResNet(), 41)
    m.train()
    dig = weight_digest(m)
    g = torch.Generator().manual_seed(42)
    x = torch.randn(2, 3, 64, 128, generator=g)
    out = m(x)[0]
    coef = torch.randn(2, 1, 64, 128, generator=g)
    (out * coef).sum().backward()
    save("dispresnet.npz", x=npy(x), coef=npy(coef), disp=npy(out), digest=dig,
         running_mean_bn1=npy(m.encoder.encoder.bn1.running_mean), running_var_bn1=npy(m.encoder.encoder.bn1.running_var),
         **grads_of(m, ["encoder.encoder.conv1.weight", "encoder.encoder.bn1.weight", "encoder.encoder.bn1.bias",
                        "encoder.encoder.layer2.0.downsample.0.weight", "encoder.encoder.layer4.1.bn2.weight",
                        "decoder.decoder.0.conv.conv.bias", "decoder.decoder.10.conv.weight"]))
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def case_dispnets():
    m = reinit_by_name(ref_dn.DispNetS(), 51)
    m.train()
    dig = weight_digest(m)
    g = torch.Generator().manual_seed(52)
    x = torch.randn(2, 3, 64, 128, generator=g)
    outs = m(x)
    coef = [torch.randn(o.shape, generator=g) for o in outs]
    sum((o * c).sum() for o, c in zip(outs, coef)).backward()
    arrays = {"disp%d" % (i + 1): npy(o) for i, o in enumerate(outs)}
    arrays.update({"coef%d" % (i + 1): npy(c) for i, c in enumerate(coef)})
    save("dispnets.npz", x=npy(x), digest=dig,
         **grads_of(m, ["conv1.0.weight", "conv1.2.weight", "upconv7.0.weight", "upconv1.0.bias", "iconv3.0.weight",
                        "predict_disp1.0.weight"]), **arrays)
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def case_posefc():
    m = reinit_by_name(ref_pf.PoseFc(), 61)      # non-trivial head (the shipped init zeroes fc_loc[-1].weight)
    g = torch.Generator().manual_seed(62)
    tgt, r0, r1 = (torch.randn(1, 3, 384, 1280, generator=g) for _ in range(3))
    out = m(tgt, [r0, r1])
    coef = torch.randn(1, 2, 6, generator=g)
    (out * coef).sum().backward()
    save("posefc.npz", coef=npy(coef), out=npy(out), digest=weight_digest(m),
         **grads_of(m, ["fc_loc.0.weight", "fc_loc.4.weight", "pose_pred.bias", "conv7.0.bias"]))
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def case_metrics():
    """evaluate.compute_errors (evaluate.py:6-39).  The function is broken as written: disp_to_depth(pred[0]) returns a nested list and
    .cpu() on it raises.  Harness-side shim (reference file untouched): the module-global name `disp_to_depth` it looks up is pointed
    at a wrapper that runs the REFERENCE's disp_to_depth on [[tensor]] and unwraps the result; pred is passed as [tensor]."""
    import evaluate as ref_eval
    ref_eval.disp_to_depth = lambda t: ref_pg.disp_to_depth([[t]])[0][0]
    g = torch.Generator().manual_seed(71)
    gt = 1.0 + 40.0 * torch.rand(2, 1, 24, 40, generator=g)
    disp = (1.0 / gt - 0.01) / 10.0 * (1.0 + 0.25 * torch.randn(gt.shape, generator=g)).clamp(0.3, 3.0)
    disp = disp.clamp(1e-4, 1.0)
    acc = ref_eval.compute_errors(gt, [disp])
    save("metrics.npz", gt=npy(gt), disp=npy(disp), **{k: np.float64(v) for k, v in acc.items()})


def case_pseudo_lidar():
    """PseudoLiDAR.project_PL (pseudo-lidar/utils/PseudoLiDAR.py:69-110) called on the reference class itself; __init__ only reads
    calibration files into T and P, which are set directly here (KITTI-like values)."""
    sys.path.insert(0, os.path.join(REF, "pseudo-lidar", "utils"))
    import PseudoLiDAR as ref_pl
    rng = np.random.RandomState(5)
    R = np.array([[7.533745e-03, -9.999714e-01, -6.166020e-04], [1.480249e-02, 7.280733e-04, -9.998902e-01],
                  [9.998621e-01, 7.523790e-03, 1.480755e-02]])          # KITTI calib_velo_to_cam-like (camera z = velodyne x)
    T = np.vstack([np.concatenate((R, np.array([[-4.069766e-03], [-7.631618e-02], [-2.717806e-01]])), axis=1), [0, 0, 0, 1]])
    out = {}
    for name, (rows, cols, sparsity) in {"a": (24, 78, 0), "b": (37, 61, 3), "c": (16, 40, 7)}.items():
        # P_rect_02-like, scaled to the small image (principal point at its centre: rows above it rise past the 1 m cut with depth)
        P = np.array([[60.0, 0.0, cols / 2.0, 4.485728e+01 / 12], [0.0, 60.0, rows / 2.0, 2.163791e-01 / 12], [0.0, 0.0, 1.0, 2.745884e-03]])
        depth = (2.0 + 60.0 * rng.rand(rows, cols)).astype(np.float32)
        pl = object.__new__(ref_pl.PseudoLiDAR)
        pl.T, pl.P, pl.sparsity = T, P, sparsity
        out["P_" + name] = P
        out["depth_" + name] = depth
        out["cloud_" + name] = pl.project_PL(depth)
        out["sparsity_" + name] = np.int64(sparsity)
    save("pseudo_lidar.npz", T=T, **out)


def case_preprocess():
    """The image transform chain of dataloaders.py:32-49 / trainer.py:97-103.  torchvision is not installed: its four transforms are
    applied through what they call (documented 0.9.1 behaviour): ToTensor(float HWC array) = transpose; ToPILImage(float tensor) =
    mul(255).byte() -> PIL RGB; Resize((h, w)) on a PIL image = Image.resize((w, h), BILINEAR) -- run with the INSTALLED Pillow, the
    library the reference itself calls; ToTensor(PIL) = byte / 255; Normalize = (x - mean) / std."""
    from PIL import Image
    rng = np.random.RandomState(17)
    out = {}
    for name, (h0, w0, h, w) in {"down": (75, 248, 40, 128), "up": (20, 33, 45, 80), "mixed": (37, 61, 37, 30), "kitti": (94, 311, 48, 160)}.items():
        yy, xx = np.mgrid[0:h0, 0:w0]
        base = 127 + 90 * np.sin(yy[..., None] / 7.0 + np.arange(3)) * np.cos(xx[..., None] / 11.0)      # smooth image + noise
        img = np.clip(base + 25 * rng.randn(h0, w0, 3), 0, 255).astype(np.uint8)
        f = np.asarray(img, dtype=np.float32) / 255.0                                   # dataloaders.py:33,38
        t = torch.from_numpy(f.transpose(2, 0, 1))                                      # ToTensor on a float array
        pil = Image.fromarray(np.transpose(t.mul(255).byte().numpy(), (1, 2, 0)))       # ToPILImage
        small = np.asarray(pil.resize((w, h), Image.BILINEAR))                          # Resize
        x = torch.from_numpy(small.transpose(2, 0, 1).copy()).to(torch.float32).div(255)          # ToTensor on a PIL image
        mean = torch.tensor((0.485, 0.456, 0.406)).view(3, 1, 1)
        std = torch.tensor((0.229, 0.224, 0.225)).view(3, 1, 1)
        out["img_" + name] = img
        out["resized_" + name] = small
        out["out_" + name] = npy((x - mean) / std)                                      # Normalize
    import PIL
    save("preprocess.npz", pillow_version=np.array(PIL.__version__), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:          # regenerate selected cases only: python tests/golden/make_golden.py case_metrics case_pseudo_lidar
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    case_metrics()
    case_pseudo_lidar()
    case_loss_small()
    case_loss_ka1()
    case_warp_edge()
    case_ssim()
    case_loss_ssim()
    case_smooth()
    case_preprocess()
    keys = {"PoseNet": case_posenet(), "DepthDecoder": case_decoder(), "DispResNet": case_dispresnet(),
            "DispNetS": case_dispnets(), "PoseFc": case_posefc()}
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    print("wrote state_dict_keys.json")
