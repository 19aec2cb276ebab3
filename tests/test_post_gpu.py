"""GPU: the steps after the training step: depth metrics and the depth -> pseudo-LiDAR projection, through the C ABI,
against the oracle and the golden vectors captured from the reference (evaluate.py:6-39, pseudo-lidar/utils/PseudoLiDAR.py:69-110)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_depth_metrics_vs_reference_golden():
    from evaluate import compute_errors
    g = np.load(os.path.join(GOLDEN, "metrics.npz"))
    gt, disp = torch.from_numpy(g["gt"]).to(DEV), torch.from_numpy(g["disp"]).to(DEV)
    acc = compute_errors(gt, [disp])
    assert acc["count"] == gt.numel()
    for k in ("abs_rel", "log10", "rms", "log_rms", "d1", "d2", "d3"):
        assert abs(acc[k] - float(g[k])) <= 2e-5 * max(1.0, abs(float(g[k]))), k
    assert abs(acc["silog"] - float(g["silog"])) <= 1e-4 * float(g["silog"])      # mean(e^2) - mean(e)^2: float32 cancellation in the reference
    # the reference stores rms under 'sq_rel' (evaluate.py:36); here it is the squared-relative error (oracle's sq_rel_fixed)
    from oracle import evaluate as oe
    want = oe.compute_errors(g["gt"], g["disp"])
    assert abs(acc["sq_rel"] - float(want["sq_rel_fixed"])) <= 2e-5 * float(want["sq_rel_fixed"])


def test_depth_metrics_mask_and_size():
    """Sparse ground truth: only gt > min_gt counts; 1.2 M elements (a 4 x 375 x 1242 KITTI batch is 1.9 M) against the oracle in float64."""
    from evaluate import compute_errors
    from oracle import evaluate as oe
    g = torch.Generator().manual_seed(5)
    gt = 1.0 + 79.0 * torch.rand(3, 1, 320, 1248, generator=g)
    gt[torch.rand(gt.shape, generator=g) < 0.7] = 0.0                              # 70 % of the pixels have no LiDAR return
    disp = ((1.0 / gt.clamp_min(1.0) - 0.01) / 10.0 * (1.0 + 0.2 * torch.randn(gt.shape, generator=g))).clamp(1e-4, 1.0)
    acc = compute_errors(gt.to(DEV), disp.to(DEV), min_gt=1e-3)
    m = gt > 1e-3
    want = oe.compute_errors(gt[m].numpy().astype(np.float64), disp[m].numpy().astype(np.float64))
    assert acc["count"] == int(m.sum())
    for k in ("abs_rel", "log10", "rms", "log_rms", "d1", "d2", "d3", "silog"):
        assert abs(acc[k] - float(want[k])) <= 1e-4 * max(1.0, abs(float(want[k]))), k


def test_pseudo_lidar_vs_reference_golden():
    from pseudo_lidar import PseudoLiDAR
    g = np.load(os.path.join(GOLDEN, "pseudo_lidar.npz"))
    for name in "abc":
        pl = PseudoLiDAR.from_matrices(g["T"], g["P_" + name], int(g["sparsity_" + name]))
        got = pl.project_PL(torch.from_numpy(g["depth_" + name]).to(DEV)).cpu().numpy()
        want = g["cloud_" + name]
        assert got.shape == want.shape and got.dtype == np.float64          # same points kept, same order, same sparsification
        assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, np.max(np.abs(want)))


def test_pseudo_lidar_full_size_properties():
    """375 x 1242 (KITTI): count and order against the oracle, idempotent re-run, sparsity k keeps exactly every k-th survivor."""
    from oracle import pseudo_lidar as op
    from pseudo_lidar import PseudoLiDAR
    g = np.load(os.path.join(GOLDEN, "pseudo_lidar.npz"))
    rng = np.random.RandomState(9)
    rows, cols = 375, 1242
    depth = (1.5 + 78.0 * rng.rand(rows, cols)).astype(np.float32)
    P = np.array([[7.215377e+02, 0.0, 6.095593e+02, 4.485728e+01], [0.0, 7.215377e+02, 1.728540e+02, 2.163791e-01], [0.0, 0.0, 1.0, 2.745884e-03]])
    d = torch.from_numpy(depth).to(DEV)
    full = PseudoLiDAR.from_matrices(g["T"], P, 0).project_PL(d)
    want = op.project_PL(depth, g["T"], P, 0)
    assert tuple(full.shape) == want.shape
    assert np.max(np.abs(full.cpu().numpy() - want)) <= 1e-11
    again = PseudoLiDAR.from_matrices(g["T"], P, 0).project_PL(d)
    assert torch.equal(full, again)
    for k in (2, 5, 64):
        sp = PseudoLiDAR.from_matrices(g["T"], P, k).project_PL(d)
        assert torch.equal(sp, full[0::k])


def test_image_preprocess_vs_pillow_golden():
    """8f row 1: the transform chain on the GPU == the reference chain run with the real Pillow, bit for bit (integer resample)."""
    from dataloaders import GpuImageTransform
    g = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    for name in ("down", "up", "mixed", "kitti"):
        img, want = g["img_" + name], g["out_" + name]
        t = GpuImageTransform(want.shape[1], want.shape[2])
        got = t(torch.from_numpy(img))                       # host uint8 -> pinned copy -> device
        assert tuple(got.shape) == want.shape
        assert np.array_equal(got.cpu().numpy(), want)
    # a batch, already on the device; and the intrinsics rescale works on a copy
    img = np.stack([g["img_kitti"], g["img_kitti"][::-1].copy()])
    t = GpuImageTransform(48, 160)
    got = t(torch.from_numpy(img).to(DEV))
    assert np.array_equal(got[0].cpu().numpy(), g["out_kitti"])
    from oracle import preprocess as op
    assert np.array_equal(got[1].cpu().numpy(), op.load_transform(img[1], 48, 160))
    K = torch.tensor([[721.5, 0.0, 609.6], [0.0, 721.5, 172.9], [0.0, 0.0, 1.0]], dtype=torch.float64)
    K2 = t.scale_intrinsics(K, 94, 311)
    assert float(K[0, 0]) == 721.5 and abs(float(K2[0, 0]) - 721.5 * 160 / 311) < 1e-12 and abs(float(K2[1, 2]) - 172.9 * 48 / 94) < 1e-12
    # full KITTI size against the oracle (Pillow restated): 375 x 1242 -> 192 x 640, a batch of 3
    rng = np.random.RandomState(8)
    big = rng.randint(0, 256, (3, 375, 1242, 3)).astype(np.uint8)
    got = GpuImageTransform(192, 640)(torch.from_numpy(big))
    for b in range(3):
        assert np.array_equal(got[b].cpu().numpy(), op.load_transform(big[b], 192, 640))


def test_prefetch_loader_on_kitti_files_equals_the_reference_chain(tmp_path):
    """8f row 1, second half: split file -> PIL decode -> pinned copy -> GPU resize + normalise, one batch ahead.  Every frame of every batch
    equals the reference's transform chain (oracle.preprocess, pinned bit for bit to Pillow) although the two drives differ in image size."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from PIL import Image
    from kitti_tree import config_for, make_tree
    from dataloaders import PrefetchLoader, UnSupKittiDataset, raw_collate
    from oracle import preprocess as op
    split, rows = make_tree(str(tmp_path))
    H, W = 24, 80
    ds = UnSupKittiDataset(config_for(split, str(tmp_path), H, W))
    order = [0, 3, 1, 4, 2, 5]                       # every batch mixes the two source sizes
    loader = torch.utils.data.DataLoader(ds, batch_size=2, sampler=order, num_workers=2, drop_last=True, collate_fn=raw_collate)
    seen = 0
    for bi, batch in enumerate(PrefetchLoader(loader, H, W, DEV)):
        assert tuple(batch["tgt"].shape) == (2, 3, H, W) and batch["intrinsics"].dtype == torch.float64 and batch["groundtruth"].is_cuda
        for j in range(2):
            r = rows[order[2 * bi + j]]
            for got, path in ((batch["tgt"][j], r[0]), (batch["ref_imgs"][0][j], r[1]), (batch["ref_imgs"][1][j], r[2])):
                want = op.load_transform(np.asarray(Image.open(path)), H, W)
                assert np.array_equal(got.cpu().numpy(), want)
            seen += 1
    assert seen == 6


def test_trainer_runs_an_epoch_on_kitti_files(tmp_path):
    """Trainer(config) with datasets.dataset: ['KITTI'] (the reference's configs) starts and trains from a split file."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from kitti_tree import config_for, make_tree
    from trainer import Trainer
    split, _ = make_tree(str(tmp_path), frames=8)
    cfg = config_for(split, str(tmp_path), 64, 128, batch=2)
    cfg["action"]["num_workers"] = 2
    t = Trainer(cfg)
    t.train()
    assert t.step >= 3 and torch.isfinite(sum(t.loss)).item()
    acc = t.validate()
    assert acc is not None and acc["count"] > 0
