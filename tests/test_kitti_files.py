"""CPU: the split-file driven KITTI reader (reference dataloaders.py:18-171) on a generated KITTI-shaped tree."""
import numpy as np
import pytest
import torch

from kitti_tree import P_RECT, SIZES, config_for, make_tree


def test_split_file_samples_intrinsics_and_ground_truth(tmp_path):
    from PIL import Image
    from dataloaders import UnSupKittiDataset
    split, rows = make_tree(str(tmp_path))
    H, W = 24, 80
    ds = UnSupKittiDataset(config_for(split, str(tmp_path), H, W), transforms=None)
    assert len(ds) == len(rows) == 6 and ds.raw
    for idx in (0, 5):
        date = "2011_09_26" if idx < 3 else "2011_09_28"
        h0, w0 = SIZES[date]
        P = np.array(P_RECT[date]).reshape(3, 4)
        for _ in range(2):                     # twice: the reference rescales its cached matrix in place on every fetch (dataloaders.py:95-98)
            s = ds[idx]
            assert s["tgt"].dtype == torch.uint8 and tuple(s["tgt"].shape) == (h0, w0, 3)
            assert np.array_equal(s["tgt"].numpy(), np.asarray(Image.open(rows[idx][0])))
            assert np.array_equal(s["ref_imgs"][1].numpy(), np.asarray(Image.open(rows[idx][2])))
            K = P[:, :3].copy()
            K[0] *= W / w0
            K[1] *= H / h0
            assert s["intrinsics"].dtype == torch.float64 and np.allclose(s["intrinsics"].numpy(), K, rtol=1e-12)
            gt = np.asarray(Image.open(rows[idx][3]), dtype=np.float32)
            want = np.asarray(Image.fromarray(gt, mode="F").resize((W, H), Image.BILINEAR), dtype=np.float32)
            assert tuple(s["groundtruth"].shape) == (1, H, W) and np.array_equal(s["groundtruth"][0].numpy(), want)


def test_host_transform_list_follows_the_reference_contract(tmp_path):
    """transforms given: all but the last applied to every image (and to the ground truth), the last (Normalize) to images only."""
    from dataloaders import UnSupKittiDataset
    split, rows = make_tree(str(tmp_path))
    calls = []
    t0 = lambda a: (calls.append("t0"), torch.from_numpy(np.ascontiguousarray(a)).float())[1]
    t1 = lambda a: (calls.append("norm"), a * 2.0)[1]
    ds = UnSupKittiDataset(config_for(split, str(tmp_path)), transforms=[t0, t1])
    s = ds[0]
    assert calls == ["t0", "norm"] * 3 + ["t0"]          # tgt, ref0, ref1 normalised; ground truth not
    assert float(s["tgt"].max()) <= 2.0 and s["tgt"].dtype == torch.float32


def test_missing_split_fails_at_configuration_time(tmp_path):
    from dataloaders import UnSupKittiDataset
    cfg = config_for(str(tmp_path / "nope.txt"), str(tmp_path))
    with pytest.raises(FileNotFoundError, match="datasets.split"):
        UnSupKittiDataset(cfg)
