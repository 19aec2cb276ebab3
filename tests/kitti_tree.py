"""Generates a tiny KITTI-raw-shaped directory tree (two dates with different image sizes, calibration files, depth ground truth,
a split file) for the dataloader tests.  KITTI itself is not available offline."""
import os

import numpy as np

P_RECT = {"2011_09_26": [721.5377, 0.0, 609.5593, 44.85728, 0.0, 721.5377, 172.854, 0.2163791, 0.0, 0.0, 1.0, 0.002745884],
          "2011_09_28": [707.0493, 0.0, 604.0814, 45.75831, 0.0, 707.0493, 180.5066, -0.3454157, 0.0, 0.0, 1.0, 0.004981016]}
SIZES = {"2011_09_26": (47, 156), "2011_09_28": (46, 153)}          # 375x1242 and 370x1226 divided by 8


def make_tree(root, frames=5, seed=0):
    """-> (split file path, list of (tgt, ref0, ref1, gt) paths per sample)."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    lines = []
    for date in ("2011_09_26", "2011_09_28"):
        ddir = os.path.join(root, "KITTI", date)
        drive = "%s_drive_0001_sync" % date
        img_dir = os.path.join(ddir, drive, "image_02", "data")
        gt_dir = os.path.join(root, "KITTI", "data_depth_annotated", "train", drive, "proj_depth", "groundtruth", "image_02")
        os.makedirs(img_dir)
        os.makedirs(gt_dir)
        with open(os.path.join(ddir, "calib_cam_to_cam.txt"), "w") as f:
            f.write("calib_time: 09-Jan-2012 13:57:47\n")
            f.write("K_02: 9.597910e+02 0.0 6.960217e+02 0.0 9.569251e+02 2.241806e+02 0.0 0.0 1.0\n")
            f.write("P_rect_02: " + " ".join("%.6e" % v for v in P_RECT[date]) + "\n")
        h, w = SIZES[date]
        for i in range(frames):
            Image.fromarray(rng.randint(0, 256, (h, w, 3)).astype(np.uint8)).save(os.path.join(img_dir, "%010d.png" % i))
            depth = (rng.rand(h, w) * 80 * 256).astype(np.uint16)
            depth[rng.rand(h, w) < 0.7] = 0                                  # sparse, as the annotated depth maps are
            Image.fromarray(depth).save(os.path.join(gt_dir, "%010d.png" % i))
        for i in range(1, frames - 1):
            lines.append(" ".join([os.path.join(img_dir, "%010d.png" % i), os.path.join(img_dir, "%010d.png" % (i - 1)),
                                   os.path.join(img_dir, "%010d.png" % (i + 1)), os.path.join(gt_dir, "%010d.png" % i)]))
    split = os.path.join(root, "split.txt")
    with open(split, "w") as f:
        f.write("\n".join(lines) + "\n")
    return split, [ln.split(" ") for ln in lines]


def config_for(split, root, H=24, W=80, batch=2):
    import yaml
    from conftest import PKG
    cfg = yaml.full_load(open(os.path.join(PKG, "configs", "basic_config.yaml")))
    cfg["datasets"].update(dataset=["KITTI"], split=split, path=os.path.join(root, "KITTI") + os.sep)
    cfg["datasets"]["augmentation"].update(image_width=W, image_height=H)
    cfg["action"].update(batch_size=batch, verbose=False, save_checkpoints=False, num_workers=0, num_epochs=1)
    return cfg
