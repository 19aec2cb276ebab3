"""dataloaders.py -- sample source for the trainer.

The reference's KITTI loader (dataloaders.py:14-252) is PIL decode + a torchvision transform chain + calib/oxts parsing.  Built here:
the split-file driven reader (UnSupKittiFiles: paths, P_rect_02 intrinsics, ground-truth maps; tested on a generated KITTI-shaped
tree, KITTI itself is not available offline), the transform chain on the GPU (GpuImageTransform: SURVEY.md section 8f "next" row 1) and
the one-batch-ahead PrefetchLoader that joins them.  Not built: the OXTS pose packets of the semi-supervised experiment.  What the
training step consumes is kept: a dict with 'tgt' [3,H,W], 'ref_imgs' [2 x [3,H,W]], 'intrinsics' [3,3] fp64,
'groundtruth' [1,H,W] (reference dataloaders.py:226-251).  SyntheticTriplets produces such samples from a seed.
"""
import ctypes

import numpy as np
import torch
from torch.utils.data import Dataset


class GpuImageTransform:
    """The reference's transform chain (trainer.py:97-103 applied by dataloaders.py:32-49 load_img) on the GPU, for batches of
    decoded uint8 RGB images: /255, ToPILImage, Resize((h, w)) -- Pillow's antialiased bilinear resample, reproduced bit for bit --,
    ToTensor, Normalize(ImageNet).  Decoding the file stays on the host (PIL); one pinned copy brings the bytes over.

        t = GpuImageTransform(192, 640)
        x = t(torch.from_numpy(np.asarray(Image.open(path))))         # [H0, W0, 3] uint8 -> [3, 192, 640] float32 on the GPU
        K = t.scale_intrinsics(K, og_h, og_w)                         # dataloaders.py:95-98, on a copy
    """
    MEAN = (0.485, 0.456, 0.406)
    STD = (0.229, 0.224, 0.225)

    def __init__(self, img_height, img_width, device="cuda"):
        self.h, self.w, self.device = int(img_height), int(img_width), torch.device(device)
        self._tables = {}

    def _axis(self, in_size, out_size):
        from mcav import lib as L
        key = (in_size, out_size)
        if key not in self._tables:
            h = L.lib()
            ks = ctypes.c_int(0)
            cap = h.mcav_resample_coeffs(in_size, out_size, ctypes.byref(ks), None, None, 0)
            if cap <= 0:
                raise L.MCAVError("mcav_resample_coeffs: cannot resample %d -> %d" % (in_size, out_size))
            bounds = np.zeros(out_size * 2, np.int32)
            kk = np.zeros(cap, np.int32)
            L.check(h.mcav_resample_coeffs(in_size, out_size, ctypes.byref(ks), bounds.ctypes.data_as(ctypes.c_void_p),
                                           kk.ctypes.data_as(ctypes.c_void_p), cap), "mcav_resample_coeffs")
            self._tables[key] = (torch.from_numpy(bounds).to(self.device), torch.from_numpy(kk).to(self.device), ks.value)
        return self._tables[key]

    def __call__(self, img_u8):
        """img_u8: uint8 [H0, W0, 3] or [B, H0, W0, 3] (host or device).  -> float32 [3, h, w] / [B, 3, h, w] on the device."""
        from mcav import lib as L
        single = img_u8.dim() == 3
        x = img_u8.unsqueeze(0) if single else img_u8
        if x.dtype != torch.uint8 or x.shape[-1] != 3:
            raise L.MCAVError("GpuImageTransform: expected uint8 [..., H, W, 3], got %s %s" % (x.dtype, tuple(x.shape)))
        if not x.is_cuda:
            x = x.contiguous().pin_memory().to(self.device, non_blocking=True)
        x = x.contiguous()
        B, H0, W0, _ = x.shape
        hb, hk, hks = self._axis(W0, self.w)
        vb, vk, vks = self._axis(H0, self.h)
        h = L.lib()
        ws = L.workspace(h.mcav_image_preprocess_workspace_bytes(B, H0, self.w), x.device, "preprocess")
        out = torch.empty((B, 3, self.h, self.w), dtype=torch.float32, device=x.device)
        mean = (ctypes.c_float * 3)(*self.MEAN)
        std = (ctypes.c_float * 3)(*self.STD)
        L.check(h.mcav_image_preprocess(L.ptr(x), B, H0, W0, self.h, self.w, L.ptr(hb), L.ptr(hk), hks, L.ptr(vb), L.ptr(vk), vks, mean, std,
                                        L.ptr(out), L.ptr(ws), ws.numel(), L.stream()), "mcav_image_preprocess")
        return out[0] if single else out

    def scale_intrinsics(self, K, og_h, og_w):
        """dataloaders.py:95-98 -- on a COPY: the reference scales the cached sample's matrix in place on every fetch."""
        K = torch.as_tensor(K, dtype=torch.float64).clone()
        K[0] *= self.w / og_w
        K[1] *= self.h / og_h
        return K


def _register():
    from mcav import lib as L
    L.register({
        "mcav_resample_coeffs": (L.c_i, [L.c_i, L.c_i, L.c_p, L.c_p, L.c_p, L.c_i]),
        "mcav_image_preprocess_workspace_bytes": (L.c_sz, [L.c_i, L.c_i, L.c_i]),
        "mcav_image_preprocess": (L.c_i, [L.c_p, L.c_i, L.c_i, L.c_i, L.c_i, L.c_i, L.c_p, L.c_p, L.c_i, L.c_p, L.c_p, L.c_i, L.c_p, L.c_p, L.c_p,
                                          L.c_p, L.c_sz, L.c_p]),
    })


_register()


class SyntheticTriplets(Dataset):
    def __init__(self, config, transforms=None, length=None):
        aug = config['datasets']['augmentation']
        self.H, self.W = aug['image_height'], aug['image_width']
        self.length = length or int(config['datasets'].get('synthetic_length', 64))
        self.seed = int(config['action'].get('random_seed', 0))

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        H, W = self.H, self.W
        imgs = []
        for _ in range(3):
            x = torch.randn(1, 3, H, W, generator=g)
            x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect"), 3, 1)[0]
            imgs.append(x.contiguous())
        K = torch.tensor([[0.58 * W, 0.0, 0.5 * W], [0.0, 1.92 * H, 0.5 * H], [0.0, 0.0, 1.0]], dtype=torch.float64)
        return {"tgt": imgs[0], "ref_imgs": [imgs[1], imgs[2]], "intrinsics": K, "groundtruth": torch.zeros(1, H, W)}


def synthetic_batch(B, H, W, seed=1234):
    """A whole seeded batch in the collated layout the trainer consumes (bench.py's input): low-passed randn images,
    KITTI-like fp64 intrinsics as the reference's loader gives them."""
    g = torch.Generator().manual_seed(seed)
    imgs = []
    for _ in range(3):
        x = torch.randn(B, 3, H, W, generator=g)
        imgs.append(torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect"), 3, 1).contiguous())
    K = torch.tensor([[0.58 * W, 0.0, 0.5 * W], [0.0, 1.92 * H, 0.5 * H], [0.0, 0.0, 1.0]], dtype=torch.float64).repeat(B, 1, 1)
    return {"tgt": imgs[0], "ref_imgs": [imgs[1], imgs[2]], "intrinsics": K, "groundtruth": torch.zeros(B, 1, H, W)}


# ------------------------------------------------------------------------------------------------ KITTI files (reference dataloaders.py:18-171)
def read_calib_file(path):
    """'key: v0 v1 ...' lines -> {key: float64 array}; non-numeric values (dates) are skipped (reference geometry/calibration.py:70-89)."""
    data = {}
    with open(path, "r") as f:
        for line in f:
            line = line.rstrip()
            if not line or ":" not in line:
                continue
            key, value = line.split(":", 1)
            try:
                data[key] = np.array([float(x) for x in value.split()])
            except ValueError:
                pass
    return data


def find_calib_dir(image_path):
    """The KITTI date directory (.../2011_09_26/) above an image path.  The reference slices a fixed number of characters off the path
    (dataloaders.py:155: [:29], 'mac - 20, beauty - 29'), which only works for its two directory layouts."""
    import os
    import re
    d = os.path.dirname(os.path.abspath(image_path)) if os.path.isabs(image_path) else os.path.dirname(image_path)
    while d and d not in ("/", "."):
        if re.fullmatch(r"\d{4}_\d{2}_\d{2}", os.path.basename(d)):
            return d + os.sep
        d = os.path.dirname(d)
    raise FileNotFoundError("no KITTI date directory (YYYY_MM_DD) above %s" % image_path)


class KittiDataset(Dataset):
    """Reference dataloaders.py:18-128.  Samples hold file paths only; an image is decoded on fetch (PIL, on the host).

    Two ways to get the reference's tensors out of it:
      * `transforms` = a list of host callables applied exactly as load_img does (all but the last to every image, the last -- Normalize --
        not to the ground truth): the reference's own contract, for callers that bring torchvision;
      * `transforms=None` (what Trainer passes): fetches return the decoded uint8 image ('raw' samples) and the transform chain
        (/255, Resize, Normalize) runs on the GPU in PrefetchLoader / GpuImageTransform, bit-exact against Pillow.
    OXTS poses ('oxts', used only by the reference's semi-supervised pose experiment, trainer.py:301-304) are not loaded."""

    def __init__(self, config, transforms=None):
        super().__init__()
        ds = config['datasets']
        self.split = ds['split']
        self.kitti_filepath = ds['path']
        self.img_width = ds['augmentation']['image_width']
        self.img_height = ds['augmentation']['image_height']
        self.seq_len = ds.get('sequence_length', 3)
        self.transforms = transforms
        self.raw = transforms is None
        self.samples = []
        self._calib = {}

    def __len__(self):
        return len(self.samples)

    def resolve(self, path):
        """Split-file paths are relative to the directory the reference is run from; also accept them relative to datasets.path."""
        import os
        if os.path.exists(path):
            return path
        tail = path.lstrip("./")
        for root in (self.kitti_filepath, os.path.dirname(os.path.normpath(self.kitti_filepath))):
            for cand in (os.path.join(root, tail), os.path.join(root, *tail.split("/")[1:]) if "/" in tail else None):
                if cand and os.path.exists(cand):
                    return cand
        return path

    def intrinsics_of(self, image_path):
        """calib.P[:, :3] of the drive's date directory (reference dataloaders.py:155-157), cached per directory; a fresh copy per call."""
        d = find_calib_dir(self.resolve(image_path))
        if d not in self._calib:
            self._calib[d] = read_calib_file(d + "calib_cam_to_cam.txt")["P_rect_02"].reshape(3, 4)[:, :3].copy()
        return self._calib[d].copy()

    def load_img(self, path, gt=False):
        """-> (image, original height, original width).  raw mode: uint8 [H0, W0, 3] tensor (the GPU runs the chain); ground truth: the
        depth PNG as float32, resized with Pillow's bilinear filter on mode 'F' (what ToPILImage + Resize do to a float map), [1, h, w]."""
        from PIL import Image
        img = Image.open(self.resolve(path))
        if gt:
            arr = np.asarray(img, dtype=np.float32)
            if self.transforms is not None:
                for t in self.transforms[:-1]:
                    arr = t(arr)
                return arr.squeeze(), None, None
            small = Image.fromarray(arr, mode="F").resize((self.img_width, self.img_height), Image.BILINEAR)
            return torch.from_numpy(np.asarray(small, dtype=np.float32).copy())[None], None, None
        arr = np.asarray(img.convert("RGB") if img.mode != "RGB" else img)
        h, w = arr.shape[0], arr.shape[1]
        if self.transforms is not None:
            x = arr.astype(np.float32) / 255.0
            for t in self.transforms[:-1]:
                x = t(x)
            return self.transforms[-1](x).squeeze(), h, w
        return torch.from_numpy(arr.copy()), h, w

    def __getitem__(self, index):
        sample = self.samples[index]
        ret = {}
        ret['tgt'], og_h, og_w = self.load_img(sample['tgt'])
        ret['ref_imgs'] = [self.load_img(p)[0] for p in sample['ref_imgs']]
        K = sample['intrinsics'].copy()                  # the reference scales the CACHED matrix in place on every fetch (dataloaders.py:95-98)
        K[0] *= self.img_width / og_w
        K[1] *= self.img_height / og_h
        ret['intrinsics'] = torch.from_numpy(K)
        if sample.get('groundtruth'):
            ret['groundtruth'] = self.load_img(sample['groundtruth'], gt=True)[0]
        else:
            ret['groundtruth'] = torch.zeros(1, self.img_height, self.img_width)
        return ret


class UnSupKittiFiles(KittiDataset):
    """Reference UnSupKittiDataset (dataloaders.py:131-171): one sample per line of the split file,
    '<tgt.png> <ref0.png> <ref1.png> <groundtruth.png>'."""

    def __init__(self, config, transforms=None):
        super().__init__(config, transforms)
        with open(self.split, "r") as f:
            lines = [ln.strip() for ln in f if ln.strip()]
        for ln in lines:
            parts = ln.split(" ")
            if len(parts) < 3:
                raise ValueError("split file %s: expected 'tgt ref0 ref1 [groundtruth]', got %r" % (self.split, ln))
            self.samples.append({'tgt': parts[0], 'ref_imgs': parts[1:3], 'intrinsics': self.intrinsics_of(parts[0]),
                                 'groundtruth': parts[3] if len(parts) > 3 else None})


def raw_collate(samples):
    """collate_fn for raw samples: KITTI drives differ in image size (375x1242, 370x1226, ...), so the uint8 images cannot be stacked on
    the host; the list goes to PrefetchLoader, which resizes on the GPU."""
    return samples


class PrefetchLoader:
    """One batch ahead: a background thread takes the DataLoader's raw sample lists, copies the decoded uint8 frames to the GPU through pinned
    memory on its own stream, runs the fused /255 + Pillow-exact resize + Normalize kernel there (GpuImageTransform, grouped by source size)
    and hands the trainer finished batches in the reference's collated layout (tgt [B,3,h,w], ref_imgs 2 x [B,3,h,w], intrinsics [B,3,3] fp64,
    groundtruth [B,1,h,w]) together with the event the consumer's stream has to wait on.  SURVEY.md 8f row 1, second half."""

    def __init__(self, loader, img_height, img_width, device="cuda", depth=2):
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())          # the worker thread needs an explicit index
        self.loader, self.h, self.w, self.device, self.depth = loader, int(img_height), int(img_width), dev, depth
        self.transform = GpuImageTransform(img_height, img_width, dev)

    def __len__(self):
        return len(self.loader)

    def _finish(self, samples, stream):
        B = len(samples)
        frames = [s['tgt'] for s in samples] + [s['ref_imgs'][0] for s in samples] + [s['ref_imgs'][1] for s in samples]
        out = torch.empty((3 * B, 3, self.h, self.w), dtype=torch.float32, device=self.device)
        with torch.cuda.stream(stream):
            groups = {}
            for i, f in enumerate(frames):
                groups.setdefault(tuple(f.shape), []).append(i)
            for shape, idx in groups.items():
                out[idx] = self.transform(torch.stack([frames[i] for i in idx]))
            K = torch.stack([s['intrinsics'] for s in samples]).to(self.device, non_blocking=True)
            gt = torch.stack([s['groundtruth'] for s in samples]).to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(stream)
        return {'tgt': out[:B], 'ref_imgs': [out[B:2 * B], out[2 * B:]], 'intrinsics': K, 'groundtruth': gt}, done

    def __iter__(self):
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth)
        stream = torch.cuda.Stream(device=self.device)
        dev = self.device

        def work():
            try:
                torch.cuda.set_device(dev)
                for samples in self.loader:
                    q.put(self._finish(samples, stream))
                q.put(None)
            except BaseException as e:          # surface loader errors in the consumer
                q.put(e)
        t = threading.Thread(target=work, daemon=True)
        t.start()
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            batch, done = item
            torch.cuda.current_stream(self.device).wait_event(done)
            for v in [batch['tgt']] + batch['ref_imgs'] + [batch['intrinsics'], batch['groundtruth']]:
                v.record_stream(torch.cuda.current_stream(self.device))
            yield batch
        t.join()


def UnSupKittiDataset(config, transforms=None):
    """The dataset the trainer asks for (reference trainer.py:106): datasets.dataset == ['synthetic'] -> seeded synthetic triplets (no KITTI
    offline); anything else -> the split-file driven KITTI reader above.  A missing split file or data root fails HERE, at configuration
    time, with the path in the message."""
    import os
    if config['datasets'].get('dataset', ['KITTI']) == ['synthetic']:
        return SyntheticTriplets(config, transforms)
    split = config['datasets']['split']
    if not os.path.exists(split):
        raise FileNotFoundError("datasets.split = %r does not exist (KITTI is not shipped with this repository; set datasets.dataset: "
                                "['synthetic'] for seeded synthetic triplets, or point datasets.split / datasets.path at a KITTI raw tree)" % split)
    return UnSupKittiFiles(config, transforms)
