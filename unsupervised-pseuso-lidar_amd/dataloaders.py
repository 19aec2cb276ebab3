"""dataloaders.py -- sample source for the trainer.

The reference's KITTI loader (dataloaders.py:14-252) is PIL decode + a torchvision transform chain + calib/oxts parsing.  KITTI
is not available offline, so the file-walking part is not built; the transform chain -- the part with arithmetic in it -- is
(GpuImageTransform: SURVEY.md section 8f "next" row 1).  What the
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


def UnSupKittiDataset(config, transforms=None):
    if config['datasets'].get('dataset', ['KITTI']) == ['synthetic']:
        return SyntheticTriplets(config, transforms)
    raise NotImplementedError("the KITTI file loader is outside this round's hot path (SURVEY.md 8f); set "
                              "datasets.dataset: ['synthetic'] or pass dataset= to Trainer")
