"""CPU restatement of the reference's image transform chain -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The chain (reference dataloaders.py:32-49 load_img + trainer.py:97-103):
    np.asarray(Image.open(path), float32) / 255 -> ToTensor -> ToPILImage -> Resize((h, w)) -> ToTensor -> Normalize(ImageNet)
runs in two third-party libraries that are not part of /root/reference:
  * torchvision.transforms (unpinned in utils/requirements.txt:2; nominal 0.9.1, docker/Dockerfile:8), NOT installed here.  Its
    published behaviour for this chain is restated: ToTensor on a float HWC array = transpose only; ToPILImage on a float tensor
    = `pic.mul(255).byte()` (TRUNCATION: float32(v)/255*255 lands just below v for some v, so those bytes drop by one -- a quirk of
    the reference's chain that is kept); Resize on a PIL image = `img.resize((w, h), Image.BILINEAR)`; ToTensor on a PIL image =
    byte / 255 in float32; Normalize = (x - mean) / std.
  * Pillow's resize (ImagingResample, src/libImaging/Resample.c), installed here as 12.2.0: antialiased separable triangle
    filter, support = max(scale, 1), coefficients normalised in double and converted to 22-bit fixed point, horizontal pass then
    vertical pass through an 8-bit intermediate, `(1 << 21) + sum >> 22` clipped to 0..255.  Restated below and PINNED bit for bit
    against Pillow itself (tests/test_oracle_golden.py when PIL is importable; tests/golden/preprocess.npz otherwise).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def byte_quirk(img_u8):
    """uint8 -> float32 / 255 (dataloaders.py:33,38) -> ToPILImage's mul(255).byte() truncation.  uint8 in, uint8 out."""
    f = img_u8.astype(np.float32) / np.float32(255.0)
    return (f * np.float32(255.0)).astype(np.uint8)          # astype truncates toward zero, like torch's .byte()


def resample_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter over the whole axis.
    -> ksize, bounds [out][2] (first source index, count), kk [out][ksize] int32 fixed-point weights."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size          # box ends are C floats in Pillow
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = []
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - a if a < 1.0 else 0.0)
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img, bounds, kk, axis):
    """One separable pass along `axis` (1 = horizontal, 0 = vertical) of a uint8 [H, W, C] image."""
    out_size = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = out_size
    out = np.zeros(shape, np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, xmax = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.take(0, axis=axis).shape, 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            acc = acc + src.take(xmin + x, axis=axis) * int(kk[xx, x])
        v = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        if axis == 1:
            out[:, xx, :] = v
        else:
            out[xx, :, :] = v
    return out


def pil_resize_bilinear_u8(img_u8, out_h, out_w):
    """== np.asarray(Image.fromarray(img_u8).resize((out_w, out_h), Image.BILINEAR)) for [H, W, 3] uint8."""
    h, w = img_u8.shape[:2]
    cur = img_u8
    if out_w != w:                       # horizontal first (Pillow restricts it to the rows the vertical pass reads: same values)
        _, b, k = resample_coeffs(w, out_w)
        cur = _pass(cur, b, k, 1)
    if out_h != h:
        _, b, k = resample_coeffs(h, out_h)
        cur = _pass(cur, b, k, 0)
    return cur


def load_transform(img_u8, out_h, out_w):
    """The whole chain on a decoded uint8 [H, W, 3] image -> float32 [3, out_h, out_w] (what load_img returns, dataloaders.py:32-49)."""
    small = pil_resize_bilinear_u8(byte_quirk(img_u8), out_h, out_w)
    x = small.astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(MEAN, np.float32)) / np.asarray(STD, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def scale_intrinsics(K, out_h, out_w, og_h, og_w):
    """dataloaders.py:95-98 on a COPY (the reference rescales the cached sample's K in place on every fetch: cumulative)."""
    K = np.array(K, dtype=np.float64, copy=True)
    K[0] *= out_w / og_w
    K[1] *= out_h / og_h
    return K
