"""CPU oracle for the image pre-processing path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference pre-processes on CPU workers with torchvision transforms on PIL images (configs/dataset/cub200.yaml:31-47:
Resize(256, bicubic) -> CenterCrop(224) -> ToTensor -> normalize(norm)); the arithmetic is the third-party Pillow
resampler (Pillow >= 7 `ImagingResample`, two passes, 8-bit fixed point).  This file restates that algorithm in numpy --
  * coefficients: support = 2 * max(scale, 1) taps of the Keys bicubic (a = -0.5) around (x + 0.5) * scale, normalised,
    quantised to 22 fractional bits;
  * horizontal pass over the source rows the vertical pass needs, result rounded to uint8; then the vertical pass; uint8;
  * crop, /255, (x - mean) / std in fp32
-- and is PINNED against Pillow itself (tests/test_preprocess.py: bit-equal uint8 images on odd aspect ratios, up- and
down-scaling), which is what the GPU kernel (csrc/preprocess.hip) is then checked against.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic_filter(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def coeffs(in_size: int, out_size: int):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the box (0, in_size): per output index (xmin, int32 weights)."""
    scale = float(np.float32(in_size) - np.float32(0)) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ss = 1.0 / filterscale
    out = []
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(k)
        if ww != 0.0:
            k = [w / ww for w in k]
        kk = [int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS)) for w in k]
        out.append((xmin, np.asarray(kk, dtype=np.int64)))
    return out


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bicubic(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """img uint8 [H, W, 3] -> uint8 [out_h, out_w, 3], Pillow Image.resize((out_w, out_h), BICUBIC) semantics."""
    H, W, _ = img.shape
    cur = img
    if out_w != W:
        ch = coeffs(W, out_w)
        tmp = np.empty((H, out_w, 3), dtype=np.uint8)
        src = img.astype(np.int64)
        for xx, (xmin, kk) in enumerate(ch):
            acc = (src[:, xmin:xmin + len(kk), :] * kk[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            tmp[:, xx, :] = _clip8(acc)
        cur = tmp
    if out_h != H:
        cv = coeffs(H, out_h)
        out = np.empty((out_h, cur.shape[1], 3), dtype=np.uint8)
        src = cur.astype(np.int64)
        for yy, (ymin, kk) in enumerate(cv):
            acc = (src[ymin:ymin + len(kk), :, :] * kk[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        cur = out
    return cur


def resized_size(w: int, h: int, size: int):
    """torchvision Resize(int): shorter side -> size, the other side int(size * long / short) (truncation)."""
    if w <= h:
        return size, max(1, int(size * h / w))
    return max(1, int(size * w / h)), size


def crop_origin(w: int, h: int, crop: int):
    """torchvision CenterCrop: int(round((dim - crop) / 2.0)) with Python's round-half-to-even."""
    return int(round((w - crop) / 2.0)), int(round((h - crop) / 2.0))


def preprocess(img: np.ndarray, resize: int, crop: int, mean, std) -> np.ndarray:
    """uint8 [H, W, 3] -> fp32 [3, crop, crop]: Resize(resize, bicubic) -> CenterCrop(crop) -> /255 -> (x - mean) / std."""
    H, W, _ = img.shape
    nw, nh = resized_size(W, H, resize)
    r = resize_bicubic(img, nw, nh)
    left, top = crop_origin(nw, nh, crop)
    if left < 0 or top < 0:
        raise ValueError("crop larger than the resized image is not supported")
    c = r[top:top + crop, left:left + crop, :]
    x = c.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    m = np.asarray(mean, dtype=np.float32)[:, None, None]
    s = np.asarray(std, dtype=np.float32)[:, None, None]
    return (x - m) / s


def preprocess_train(img: np.ndarray, box, flip: bool, crop: int, mean, std) -> np.ndarray:
    """The TRAINING chain (configs/dataset/cub200.yaml:13-23) for given random draws: crop `box` = (top, left, height, width)
    (torchvision RandomResizedCrop.get_params), resize the box to crop x crop as an image of its own (PIL `img.crop(box).resize`),
    mirror the columns when `flip` (PIL FLIP_LEFT_RIGHT), /255, (x - mean) / std.  uint8 [H, W, 3] -> fp32 [3, crop, crop]."""
    top, left, bh, bw = (int(v) for v in box)
    r = resize_bicubic(np.ascontiguousarray(img[top:top + bh, left:left + bw, :]), crop, crop)
    if flip:
        r = r[:, ::-1, :]
    x = r.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    m = np.asarray(mean, dtype=np.float32)[:, None, None]
    s = np.asarray(std, dtype=np.float32)[:, None, None]
    return (x - m) / s
