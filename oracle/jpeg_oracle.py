"""TEST INFRASTRUCTURE (oracle): numpy restatement of the data-parallel half of libjpeg-turbo's default decompression path --
what `PIL.Image.open(path).convert("RGB")` runs for the reference's loader workers (engine.py:41-54; Pillow = libjpeg-turbo with
dct_method JDCT_ISLOW and do_fancy_upsampling TRUE).  Third-party algorithm, absent from /root/reference: libjpeg-turbo (bundled in
the Pillow 12.2.0 wheel, reports libjpeg API 6.2); restated from its published sources
  * jidctint.c `jpeg_idct_islow` (13-bit fixed point, PASS1_BITS 2, range limit through the post-IDCT table of jdmaster.c),
  * jdsample.c `h2v1_fancy_upsample` / `h2v2_fancy_upsample` (+ jdmainct.c's context rows: the row above the first / below the last
    real sample row is that row itself),
  * jdcolor.c `ycc_rgb_convert` (SCALEBITS 16 tables),
and PINNED against Pillow itself: tests/test_jpeg.py decodes the same files with PIL and requires equal bytes.

Input = the int16 coefficient blocks the product's HOST entropy decoder produces (`ch_jpeg_entropy_decode`, plain C++ that runs without
a GPU), so the CPU test of this file pins both the host half of the product and this restatement; the GPU test then compares
`ch_jpeg_reconstruct` with PIL directly.  Only tests/ may import this module."""
from __future__ import annotations

import numpy as np

CONST_BITS, PASS1_BITS = 13, 2
F = dict(f0_298=2446, f0_390=3196, f0_541=4433, f0_765=6270, f0_899=7373, f1_175=9633, f1_501=12299, f1_847=15137, f1_961=16069,
         f2_053=16819, f2_562=20995, f3_072=25172)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(v, shift):
    """v: [..., 8] int64 -> [..., 8]; one pass of jidctint.c along the last axis."""
    z2, z3 = v[..., 2], v[..., 6]
    z1 = (z2 + z3) * F["f0_541"]
    tmp2 = z1 + z3 * (-F["f1_847"])
    tmp3 = z1 + z2 * F["f0_765"]
    z2, z3 = v[..., 0], v[..., 4]
    tmp0 = (z2 + z3) << CONST_BITS
    tmp1 = (z2 - z3) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F["f1_175"]
    tmp0, tmp1, tmp2, tmp3 = tmp0 * F["f0_298"], tmp1 * F["f2_053"], tmp2 * F["f3_072"], tmp3 * F["f1_501"]
    z1, z2, z3, z4 = z1 * -F["f0_899"], z2 * -F["f2_562"], z3 * -F["f1_961"] + z5, z4 * -F["f0_390"] + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    out = np.stack([tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3], axis=-1)
    return _descale(out, shift)


def idct_blocks(coef: np.ndarray, quant: np.ndarray) -> np.ndarray:
    """coef [nblocks, 64] int16 (natural order), quant [64] -> samples [nblocks, 8, 8] uint8."""
    v = coef.astype(np.int64).reshape(-1, 8, 8) * quant.astype(np.int64).reshape(1, 8, 8)
    ws = _idct_1d(np.swapaxes(v, 1, 2), CONST_BITS - PASS1_BITS)          # pass 1: columns ([block, col, row])
    ws = np.swapaxes(ws, 1, 2)                                            # -> [block, row, col]
    out = _idct_1d(ws, CONST_BITS + PASS1_BITS + 3)                       # pass 2: rows
    idx = out & 1023                                                      # post-IDCT range-limit table, level shift included
    return np.where(idx < 128, idx + 128, np.where(idx < 512, 255, np.where(idx < 896, 0, idx - 896))).astype(np.uint8)


def _plane(blocks: np.ndarray, bh: int, bw: int) -> np.ndarray:
    return blocks.reshape(bh, bw, 8, 8).transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)


def _h2v1(p: np.ndarray) -> np.ndarray:
    """p [rows, dw] -> [rows, 2 dw] (jdsample.c h2v1_fancy_upsample)."""
    p = p.astype(np.int64)
    left = np.concatenate([p[:, :1], p[:, :-1]], axis=1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], axis=1)
    even = (3 * p + left + 1) >> 2
    odd = (3 * p + right + 2) >> 2
    even[:, 0] = p[:, 0]
    odd[:, -1] = p[:, -1]
    return np.stack([even, odd], axis=2).reshape(p.shape[0], -1)


def _h2v2(p: np.ndarray) -> np.ndarray:
    """p [dh, dw] -> [2 dh, 2 dw] (jdsample.c h2v2_fancy_upsample with jdmainct.c's replicated edge context rows)."""
    p = p.astype(np.int64)
    above = np.concatenate([p[:1], p[:-1]], axis=0)
    below = np.concatenate([p[1:], p[-1:]], axis=0)
    rows = np.stack([3 * p + above, 3 * p + below], axis=1).reshape(2 * p.shape[0], p.shape[1])   # column sums per output row
    last = np.concatenate([rows[:, :1], rows[:, :-1]], axis=1)
    nxt = np.concatenate([rows[:, 1:], rows[:, -1:]], axis=1)
    even = (rows * 3 + last + 8) >> 4
    odd = (rows * 3 + nxt + 7) >> 4
    even[:, 0] = (rows[:, 0] * 4 + 8) >> 4
    odd[:, -1] = (rows[:, -1] * 4 + 7) >> 4
    return np.stack([even, odd], axis=2).reshape(rows.shape[0], -1)


def reconstruct(coef: np.ndarray, desc) -> np.ndarray:
    """coef: this image's int16 coefficients (component 0 blocks row-major, then components 1, 2); desc: its ch_jpeg_desc (any object
    with width, height, ncomp, hs, vs, mcu_w, mcu_h, quant) -> RGB uint8 [height, width, 3]."""
    get = (lambda k: desc[k]) if isinstance(desc, (np.void, dict)) else (lambda k: getattr(desc, k))   # numpy record or ctypes struct
    w, h, hs, vs, mw, mh = (int(get(k)) for k in ("width", "height", "hs", "vs", "mcu_w", "mcu_h"))
    q = np.asarray(get("quant"), dtype=np.int64).reshape(3, 64)
    nby = mw * hs * mh * vs
    coef = coef.reshape(-1, 64)
    Y = _plane(idct_blocks(coef[:nby], q[0]), mh * vs, mw * hs)[:h, :w].astype(np.int64)
    if int(get("ncomp")) == 1:
        return np.repeat(Y[:, :, None], 3, axis=2).astype(np.uint8)
    nbc = mw * mh
    dw, dh = -(-w // hs), -(-h // vs)
    planes = []
    for c in (1, 2):
        p = _plane(idct_blocks(coef[nby + (c - 1) * nbc: nby + c * nbc], q[c]), mh, mw)[:dh, :dw]
        if hs == 2 and vs == 2:
            p = _h2v2(p)
        elif hs == 2 and vs == 1:
            p = _h2v1(p)
        planes.append(p[:h, :w].astype(np.int64) - 128)
    cb, cr = planes
    r = Y + ((91881 * cr + 32768) >> 16)
    b = Y + ((116130 * cb + 32768) >> 16)
    g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    return np.clip(np.stack([r, g, b], axis=2), 0, 255).astype(np.uint8)
