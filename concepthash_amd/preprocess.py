"""GPU image pre-processing through the C-ABI (csrc/preprocess.hip): decoded uint8 HWC images of arbitrary size ->
Resize(shorter side, bicubic) -> CenterCrop -> ToTensor -> normalize -> the NCHW bf16 / fp32 batch the encoder reads.

Replaces the CPU-worker transform chain of the reference's dataset configs (configs/dataset/cub200.yaml:31-47) for the
evaluation loop; the arithmetic is Pillow's two-pass 8-bit bicubic resampler, reproduced bit for bit on the GPU, so the
output equals ``utils.transforms``' PIL path exactly.  Host code here only derives the per-image integers whose rounding
rules are Python's (torchvision ``int(size * long / short)``, round-half-even crop origin, the first / last source row the
vertical pass touches).  PyTorch supplies device buffers and streams.
"""
from __future__ import annotations

import ctypes
import math
from typing import Sequence

import numpy as np
import torch

from . import _lib


def _resized_size(w: int, h: int, size: int):
    if w <= h:
        return size, max(1, int(size * h / w))
    return max(1, int(size * w / h)), size


def _row_bounds(in_size: int, out_size: int, xx: int):
    """(first source index, count) of output index xx -- Pillow precompute_coeffs bounds."""
    scale = float(np.float32(in_size)) / out_size
    support = 2.0 * max(scale, 1.0)
    center = (xx + 0.5) * scale
    xmin = max(int(center - support + 0.5), 0)
    xmax = min(int(center + support + 0.5), in_size) - xmin
    return xmin, xmax


def _taps(in_size: int, out_size: int) -> int:
    """Pillow's ksize for a (0, in_size) -> out_size bicubic resize: an upper bound of the taps of any output index."""
    scale = float(np.float32(in_size)) / out_size
    return int(math.ceil(2.0 * max(scale, 1.0))) * 2 + 1


class GpuPreprocess:
    """Resize(resize, bicubic) -> CenterCrop(crop) -> ToTensor -> Normalize(mean, std), on the GPU."""

    def __init__(self, resize: int = 256, crop: int = 224, mean: Sequence[float] = (0.48145466, 0.4578275, 0.40821073),
                 std: Sequence[float] = (0.26862954, 0.26130258, 0.27577711), out_dtype: torch.dtype = torch.bfloat16,
                 device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("GpuPreprocess needs a GPU (MI355X); the CPU chain is utils.transforms")
        if out_dtype not in (torch.bfloat16, torch.float32):
            raise TypeError("out_dtype must be bfloat16 or float32")
        if not (1 <= crop <= 256 and crop <= resize):
            raise ValueError("need 1 <= crop <= 256 and crop <= resize")
        self.resize, self.crop, self.out_dtype = int(resize), int(crop), out_dtype
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._mean = (ctypes.c_float * 3)(*[float(np.float32(m)) for m in mean])
        self._std = (ctypes.c_float * 3)(*[float(np.float32(s)) for s in std])
        self.max_taps = int(self.lib.ch_preprocess_max_taps())
        self._geo = {}
        self._ring = [{"buf": None, "event": None} for _ in range(self._RING)]
        self._ring_pos = 0
        self.host_routed = 0           # images done by `_host_route` so far

    _DESC_DTYPE = np.dtype([("src_offset", "<i8"), ("tmp_offset", "<i8"), ("h", "<i4"), ("w", "<i4"), ("nh", "<i4"), ("nw", "<i4"),
                            ("top", "<i4"), ("left", "<i4"), ("row0", "<i4"), ("nrows", "<i4"), ("stride", "<i4"),
                            ("flip", "<i4")])   # == ch_image_desc / _lib.ImageDesc
    _RING = 8          # pinned descriptor staging buffers in flight

    def _geometry(self, h: int, w: int):
        """(nh, nw, top, left, row0, nrows, horizontal taps) of one image size; cached -- dataset images repeat a handful of sizes."""
        key = (h, w)
        g = self._geo.get(key)
        if g is None:
            nw, nh = _resized_size(w, h, self.resize)
            left, top = int(round((nw - self.crop) / 2.0)), int(round((nh - self.crop) / 2.0))
            if any(2 * math.ceil(2.0 * max(n_in / n_out, 1.0)) + 1 > self.max_taps for n_in, n_out in ((w, nw), (h, nh))):
                # down-scaling beyond ~15x (a 12-megapixel photograph to 256): more filter taps than the kernels hold.  nrows = 0 makes the
                # kernels skip the image; __call__ runs the chain for it with Pillow on the host (`_host_route`), the reference's own path
                g = self._geo[key] = (nh, nw, top, left, 0, 0, 0)
                return g
            r0, _ = _row_bounds(h, nh, top)
            rl, cl = _row_bounds(h, nh, top + self.crop - 1)
            g = self._geo[key] = (nh, nw, top, left, r0, rl + cl - r0, _taps(w, nw))
        return g

    def plan(self, sizes: Sequence[tuple]):
        """sizes: [(h, w)] -> (descriptor array (numpy, ch_image_desc layout), total source bytes, workspace bytes, max rows,
        max horizontal taps)."""
        B = len(sizes)
        hw = np.asarray(sizes, dtype=np.int64).reshape(B, 2)
        geo = np.asarray([self._geometry(int(h), int(w)) for h, w in sizes], dtype=np.int64).reshape(B, 7)
        desc = np.zeros(B, dtype=self._DESC_DTYPE)
        src = hw[:, 0] * hw[:, 1] * 3
        tmp = geo[:, 5] * self.crop * 3
        desc["src_offset"] = np.cumsum(src) - src
        desc["tmp_offset"] = np.cumsum(tmp) - tmp
        desc["h"], desc["w"] = hw[:, 0], hw[:, 1]
        for j, name in enumerate(("nh", "nw", "top", "left", "row0", "nrows")):
            desc[name] = geo[:, j]
        return desc, int(src.sum()), int(tmp.sum()), (int(max(1, geo[:, 5].max())) if B else 1), (int(geo[:, 6].max()) if B else 0)

    def plan_boxes(self, sizes: Sequence[tuple], boxes, flips=None):
        """The training chain (configs/dataset/cub200.yaml:13-23): RandomResizedCrop(crop, bicubic) -> RandomHorizontalFlip.  boxes:
        [(top, left, box_h, box_w)] drawn on the host (utils.transforms.RandomResizedCrop.get_params, the reference's random stream),
        flips: [bool].  PIL crops the box and resizes it as an image of its own to crop x crop, so the descriptor describes the box:
        (h, w) = its size, src_offset = its first pixel, stride = the full image's width, nothing cropped afterwards."""
        B = len(sizes)
        hw = np.asarray(sizes, dtype=np.int64).reshape(B, 2)
        bx = np.asarray(boxes, dtype=np.int64).reshape(B, 4)
        if B and ((bx[:, 0] < 0).any() or (bx[:, 1] < 0).any() or (bx[:, 2] < 1).any() or (bx[:, 3] < 1).any()
                  or (bx[:, 0] + bx[:, 2] > hw[:, 0]).any() or (bx[:, 1] + bx[:, 3] > hw[:, 1]).any()):
            raise ValueError("a crop box leaves its image")
        too_big = np.asarray([2 * math.ceil(2.0 * max(max(int(b[2]), int(b[3])) / self.crop, 1.0)) + 1 > self.max_taps for b in bx], dtype=bool)
        desc = np.zeros(B, dtype=self._DESC_DTYPE)
        src = hw[:, 0] * hw[:, 1] * 3
        tmp = np.where(too_big, 0, bx[:, 2]) * self.crop * 3    # every source row of the box feeds the vertical pass
        desc["src_offset"] = np.cumsum(src) - src + (bx[:, 0] * hw[:, 1] + bx[:, 1]) * 3
        desc["tmp_offset"] = np.cumsum(tmp) - tmp
        desc["h"], desc["w"] = bx[:, 2], bx[:, 3]
        desc["nh"] = desc["nw"] = self.crop
        desc["nrows"] = np.where(too_big, 0, bx[:, 2])      # 0: skipped by the kernels, done on the host (`_host_route`)
        desc["stride"] = hw[:, 1]
        if flips is not None:
            desc["flip"] = np.asarray(flips, dtype=np.int64).reshape(B) != 0
        taps = max([_taps(int(b[3]), self.crop) for b, big in zip(bx, too_big) if not big], default=0)
        return desc, int(src.sum()), int(tmp.sum()), (int(max(1, desc["nrows"].max())) if B else 1), taps

    def _host_route(self, pixels, sizes, desc, boxes, flips, out, stream):
        """Images the kernels skipped (nrows == 0: down-scaling beyond the tap limit) through Pillow on the host -- rare (a handful of very
        large photographs in a dataset), exact (it IS the reference's chain), slow (one device -> host copy and one PIL resize per image)."""
        from PIL import Image
        mean = torch.tensor(list(self._mean)).view(3, 1, 1)
        std = torch.tensor(list(self._std)).view(3, 1, 1)
        offsets = np.cumsum([0] + [h * w * 3 for h, w in sizes])
        s = stream if stream is not None else torch.cuda.current_stream(pixels.device)
        for i in np.nonzero(desc["nrows"] == 0)[0]:
            h, w = sizes[i]
            s.synchronize()
            img = Image.fromarray(pixels[int(offsets[i]):int(offsets[i + 1])].view(h, w, 3).cpu().numpy())
            if boxes is None:
                nh, nw, top, left = (int(desc[n][i]) for n in ("nh", "nw", "top", "left"))
                img = img.resize((nw, nh), Image.BICUBIC).crop((left, top, left + self.crop, top + self.crop))
            else:
                top, left, bh, bw = (int(v) for v in np.asarray(boxes[i]).reshape(4))
                img = img.crop((left, top, left + bw, top + bh)).resize((self.crop, self.crop), Image.BICUBIC)
                if flips is not None and bool(flips[i]):
                    img = img.transpose(Image.FLIP_LEFT_RIGHT)
            x = torch.from_numpy(np.array(img, dtype=np.uint8)).permute(2, 0, 1).float().div_(255.0)
            x = (x - mean) / std
            with torch.cuda.stream(s):
                out[i].copy_(x.to(out.dtype).to(out.device, non_blocking=False))
        self.host_routed += int((desc["nrows"] == 0).sum())

    def _stage(self, desc: np.ndarray, device, stream=None) -> torch.Tensor:
        """descriptors -> device without synchronising: through a ring of pinned host buffers and a non-blocking copy (a copy from
        pageable memory would wait for everything queued on the stream -- once per batch, in the evaluator loop).  The copy and the
        ring event go on the stream `ch_preprocess` is launched on, so that the kernel can never read the descriptors before they have
        landed and a ring slot is never rewritten under a copy in flight, whatever stream the caller passes."""
        raw = torch.from_numpy(desc.view(np.uint8).reshape(-1))
        slot = self._ring[self._ring_pos % self._RING]
        self._ring_pos += 1
        if slot["event"] is not None:
            slot["event"].synchronize()          # the copy that last used this buffer (8 calls ago) has long finished
        if slot["buf"] is None or slot["buf"].numel() < raw.numel():
            slot["buf"] = torch.empty(max(raw.numel(), 48 * 256), dtype=torch.uint8, pin_memory=True)
        slot["buf"][:raw.numel()].copy_(raw)
        s = stream if stream is not None else torch.cuda.current_stream(device)
        with torch.cuda.stream(s):
            ddev = torch.empty(raw.numel(), dtype=torch.uint8, device=device)
            ddev.copy_(slot["buf"][:raw.numel()], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(s)
        slot["event"] = ev
        return ddev

    def __call__(self, pixels: torch.Tensor, sizes: Sequence[tuple], stream=None, boxes=None, flips=None) -> torch.Tensor:
        """pixels: uint8 device tensor, the images' HWC bytes back to back (image i is [h_i, w_i, 3]); sizes: [(h, w)].
        boxes / flips given: the training chain (`plan_boxes`) instead of Resize -> CenterCrop."""
        if pixels.dtype != torch.uint8 or not pixels.is_cuda:
            raise TypeError("pixels must be a uint8 GPU tensor (decoded RGB bytes, images concatenated)")
        B = len(sizes)
        out = torch.empty(B, 3, self.crop, self.crop, dtype=self.out_dtype, device=pixels.device)
        if B == 0:
            return out
        desc, nbytes, ws_bytes, max_rows, max_taps = self.plan(sizes) if boxes is None else self.plan_boxes(sizes, boxes, flips)
        if pixels.numel() != nbytes:
            raise ValueError(f"pixels holds {pixels.numel()} bytes, the sizes add up to {nbytes}")
        pixels = pixels.contiguous()
        with torch.cuda.device(pixels.device):
            ddev = self._stage(desc, pixels.device, stream)
            ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=pixels.device)
            _lib.check(self.lib.ch_preprocess(_lib.ptr(pixels), _lib.ptr(ddev), B, max_rows, max_taps, self.crop, self._mean, self._std,
                                              _lib.ptr(out), 1 if self.out_dtype == torch.bfloat16 else 0, _lib.ptr(ws),
                                              _lib.stream_ptr(stream)), "ch_preprocess")
            if (desc["nrows"] == 0).any():
                self._host_route(pixels, sizes, desc, boxes, flips, out, stream)
        return out
