"""JPEG decode split for the evaluation loader: host threads entropy-decode, the GPU reconstructs.

Reference: the loader workers of `engine.dataloader` (engine.py:41-54) run `PIL.Image.open(path).convert("RGB")` per item -- a full
libjpeg-turbo decode on a CPU core, ~1-2 ms per 500 x 375 image -- and then the transform chain of configs/dataset/cub200.yaml:31-47.
A 20k images/s encoder outruns that on any host, so here the loader only READS the files (in-process, `engine.FileBatchLoader`);
`GpuJpegDecoder` then
  1. parses the headers and Huffman-decodes the entropy-coded segments on a pool of host threads (`ch_jpeg_plan`,
     `ch_jpeg_entropy_decode`: plain C++ inside the library, the GIL is released for the whole call) straight into pinned memory,
  2. copies the int16 coefficient blocks to the device (non-blocking) and
  3. runs dequantisation + inverse DCT + chroma upsampling + YCbCr -> RGB on the GPU (`ch_jpeg_reconstruct`),
leaving decoded RGB bytes in the layout `GpuPreprocess` / `ch_preprocess` consume.  The bytes are BIT-EQUAL to Pillow's
(tests/test_jpeg.py).  Baseline and progressive Huffman files are covered; files outside the supported subset (CMYK, arithmetic coding,
exotic sampling, truncated data, ...: `ch_jpeg_desc.status != 0`) are decoded with PIL on the host -- the reference's own path -- and copied
into their slots; `stats` counts them.  There is no CPU fallback for the
supported files: without the HIP library or a GPU, construction raises."""
from __future__ import annotations

import ctypes
import io
import os
import time
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

DESC_DTYPE = np.dtype([("coef_offset", "<i8"), ("pix_offset", "<i8"), ("plane_offset", "<i8"), ("width", "<i4"), ("height", "<i4"),
                       ("ncomp", "<i4"), ("hs", "<i4"), ("vs", "<i4"), ("mcu_w", "<i4"), ("mcu_h", "<i4"), ("status", "<i4"),
                       ("nblocks", "<i4"), ("reserved", "<i4"), ("quant", "<u2", (3, 64))])   # == ch_jpeg_desc / _lib.JpegDesc
assert DESC_DTYPE.itemsize == ctypes.sizeof(_lib.JpegDesc)

STATUS = {0: "ok", 1: "not a JPEG", 2: "truncated", 3: "lossless / arithmetic / unfinished progressive", 4: "not 8 bit", 5: "component count",
          6: "sampling factors", 7: "multi-scan", 8: "colour space", 9: "tables", 10: "smaller than 16x16", 11: "corrupt entropy data",
          12: "beyond Pillow's decompression-bomb limit"}


_libc = None


def _dont_fork(buf: torch.Tensor) -> None:
    """madvise(MADV_DONTFORK) on a pinned staging buffer.  DataLoader workers are FORKED from the trainer process; without this every page
    of the 150 MB coefficient buffer the entropy decoder rewrites per batch is a copy-on-write fault while workers are alive -- measured
    on the MI355X box: 98 images/s through the loader against 31.8k images/s for the same decode with no worker process alive."""
    global _libc
    try:
        if _libc is None:
            _libc = ctypes.CDLL(None, use_errno=True)
            _libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        page = os.sysconf("SC_PAGE_SIZE")
        lo = (buf.data_ptr() + page - 1) // page * page
        hi = (buf.data_ptr() + buf.numel()) // page * page
        if hi > lo:
            _libc.madvise(ctypes.c_void_p(lo), ctypes.c_size_t(hi - lo), 10)     # MADV_DONTFORK
    except Exception:        # advisory only
        pass


def default_threads() -> int:
    """Entropy-decode threads: what the container's CPU quota allows (hostcpu.cpu_budget), at most 16."""
    from .hostcpu import cpu_budget
    return max(1, min(16, cpu_budget()))


def _as_bytes_array(f) -> np.ndarray:
    if isinstance(f, np.ndarray):
        return np.ascontiguousarray(f, dtype=np.uint8).reshape(-1)
    if torch.is_tensor(f):
        return f.contiguous().view(-1).numpy()
    return np.frombuffer(f, dtype=np.uint8)


class GpuJpegDecoder:
    """files (bytes / uint8 arrays) -> (pixels: uint8 device tensor, all images' RGB bytes back to back; sizes: [(h, w)])."""

    _RING = 4   # pinned staging sets in flight (a set is reused only after the copies that read it have completed)

    def __init__(self, device=None, threads: Optional[int] = None, strict: bool = False):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("GpuJpegDecoder needs a GPU (MI355X); there is no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.threads = int(threads) if threads else default_threads()
        self.pretouch = False          # diagnostic (tools/loader_probe.py --pretouch): time the first touch of a worker's batch separately
        self.strict = bool(strict)     # True: a file outside the supported subset raises instead of going through PIL
        self._ring = [dict(coef=None, desc=None, fb=None, event=None, coef_dev=None, desc_dev=None, copied=None) for _ in range(self._RING)]
        self._pos = 0
        self._pool = None              # thread pool of the PIL route (files outside the supported subset), created on first use
        self._copy_stream = None       # host -> device copies of a staged batch run here, beside the caller's kernels (SDMA engine)
        self.stats = {"images": 0, "gpu": 0, "pil_fallback": 0, "fallback_reasons": {}, "plan_s": 0.0, "ring_wait_s": 0.0, "entropy_s": 0.0,
                      "enqueue_s": 0.0, "touch_s": 0.0}

    # -- host half ------------------------------------------------------------------------------------------------
    def plan(self, files: Sequence) -> Tuple[np.ndarray, list, ctypes.Array, ctypes.Array]:
        """-> (descriptor array viewed as numpy, keep-alive buffers, pointer array, length array)"""
        n = len(files)
        bufs = [_as_bytes_array(f) for f in files]
        ptrs = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bufs])
        lens = (ctypes.c_int64 * n)(*[b.size for b in bufs])
        desc = np.zeros(n, dtype=DESC_DTYPE)
        _lib.check(self.lib.ch_jpeg_plan(ptrs, lens, n, desc.ctypes.data, None, None, None), "ch_jpeg_plan")
        return desc, bufs, ptrs, lens

    @staticmethod
    def layout(desc: np.ndarray) -> Tuple[int, int, int]:
        """back-to-back offsets: coefficients / planes of the images this path decodes, an output slot for EVERY image"""
        ok = desc["status"] == 0
        nco = np.where(ok, desc["nblocks"].astype(np.int64) * 64, 0)
        npl = (nco + 15) // 16 * 16
        npx = desc["width"].astype(np.int64) * desc["height"].astype(np.int64) * 3
        desc["coef_offset"] = np.cumsum(nco) - nco
        desc["plane_offset"] = np.cumsum(npl) - npl
        desc["pix_offset"] = np.cumsum(npx) - npx
        return int(nco.sum()), int(npx.sum()), int(npl.sum())

    def _pil(self, data: np.ndarray) -> np.ndarray:
        from PIL import Image
        return np.asarray(Image.open(io.BytesIO(data.tobytes())).convert("RGB"), dtype=np.uint8)

    def _slot(self):
        slot = self._ring[self._pos % self._RING]
        self._pos += 1
        if slot.get("busy"):
            raise RuntimeError("GpuJpegDecoder: more than _RING - 1 staged batches are waiting for device_stage (bound the prefetch depth)")
        if slot["event"] is not None:
            slot["event"].synchronize()
        slot["busy"] = True
        return slot

    @staticmethod
    def _pinned(slot, key, nbytes):
        buf = slot[key]
        if buf is None or buf.numel() < nbytes:
            buf = slot[key] = torch.empty(max(int(nbytes * 1.25), 1 << 16), dtype=torch.uint8, pin_memory=True)
            _dont_fork(buf)
        return buf

    def _device(self, slot, key, nbytes):
        buf = slot[key]
        if buf is None or buf.numel() < nbytes:
            buf = slot[key] = torch.empty(max(int(nbytes * 1.25), 1 << 16), dtype=torch.uint8, device=self.device)
        return buf

    def _upload(self, slot, desc_nbytes, total_coef):
        """Last step of the host half: the slot's descriptors and coefficient blocks go to the slot's DEVICE buffers on the decoder's own
        copy stream, as soon as they exist -- not when the consumer gets round to the batch.  The copy (151 MB for 256 images of
        500 x 375, ~2.7 ms over PCIe) then runs on the DMA engine under the previous batch's encode instead of in front of this one's
        kernels on the caller's stream.  `device_stage` makes its stream wait for `slot["copied"]`."""
        with torch.cuda.device(self.device):
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(self.device)
            cs = self._copy_stream
            with torch.cuda.stream(cs):
                self._device(slot, "desc_dev", desc_nbytes)[:desc_nbytes].copy_(slot["desc"][:desc_nbytes], non_blocking=True)
                if total_coef:
                    self._device(slot, "coef_dev", total_coef * 2)[:total_coef * 2].copy_(slot["coef"][:total_coef * 2], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(cs)
        slot["copied"] = ev

    # -- decode -----------------------------------------------------------------------------------------------------
    def _host_stage_packed(self, data: torch.Tensor, lengths) -> Optional["StagedJpegBatch"]:
        """The common case in two GIL-free calls: a batch whose files sit back to back in one buffer (RawJpegBatch) and all belong to the
        supported subset.  Returns None when some file needs the PIL path (the caller then takes the general route)."""
        n = len(lengths)
        offsets = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.asarray(lengths, dtype=np.int64), out=offsets[1:])
        base = data.data_ptr()
        desc = np.zeros(n, dtype=DESC_DTYPE)
        tc, tp, tl = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        t0 = time.perf_counter()
        _lib.check(self.lib.ch_jpeg_plan_packed(base, offsets.ctypes.data, n, desc.ctypes.data, ctypes.byref(tc), ctypes.byref(tp),
                                                ctypes.byref(tl)), "ch_jpeg_plan_packed")
        if desc["status"].any():
            return None
        totals = (tc.value, tp.value, tl.value)
        t1 = time.perf_counter()
        with torch.cuda.device(self.device):
            slot = self._slot()
            coef_host = self._pinned(slot, "coef", max(totals[0], 1) * 2)
            desc_host = self._pinned(slot, "desc", desc.nbytes)
        if self.pretouch:                              # diagnostic: fault the worker's shared-memory pages in before the threads start
            t_touch = time.perf_counter()
            int(data.numpy()[::4096].sum())
            self.stats["touch_s"] += time.perf_counter() - t_touch
        t2 = time.perf_counter()
        _lib.check(self.lib.ch_jpeg_entropy_decode_packed(base, offsets.ctypes.data, n, desc.ctypes.data, coef_host.data_ptr(), self.threads),
                   "ch_jpeg_entropy_decode_packed")
        t3 = time.perf_counter()
        fallback = {}
        if desc["status"].any():                       # a corrupt stream (rare): PIL for that file, its slot keeps its size
            bufs = [data[int(offsets[i]):int(offsets[i + 1])].numpy() for i in range(n)]
            self._fallback([int(i) for i in np.nonzero(desc["status"])[0]], desc, bufs, fallback, keep_size=True)
        sizes = list(zip(desc["height"].tolist(), desc["width"].tolist()))
        desc_host.numpy()[:desc.nbytes] = desc.view(np.uint8).reshape(-1)     # numpy, not torch copy_: no OpenMP team on this thread
        fb = None
        if fallback:
            with torch.cuda.device(self.device):
                fb = self._pinned(slot, "fb", sum(a.size for a in fallback.values()))
            o = 0
            for a in fallback.values():
                fb.numpy()[o:o + a.size] = a.reshape(-1)
                o += a.size
        st = self.stats
        st["plan_s"] += t1 - t0
        st["ring_wait_s"] += t2 - t1
        st["entropy_s"] += t3 - t2
        self._upload(slot, desc.nbytes, totals[0])
        return StagedJpegBatch(self, slot, desc, fallback, totals, sizes, fb)

    def host_stage(self, files) -> "StagedJpegBatch":
        staged = self._host_stage(files)
        staged.boxes, staged.flips = getattr(files, "boxes", None), getattr(files, "flips", None)
        return staged

    def _host_stage(self, files) -> "StagedJpegBatch":
        """HOST half: headers, PIL for the files outside the subset, offsets, Huffman decode into a pinned ring slot.  No GPU call except
        waiting for the slot's previous copies: may run on a background thread one or two batches ahead of `device_stage`
        (`prefetch_decoded`); the ring holds `_RING` slots, so at most `_RING - 1` staged batches may wait for their device stage.
        files: a sequence of bytes / uint8 arrays, or an object with `.data` (one uint8 tensor, files back to back) and `.lengths`."""
        if hasattr(files, "data") and hasattr(files, "lengths") and torch.is_tensor(files.data):
            staged = self._host_stage_packed(files.data.contiguous(), files.lengths) if len(files.lengths) else None
            if staged is not None:
                return staged
            files = files.files
        n = len(files)
        if n == 0:
            return StagedJpegBatch(self, None, None, {}, (0, 0, 0), [], None)
        t0 = time.perf_counter()
        desc, bufs, ptrs, lens = self.plan(files)
        fallback = {}                                  # index -> decoded RGB array (PIL)
        self._fallback([int(i) for i in np.nonzero(desc["status"])[0]], desc, bufs, fallback)
        totals = self.layout(desc)
        t1 = time.perf_counter()
        with torch.cuda.device(self.device):           # (a prefetch thread starts on device 0: events / pinned buffers belong to OUR device)
            slot = self._slot()
            coef_host = self._pinned(slot, "coef", max(totals[0], 1) * 2)
        t2 = time.perf_counter()
        _lib.check(self.lib.ch_jpeg_entropy_decode(ptrs, lens, n, desc.ctypes.data, coef_host.data_ptr(), self.threads),
                   "ch_jpeg_entropy_decode")
        t3 = time.perf_counter()
        # streams that turned out corrupt: their slots keep their sizes
        self._fallback([int(i) for i in np.nonzero(desc["status"])[0] if int(i) not in fallback], desc, bufs, fallback, keep_size=True)
        sizes = [(int(h), int(w)) for h, w in zip(desc["height"], desc["width"])]
        with torch.cuda.device(self.device):
            desc_host = self._pinned(slot, "desc", desc.nbytes)
            fb = self._pinned(slot, "fb", sum(a.size for a in fallback.values())) if fallback else None
        desc_host.numpy()[:desc.nbytes] = desc.view(np.uint8).reshape(-1)     # numpy, not torch copy_: no OpenMP team on this thread
        if fallback:
            o = 0
            for a in fallback.values():
                fb.numpy()[o:o + a.size] = a.reshape(-1)
                o += a.size
        st = self.stats
        st["plan_s"] += t1 - t0
        st["ring_wait_s"] += t2 - t1
        st["entropy_s"] += t3 - t2
        self._upload(slot, desc.nbytes, totals[0])
        return StagedJpegBatch(self, slot, desc, fallback, totals, sizes, fb)

    def device_stage(self, staged: "StagedJpegBatch", stream=None):
        """GPU half: wait (on the stream) for the host half's uploads, `ch_jpeg_reconstruct` from the slot's device buffers, the
        PIL-decoded files into their slots.  -> (pixels, sizes)."""
        dev = self.device
        n = len(staged.sizes)
        if n == 0:
            return torch.empty(0, dtype=torch.uint8, device=dev), []
        t3 = time.perf_counter()
        slot, desc, fallback = staged.slot, staged.desc, staged.fallback
        total_coef, total_pix, total_plane = staged.totals
        s = stream if stream is not None else torch.cuda.current_stream(dev)
        with torch.cuda.device(dev), torch.cuda.stream(s):
            pixels = torch.empty(max(total_pix, 1), dtype=torch.uint8, device=dev)
            desc_host = slot["desc"]
            s.wait_event(slot["copied"])                     # the host half's uploads (copy stream)
            if total_coef:
                planes = torch.empty(max(total_plane, 16), dtype=torch.uint8, device=dev)
                _lib.check(self.lib.ch_jpeg_reconstruct(_lib.ptr(slot["coef_dev"]), _lib.ptr(slot["desc_dev"]), desc_host.data_ptr(), n,
                                                        _lib.ptr(planes), _lib.ptr(pixels), _lib.stream_ptr(s)), "ch_jpeg_reconstruct")
            o = 0
            for i, a in fallback.items():
                po = int(desc["pix_offset"][i])
                pixels[po:po + a.size].copy_(staged.fb[o:o + a.size], non_blocking=True)
                o += a.size
            ev = torch.cuda.Event()
            ev.record(s)
        slot["event"] = ev
        slot["busy"] = False
        st = self.stats
        st["enqueue_s"] += time.perf_counter() - t3
        st["images"] += n
        st["gpu"] += n - len(fallback)
        st["pil_fallback"] += len(fallback)
        return pixels[:total_pix], staged.sizes

    def decode(self, files: Sequence, stream=None):
        return self.device_stage(self.host_stage(files), stream)

    def _fallback(self, indices, desc, bufs, fallback, keep_size=False):
        """The files outside the supported subset (or with corrupt entropy data) through PIL, the reference's own decoder.  More than a
        couple of them are decoded on the decoder's thread pool (PIL releases the GIL while it decodes): a dataset with many such
        files must not fall back to one image at a time on the host-stage thread."""
        if not indices:
            return
        for i in indices:
            if self.strict:
                reason = STATUS.get(int(desc["status"][i]), str(int(desc["status"][i])))
                raise ValueError(f"image {i}: outside the GPU JPEG subset ({reason}) and strict=True")
        if len(indices) > 2 and self.threads > 1:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=self.threads, thread_name_prefix="jpeg-pil")
            decoded = list(self._pool.map(lambda i: self._pil(bufs[i]), indices))
        else:
            decoded = [self._pil(bufs[i]) for i in indices]
        r = self.stats["fallback_reasons"]
        for i, a in zip(indices, decoded):
            if keep_size and (a.shape[0] != int(desc["height"][i]) or a.shape[1] != int(desc["width"][i])):
                raise ValueError(f"image {i}: corrupt entropy data and PIL decodes it to another size")
            desc["height"][i], desc["width"][i] = a.shape[0], a.shape[1]
            fallback[i] = a
            reason = STATUS.get(int(desc["status"][i]), str(int(desc["status"][i])))
            r[reason] = r.get(reason, 0) + 1

    __call__ = decode


class StagedJpegBatch:
    """A batch after the host half of the decode (coefficients in a pinned ring slot), waiting for `device_stage`."""

    def __init__(self, decoder, slot, desc, fallback, totals, sizes, fb):
        self.decoder, self.slot, self.desc, self.fallback, self.totals, self.sizes, self.fb = decoder, slot, desc, fallback, totals, sizes, fb
        self.staged = True
        self.boxes = self.flips = None      # a training dataset's crop boxes / flips ride along (utils.datasets.RawJpegBatch)

    def size(self, dim=0):
        if dim != 0:
            raise IndexError("StagedJpegBatch only has a batch dimension")
        return len(self.sizes)

    def to(self, device, non_blocking=False):
        return self

    def finish(self, stream=None):
        return self.decoder.device_stage(self, stream)


def prefetch_decoded(loader, decoder: GpuJpegDecoder, depth: int = 2):
    """Iterate a `gpu_decode` loader as a three-stage pipeline: a FETCH thread takes batches from the loader's workers (unpickling, mapping
    the shared-memory segment), a HOST-STAGE thread turns them into pinned coefficient blocks (header parse + Huffman decode: GIL-free C
    with its own thread pool), up to `depth` batches ahead, and the caller's thread enqueues copies and kernels: yields
    (StagedJpegBatch, targets, indices).  The step time becomes max(fetch, host decode, GPU) instead of their sum.
    depth <= GpuJpegDecoder._RING - 2."""
    import queue
    import threading
    depth = max(1, min(int(depth), decoder._RING - 2))
    q_raw, q_staged = queue.Queue(maxsize=2), queue.Queue(maxsize=depth)
    stop = threading.Event()

    def put(q, item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def fetch():
        try:
            for item in loader:
                if not put(q_raw, item):
                    return
            put(q_raw, None)
        except BaseException as e:      # re-raised in the consumer
            put(q_raw, e)

    def stage():
        try:
            while not stop.is_set():
                try:
                    item = q_raw.get(timeout=0.1)
                except queue.Empty:
                    continue
                if item is None or isinstance(item, BaseException):
                    put(q_staged, item)
                    return
                raw, targets, index = item
                if not put(q_staged, (decoder.host_stage(raw), targets, index)):
                    return
        except BaseException as e:
            put(q_staged, e)

    threads = [threading.Thread(target=fetch, name="jpeg-fetch", daemon=True), threading.Thread(target=stage, name="jpeg-host-stage", daemon=True)]
    for th in threads:
        th.start()
    try:
        while True:
            item = q_staged.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:
        stop.set()
        for th in threads:
            th.join(timeout=5.0)
        for slot in decoder._ring:      # staged batches that were never device-staged (early exit): their slots are free again
            slot["busy"] = False


def decode_to_list(decoder: GpuJpegDecoder, files: Sequence) -> List[torch.Tensor]:
    """Convenience for tests: one [h, w, 3] uint8 device tensor per file."""
    pixels, sizes = decoder.decode(files)
    out, o = [], 0
    for h, w in sizes:
        out.append(pixels[o:o + h * w * 3].view(h, w, 3))
        o += h * w * 3
    return out
