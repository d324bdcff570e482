#!/usr/bin/env python3
"""Host half of the JPEG split alone: entropy-decode throughput vs thread count on this box (no GPU work), plus what the box grants."""
import ctypes, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
from concepthash_amd import _lib
from concepthash_amd.jpeg import DESC_DTYPE, GpuJpegDecoder


def main():
    lib = _lib.load()
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        if os.path.exists(f):
            print(f, open(f).read().strip())
    print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), flush=True)
    rng = np.random.default_rng(0)
    files = []
    for i in range(256):
        low = rng.integers(0, 256, (12 + i % 13, 16 + i % 11, 3), dtype=np.uint8)
        img = np.asarray(Image.fromarray(low).resize((500, 375), Image.BICUBIC), dtype=np.int16)
        bio = io.BytesIO()
        Image.fromarray(np.clip(img + rng.normal(0, 7, img.shape), 0, 255).astype(np.uint8)).save(bio, "JPEG", quality=85)
        files.append(np.frombuffer(bio.getvalue(), dtype=np.uint8))
    n = len(files)
    ptrs = (ctypes.c_void_p * n)(*[b.ctypes.data for b in files])
    lens = (ctypes.c_int64 * n)(*[b.size for b in files])
    desc = np.zeros(n, dtype=DESC_DTYPE)
    lib.ch_jpeg_plan(ptrs, lens, n, desc.ctypes.data, None, None, None)
    total, _, _ = GpuJpegDecoder.layout(desc)
    coef = np.ones(total, np.int16)
    print("mean file KiB", sum(b.size for b in files) / n / 1024, "coef MB per batch", total * 2 / 1e6, flush=True)
    for nt in (1, 2, 4, 8, 12, 16, 24, 32):
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            lib.ch_jpeg_entropy_decode(ptrs, lens, n, desc.ctypes.data, coef.ctypes.data, nt)
            best = min(best, time.perf_counter() - t0)
        print(f"threads {nt:2d}: best of 5 {best * 1e3:7.2f} ms per 256 images = {n / best:8.0f} images/s = {n / best / nt:6.0f} per thread", flush=True)
    t0 = time.perf_counter()
    for f in files[:64]:
        np.asarray(Image.open(io.BytesIO(f.tobytes())).convert("RGB"))
    print(f"PIL full decode, 1 thread: {64 / (time.perf_counter() - t0):.0f} images/s", flush=True)


if __name__ == "__main__":
    main()
