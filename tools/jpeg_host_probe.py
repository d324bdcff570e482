#!/usr/bin/env python3
"""Host half of the JPEG split alone: entropy-decode throughput vs thread count on this box (no GPU work), plus what the box grants."""
import ctypes, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
from concepthash_amd import _lib
from concepthash_amd.jpeg import DESC_DTYPE, GpuJpegDecoder


def _timed(lib, ptrs, lens, n, desc, dst_ptr, nt, reps=6):
    """(best wall ms, user CPU ms, system CPU ms of the best-of run's process) for one entropy decode of the batch."""
    import resource
    best = (1e9, 0.0, 0.0)
    for _ in range(reps):
        r0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        lib.ch_jpeg_entropy_decode(ptrs, lens, n, desc.ctypes.data, dst_ptr, nt)
        dt = time.perf_counter() - t0
        r1 = resource.getrusage(resource.RUSAGE_SELF)
        if dt < best[0]:
            best = (dt, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime)
    return tuple(v * 1e3 for v in best)


def gpu_section(lib, ptrs, lens, n, desc, total):
    """The same call inside a process that has initialised the GPU: plain destination, pinned destination, and with copies + kernels
    running beside it -- to see which of those makes the in-pipeline decode slower than the stand-alone one."""
    import threading
    import torch
    plain = np.ones(total, np.int16)
    print("before GPU init, plain dst, 16 threads: wall %.2f ms user %.1f ms sys %.1f ms" % _timed(lib, ptrs, lens, n, desc, plain.ctypes.data, 16), flush=True)
    torch.cuda.init()
    dev = torch.device("cuda:0")
    x = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    print("after GPU init, plain dst: wall %.2f ms user %.1f ms sys %.1f ms" % _timed(lib, ptrs, lens, n, desc, plain.ctypes.data, 16), flush=True)
    pinned = torch.empty(total, dtype=torch.int16, pin_memory=True)
    pinned.fill_(1)
    print("after GPU init, pinned dst: wall %.2f ms user %.1f ms sys %.1f ms" % _timed(lib, ptrs, lens, n, desc, pinned.data_ptr(), 16), flush=True)
    dcoef = torch.empty(total, dtype=torch.int16, device=dev)
    stop = False

    def feeder():                       # what the pipeline's consumer does meanwhile: H2D copies of the previous batch + kernels
        a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
        other = torch.empty(total, dtype=torch.int16, pin_memory=True)
        while not stop:
            dcoef.copy_(other, non_blocking=True)
            for _ in range(20):
                a @ a
            torch.cuda.synchronize()

    th = threading.Thread(target=feeder)
    th.start()
    time.sleep(0.5)
    print("pinned dst, copies + GEMMs beside it: wall %.2f ms user %.1f ms sys %.1f ms" % _timed(lib, ptrs, lens, n, desc, pinned.data_ptr(), 16), flush=True)
    print("plain dst, copies + GEMMs beside it: wall %.2f ms user %.1f ms sys %.1f ms" % _timed(lib, ptrs, lens, n, desc, plain.ctypes.data, 16), flush=True)
    stop = True
    th.join()
    # sources in a freshly mapped shared-memory tensor (what a loader worker hands over)
    blob = torch.from_numpy(np.concatenate([np.ctypeslib.as_array((ctypes.c_uint8 * lens[i]).from_address(ptrs[i])) for i in range(n)]))
    offs = np.zeros(n + 1, np.int64); np.cumsum(np.asarray(list(lens), np.int64), out=offs[1:])
    import resource

    def one(tag, src_tensor):
        r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
        lib.ch_jpeg_entropy_decode_packed(src_tensor.data_ptr(), offs.ctypes.data, n, desc.ctypes.data, pinned.data_ptr(), 16)
        dt = time.perf_counter() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF)
        print("%-58s wall %6.2f ms user %6.1f ms sys %6.1f ms" % (tag, dt * 1e3, (r1.ru_utime - r0.ru_utime) * 1e3, (r1.ru_stime - r0.ru_stime) * 1e3), flush=True)

    print("torch threads", torch.get_num_threads(), flush=True)
    for rep in range(4):
        one("private source (the blob itself)", blob)
    for rep in range(4):
        c = blob.clone()
        one("fresh private clone", c)
    for rep in range(4):
        c = blob.clone()
        time.sleep(0.05)
        one("fresh private clone, 50 ms after the clone", c)
    torch.set_num_threads(1)
    for rep in range(4):
        c = blob.clone()
        one("fresh private clone, torch.set_num_threads(1)", c)
    sh = blob.clone().share_memory_()
    for rep in range(4):
        one("one shared-memory source, reused", sh)
    for rep in range(4):
        sh = blob.clone().share_memory_()
        one("fresh shared-memory source", sh)


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
    if "--gpu" in sys.argv:
        gpu_section(lib, ptrs, lens, n, desc, total)
    t0 = time.perf_counter()
    for f in files[:64]:
        np.asarray(Image.open(io.BytesIO(f.tobytes())).convert("RGB"))
    print(f"PIL full decode, 1 thread: {64 / (time.perf_counter() - t0):.0f} images/s", flush=True)


if __name__ == "__main__":
    main()
