#!/usr/bin/env python3
"""Where does a 256x256 tile of the ping-pong GEMM spend its time?  Runs the stamped build of the kernel (ch_debug_gemm variant
28 / 29: s_memtime at entry, after the prologue, after the K loop, after the last epilogue store is issued, after the stores
completed; HW_ID / XCC_ID; s_memrealtime at entry and exit) on the encoder's shapes and prints per-phase statistics plus the gap
between consecutive tiles on the same CU.      python tools/tile_timeline.py"""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from concepthash_amd import _lib

lib = _lib.load()
M = 51456
Mp = (M + 255) // 256 * 256
for name, N, K, variant in (("qkv-like (bias)", 2304, 768, 28), ("fc1-like (bias + quick_gelu)", 3072, 768, 29),
                            ("out-like (bias)", 768, 768, 28), ("fc2-like (bias)", 768, 3072, 28)):
    X = torch.randn(Mp, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(Mp, N, dtype=torch.bfloat16, device="cuda")
    tiles = (Mp // 256) * (N // 256)
    stamps = torch.zeros(tiles, 8, dtype=torch.int64, device="cuda")
    for it in range(3):
        _lib.check(lib.ch_debug_gemm(variant, _lib.ptr(X), Mp, _lib.ptr(W), _lib.ptr(bias), M, N, K, 0, _lib.ptr(out), N,
                                     _lib.ptr(stamps), N, None, None, _lib.stream_ptr()), "gemm")
        torch.cuda.synchronize()
    st = stamps.cpu().numpy().astype(np.int64)
    rt0, t_in, t_pro, t_loop, t_epi, t_done, hw, rt1 = [st[:, i] for i in range(8)]
    wall_us = (rt1.max() - rt0.min()) / 100.0            # s_memrealtime ticks at 100 MHz
    clk = np.median((t_done - t_in) / np.maximum(rt1 - rt0, 1)) * 100e6 / 1e9   # shader GHz
    us = lambda c: c / (clk * 1e3)
    print(f"\n== {name}: N {N} K {K}, {tiles} tiles, launch {wall_us:.1f} us wall, shader clock {clk:.2f} GHz")
    for label, d in (("prologue (entry -> first K-tile staged + barrier)", t_pro - t_in), ("K loop", t_loop - t_pro),
                     ("epilogue until the last store is issued", t_epi - t_loop), ("stores in flight -> completed", t_done - t_epi),
                     ("whole tile", t_done - t_in)):
        d = us(d.astype(np.float64))
        print(f"   {label:52s} median {np.median(d):6.2f} us   p10 {np.percentile(d, 10):6.2f}   p90 {np.percentile(d, 90):6.2f}")
    # consecutive tiles on the same CU: key = XCC id + SE/SH/CU bits of HW_ID (wave / SIMD / pipe bits masked out)
    key = ((hw >> 32) << 16) | ((hw & 0xFFFFFFFF) >> 8 & 0xFFFF)
    per_cu = defaultdict(list)
    for i in range(tiles):
        per_cu[int(key[i])].append((int(rt0[i]), int(rt1[i])))
    gaps, counts = [], []
    for k, lst in per_cu.items():
        lst.sort()
        counts.append(len(lst))
        gaps += [(lst[i + 1][0] - lst[i][1]) / 100.0 for i in range(len(lst) - 1)]
    gaps = np.array(gaps)
    print(f"   distinct CU keys {len(per_cu)} (tiles per key: min {min(counts)} median {int(np.median(counts))} max {max(counts)}); "
          f"gap between a tile's end and the next tile's entry on the same CU: median {np.median(gaps):.2f} us, p90 {np.percentile(gaps, 90):.2f} us")
    order = np.argsort(rt0)
    starts = (rt0[order] - rt0.min()) / 100.0
    print(f"   tile entries: first wave of {min(256, tiles)} tiles enters within {starts[min(255, tiles - 1)]:.2f} us; last tile enters at {starts[-1]:.1f} us")
