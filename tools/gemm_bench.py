#!/usr/bin/env python3
"""Micro-benchmark of the GEMM kernel variants on the encoder's shapes (random data, interleaved rounds in one process).
    python tools/gemm_bench.py [--rows 51456] [--rounds 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from concepthash_amd import _lib

SHAPES = [("qkv", 2304, 768, 0), ("out", 768, 768, 3), ("down", 384, 768, 2), ("up", 768, 384, 4), ("fc1", 3072, 768, 1),
          ("fc2", 768, 3072, 3)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=51456)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--variants", default="1,2")
    a = ap.parse_args()
    lib = _lib.load()
    M = a.rows
    Mp = (M + 255) // 256 * 256
    variants = [int(v) for v in a.variants.split(",")]
    scale = torch.tensor([0.5], device="cuda")
    for name, N, K, epi in SHAPES:
        X = torch.randn(Mp, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        out = torch.empty(Mp, N, dtype=torch.bfloat16, device="cuda")
        resid = torch.zeros(Mp, N, device="cuda")
        times = {v: [] for v in variants}
        for r in range(a.rounds + 2):
            for v in variants:
                if v in (2, 5, 21, 22, 23, 24, 25, 26, 27) and (N % 256 or K % 128):
                    continue
                if v == 5 and epi > 2:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.ch_debug_gemm(v, _lib.ptr(X), Mp, _lib.ptr(W), _lib.ptr(bias), M, N, K, epi, _lib.ptr(out), N,
                                             _lib.ptr(resid), N, _lib.ptr(scale), None, _lib.stream_ptr()), "gemm")
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    times[v].append(e0.elapsed_time(e1))
        fl = 2.0 * M * N * K
        msg = f"{name:5s} M={M} N={N:5d} K={K:5d} epi={epi}:"
        for v in variants:
            if times[v]:
                t = sorted(times[v])
                med, mn = t[len(t) // 2], t[0]
                msg += f"  v{v}: med {med * 1e3:7.1f} us {fl / med / 1e9:7.1f} TF (min {mn * 1e3:7.1f} us {fl / mn / 1e9:7.1f} TF)"
        print(msg, flush=True)


if __name__ == "__main__":
    main()
