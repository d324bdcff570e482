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
    ap.add_argument("--variants", default="1,2", help="comma list; '5s12000' = variant 5 with CH_PPP_SKEW_NS=12000, '5f1' = CH_PPP_FLAGS=1")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--custom", default="", help="extra shape name:N:K:epi, e.g. d512:512:768:2")
    ap.add_argument("--epi", type=int, default=-1, help="override the epilogue of every shape")
    a = ap.parse_args()
    lib = _lib.load()
    M = a.rows
    Mp = (M + 255) // 256 * 256
    import re
    variants = a.variants.split(",")

    def parse(tok):
        m = re.fullmatch(r"(\d+)(?:s(\d+))?(?:f(\d+))?(?:e(\d+))?(?:d(\d+))?(?:k(\d))?", tok)
        return int(m.group(1)), m.group(2) or "0", m.group(3) or "0", int(m.group(4)) if m.group(4) else None, m.group(5) or "0", int(m.group(6) or 0)
    scale = torch.tensor([0.5], device="cuda")
    shapes = list(SHAPES)
    if a.custom:
        nm, n_, k_, e_ = a.custom.split(":")
        shapes.append((nm, int(n_), int(k_), int(e_)))
    for name, N, K, epi in shapes:
        if a.shapes and name not in a.shapes.split(","):
            continue
        if a.epi >= 0:
            epi = a.epi
        X = torch.randn(Mp, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        out = torch.empty(Mp, N, dtype=torch.bfloat16, device="cuda")
        resid = torch.zeros(Mp, N, device="cuda")
        stats_in = torch.rand(Mp, K // 64, 2, device="cuda") * 64
        stats_in[..., 1] += 64
        stats_out = torch.zeros(Mp, N // 64, 2, device="cuda")
        fold_c = torch.randn(N, device="cuda")
        hb = torch.empty(Mp, N, dtype=torch.bfloat16, device="cuda")
        addend = torch.zeros(Mp, N, dtype=torch.bfloat16, device="cuda")
        times = {v: [] for v in variants}
        for r in range(a.rounds + 2):
            for vt in variants:
                v, skew, flags, epi_v, dbg, sk = parse(vt)
                lib.ch_debug_set_gemm_splitk(sk)
                ep = epi if epi_v is None else epi_v
                os.environ["CH_PPP_SKEW_NS"], os.environ["CH_PPP_FLAGS"] = skew, flags
                if v == 6 and (N % 128 or K % 128):
                    continue
                if v in (2, 4, 5, 8, 21, 22, 23, 24, 25, 26, 27) and (N % 256 or K % 128):
                    continue
                if v == 5 and ep not in (0, 1, 2, 6, 8, 9, 10):
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if ep >= 6:
                    _lib.check(lib.ch_debug_gemm_ln(v, _lib.ptr(X), Mp, _lib.ptr(W), _lib.ptr(bias), M, N, K, ep, _lib.ptr(out), N,
                                                    _lib.ptr(resid), N, _lib.ptr(scale), _lib.ptr(addend), _lib.ptr(stats_in),
                                                    _lib.ptr(fold_c), 1e-5, _lib.ptr(stats_out), _lib.ptr(hb),
                                                    _lib.stream_ptr()), "gemm_ln")
                else:
                    _lib.check(lib.ch_debug_gemm(v, _lib.ptr(X), Mp, _lib.ptr(W), _lib.ptr(bias), M, N, K, ep, _lib.ptr(out), N,
                                                 _lib.ptr(resid), N, _lib.ptr(scale), _lib.ptr(addend) if ep == 4 else None,
                                                 _lib.stream_ptr()), "gemm")
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    times[vt].append(e0.elapsed_time(e1))
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
