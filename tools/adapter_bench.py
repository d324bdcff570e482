#!/usr/bin/env python3
"""Micro-benchmark + fp32 check of the fused adapter kernel, with timing-only ablations.
    python tools/adapter_bench.py [--rows 51456] [--dim 768] [--b 384]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from concepthash_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=51456)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--b", type=int, default=384)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--dbg", default="0,1,2,4,8,6,14,15")
a = ap.parse_args()
lib = _lib.load()
M, D, b = a.rows, a.dim, a.b
bp = (b + 127) // 128 * 128
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, D, generator=g, device="cuda").to(torch.bfloat16)
H0 = torch.randn(M, D, generator=g, device="cuda")
Wd = torch.randn(b, D, generator=g, device="cuda") * D ** -0.5
bd = torch.randn(b, generator=g, device="cuda") * 0.1
gamma = 1 + 0.1 * torch.randn(D, generator=g, device="cuda")
beta = 0.1 * torch.randn(D, generator=g, device="cuda")
Wu = torch.zeros(D, bp, device="cuda")
Wu[:, :b] = torch.randn(D, b, generator=g, device="cuda") * b ** -0.5
Wu = Wu.to(torch.bfloat16)
bu = torch.randn(D, generator=g, device="cuda") * 0.1
scale = torch.tensor([0.7], device="cuda")
wdf = torch.empty(bp, D, dtype=torch.bfloat16, device="cuda")
wc, wd_ = torch.empty(bp, device="cuda"), torch.empty(bp, device="cuda")


def run(H, dbg):
    _lib.check(lib.ch_debug_adapter(_lib.ptr(A), _lib.ptr(H), M, D, b, _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(gamma), _lib.ptr(beta),
                                    _lib.ptr(Wu), _lib.ptr(bu), _lib.ptr(scale), _lib.ptr(wdf), _lib.ptr(wc), _lib.ptr(wd_), dbg,
                                    _lib.stream_ptr()), "adapter")


H = H0.clone()
run(H, 0)
torch.cuda.synchronize()
n = min(M, 1024)
x = A[:n].float()
ln = torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-5)
ref = H0[:n] + x + 0.7 * (torch.nn.functional.gelu(ln @ Wd.t() + bd) @ Wu[:, :b].float().t() + bu)
print("max abs err vs fp32 (first rows):", float((H[:n] - ref).abs().max()), " rms", float(ref.pow(2).mean().sqrt()))
fl = 4.0 * M * D * b
by = M * D * (2 + 8)
for dbg in [int(v) for v in a.dbg.split(",")]:
    ts = []
    for r in range(a.rounds + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(H, dbg)
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"dbg={dbg:2d}: med {med * 1e3:7.1f} us  {fl / med / 1e9:7.1f} TF  {by / med / 1e6:6.0f} GB/s")
