#!/usr/bin/env python3
"""Micro-benchmark + fp32 check of the attention kernel.  python tools/attn_bench.py [--batch 256] [--ntok 201] [--heads 12]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from concepthash_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--ntok", type=int, default=201)
ap.add_argument("--heads", type=int, default=12)
ap.add_argument("--rounds", type=int, default=10)
a = ap.parse_args()
lib = _lib.load()
B, N, H = a.batch, a.ntok, a.heads
D = H * 64
qkv = torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16)
out = torch.empty(B * N, D, dtype=torch.bfloat16, device="cuda")
ts = []
for r in range(a.rounds + 2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.ch_debug_attention(_lib.ptr(qkv), B, N, H, _lib.ptr(out), _lib.stream_ptr()), "attn")
    e1.record()
    torch.cuda.synchronize()
    if r >= 2:
        ts.append(e0.elapsed_time(e1))
ts.sort()
fl = 4.0 * B * N * N * D
by = (B * N * 3 * D + B * N * D) * 2
print(f"attention B={B} N={N} H={H}: med {ts[len(ts)//2]*1e3:.1f} us  {fl/ts[len(ts)//2]/1e9:.1f} TF  {by/ts[len(ts)//2]/1e6:.0f} GB/s (min {ts[0]*1e3:.1f} us)")
nb = min(B, 4)
q, k, v = qkv[: nb * N].float().view(nb, N, 3, H, 64).permute(2, 0, 3, 1, 4)
ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(nb * N, D)
print("max abs err vs fp32:", float((out[: nb * N].float() - ref).abs().max()))
