#!/usr/bin/env python3
"""Calibration only: what the vendor BLAS (torch -> hipBLASLt / rocBLAS) reaches on the encoder's GEMM shapes, plain bf16
GEMM with bias, no fused epilogue.  Not used by the product path."""
import torch

M = 51456
for name, N, K in [("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072), ("down", 384, 768), ("up", 768, 384)]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    ts = []
    for r in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = torch.nn.functional.linear(x, w, b)
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    fl = 2.0 * M * N * K
    print(f"{name:5s} N={N:5d} K={K:5d}: med {ts[len(ts)//2]*1e3:7.1f} us {fl/ts[len(ts)//2]/1e9:7.1f} TF  (min {ts[0]*1e3:7.1f} us {fl/ts[0]/1e9:7.1f} TF)")
