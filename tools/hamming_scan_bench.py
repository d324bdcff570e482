#!/usr/bin/env python3
"""The 1M x 128-bit top-10 scan alone (for rocprofv3 --pmc runs and timing).  python tools/hamming_scan_bench.py [--queries 16384]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from concepthash_amd import retrieval as rt

ap = argparse.ArgumentParser()
ap.add_argument("--queries", type=int, default=16384)
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
g = torch.randint(-2 ** 63, 2 ** 63 - 1, (a.rows, 2), dtype=torch.int64, device="cuda")
q = torch.randint(-2 ** 63, 2 ** 63 - 1, (a.queries, 2), dtype=torch.int64, device="cuda")
rt.hamming_topk(q, g, 10)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    rt.hamming_topk(q, g, 10)
torch.cuda.synchronize()
s = (time.perf_counter() - t0) / a.reps
print(f"{a.queries} x {a.rows} x 128 bit top-10: {s * 1e3:.3f} ms  {a.queries * a.rows / s:.4g} cmp/s")
