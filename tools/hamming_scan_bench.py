#!/usr/bin/env python3
"""The Hamming scans alone (for rocprofv3 --kernel-trace / --pmc runs and timing):
    python tools/hamming_scan_bench.py [--mode topk|map|both] [--queries 16384] [--rows 1000000] [--nbit 128] [--classes 200]
mode topk: the exact top-10 scan; mode map: mAP@all + P@{1,5,10} (histogram pass, prefix, one multi-limit AP pass).
Per-kernel times are measured with HIP events around each launch (torch's current stream is the launch stream)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from concepthash_amd import retrieval as rt

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="topk")
ap.add_argument("--queries", type=int, default=16384)
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--nbit", type=int, default=128)
ap.add_argument("--classes", type=int, default=200)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
W = a.nbit // 64
gen = torch.Generator(device="cuda").manual_seed(1234)
g = torch.randint(-2 ** 63, 2 ** 63 - 1, (a.rows, W), dtype=torch.int64, device="cuda", generator=gen)
q = torch.randint(-2 ** 63, 2 ** 63 - 1, (a.queries, W), dtype=torch.int64, device="cuda", generator=gen)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps, out


pairs = a.queries * a.rows
if a.mode in ("topk", "both"):
    s, _ = timed(lambda: rt.hamming_topk(q, g, 10), a.reps)
    print(f"top-10  {a.queries} x {a.rows} x {a.nbit} bit: {s * 1e3:.3f} ms  {pairs / s:.4g} cmp/s")
if a.mode in ("map", "both"):
    gl = torch.randint(0, a.classes, (a.rows,), dtype=torch.int32, device="cuda", generator=gen)
    ql = torch.randint(0, a.classes, (a.queries,), dtype=torch.int32, device="cuda", generator=gen)
    seg = rt.map_seg_rows(a.queries, a.rows, W)
    s_h, hist = timed(lambda: rt.hamming_hist(q, g, ql, gl, 0, seg), a.reps)
    s_p, (base, totals) = timed(lambda: rt.hist_prefix(hist), a.reps)
    limits, _ = rt.normalize_limits([-1, 1, 5, 10])
    s_a1, _ = timed(lambda: rt.hamming_ap_multi(q, g, ql, gl, 0, seg, base, [0]), a.reps)
    s_a4, _ = timed(lambda: rt.hamming_ap_multi(q, g, ql, gl, 0, seg, base, limits), a.reps)
    s_hr, (hist2, recs) = timed(lambda: rt.hamming_hist_rec(q, g, ql, gl, 0, seg), a.reps)
    s_ar, _ = timed(lambda: rt.hamming_ap_rec(q, g, ql, gl, 0, seg, base, recs, limits), a.reps)
    s_e2, ev2 = timed(lambda: rt.evaluate(q, g, ql, gl, R=-1, ks=(1, 5, 10), records=False), max(1, a.reps // 2))
    s_e, ev = timed(lambda: rt.evaluate(q, g, ql, gl, R=-1, ks=(1, 5, 10)), max(1, a.reps // 2))
    print(f"mAP@all {a.queries} x {a.rows} x {a.nbit} bit, {a.classes} classes, seg_rows {seg} ({hist.shape[0]} segments):")
    print(f"  hist pass     {s_h * 1e3:9.3f} ms  {pairs / s_h:.4g} cmp/s")
    print(f"  hist_prefix   {s_p * 1e3:9.3f} ms  ({hist.numel() * 4 * 2 / s_p / 1e9:.0f} GB/s over read + write)")
    print(f"  AP pass, 1 limit  {s_a1 * 1e3:9.3f} ms  {pairs / s_a1:.4g} cmp/s")
    print(f"  AP pass, 4 limits {s_a4 * 1e3:9.3f} ms  {pairs / s_a4:.4g} cmp/s")
    print(f"  one-scan form: hist + records {s_hr * 1e3:9.3f} ms  {pairs / s_hr:.4g} cmp/s (lists of {recs[1]} entries, "
          f"{int(recs[3].sum())} of {recs[3].numel()} workgroups overflowed); AP from records, 4 limits {s_ar * 1e3:9.3f} ms")
    print(f"  evaluate() two-scan form                     {s_e2 * 1e3:9.3f} ms  mAP {ev2['mAP']:.6f}")
    print(f"  evaluate() end to end (mAP@all + P/R@1,5,10) {s_e * 1e3:9.3f} ms  {a.queries / s_e:.4g} queries/s  mAP {ev['mAP']:.6f}")
