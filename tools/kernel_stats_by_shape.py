#!/usr/bin/env python3
"""Per-(kernel, grid) launch statistics from a `rocprofv3 --kernel-trace` CSV -- what `--stats` prints, split by problem shape.

`rocprofv3 --stats` groups by kernel NAME; one GEMM instantiation serves several shapes of the chain (the 128x128 kernel runs the
adapter projections on all token rows and, in the final layer, on the compact head rows), so its averages mix shapes.  This tool
groups the same dispatch rows by (short kernel name, grid work-items, workgroup size): one row = one shape.

    python tools/kernel_stats_by_shape.py <*_kernel_trace.csv> [--bench <bench.json>] [--skip-before <kernel substring>] > by_shape.csv

--bench: the JSON line the traced `bench.py` run printed.  Appends, for every `roofline_per_kernel` row of it, the agreement check
the profile has to pass: CSV average duration x launches per step against the bench's own `ms_per_step` (HIP events), in percent
(rows within EVENT_US microseconds per launch of each other pass whatever their ratio: see EVENT_US).
Launches of the model build / warm-up that use other grids simply show up as rows of their own.
"""
import argparse
import csv
import json
import re
import sys
from collections import defaultdict


# The bench brackets every GEMM / attention dispatch with the start / stop events of hipExtLaunchKernelGGL.  Their timestamps sit
# on the command processor's side of the dispatch and read 0.3-3.2 us longer than rocprofv3's begin -> end of the kernel itself
# (measured: +3.0 us on 53-111 us launches, +0.3-1.4 us on 200-270 us ones), so two figures agree when they are within 3 % OR
# within this many microseconds per launch.
EVENT_US = 3.5


def short_name(k: str) -> str:
    k = k.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void\s+)?([\w:]+(?:<[^>]*>)?)", k)
    return m.group(1) if m else k


def load(path):
    rows = defaultdict(list)
    meta = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Kind", "KERNEL_DISPATCH") != "KERNEL_DISPATCH":
                continue
            name = short_name(r["Kernel_Name"])
            grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
            wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
            key = (name, grid, wg)
            rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)   # us
            meta[key] = (r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("LDS_Block_Size", ""))
    return rows, meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--bench")
    a = ap.parse_args()
    rows, meta = load(a.trace)
    total = sum(sum(v) for v in rows.values())
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Grid", "Workgroup", "Calls", "TotalDurationUs", "AverageUs", "MinUs", "MaxUs", "Percentage", "VGPR", "AGPR", "LDS"])
    for key, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        name, grid, wg = key
        w.writerow([name, grid, wg, len(v), round(sum(v), 1), round(sum(v) / len(v), 2), round(min(v), 2), round(max(v), 2),
                    round(100.0 * sum(v) / total, 2), *meta[key]])
    if a.bench:
        js = [ln for ln in open(a.bench).read().splitlines() if ln.startswith("{")]
        b = json.loads(js[-1])
        print()
        print("# agreement with the bench line of the same run: rocprofv3 average x launches per step vs kernel_ms_per_step (HIP events)")
        w.writerow(["Name", "Grid", "LaunchesPerStep", "RocprofAverageUs", "BenchAverageUs", "RocprofMsPerStep", "BenchMsPerStep", "DeltaPercent"])
        worst = 0.0
        for r in b.get("roofline_per_kernel", []):
            key = next((k for k in rows if k[0] == r["rocprof_name"] and k[1] == r["grid"]), None)
            if key is None:
                w.writerow([r["rocprof_name"], r["grid"], r["launches_per_step"], "MISSING", r["avg_launch_us"], "", r["ms_per_step"], ""])
                worst = float("inf")
                continue
            avg = sum(rows[key]) / len(rows[key])
            ms = avg * r["launches_per_step"] * 1e-3
            d = 100.0 * (ms - r["ms_per_step"]) / r["ms_per_step"]
            if abs(avg - r["avg_launch_us"]) > EVENT_US:       # beyond what the start / stop events themselves add to a launch
                worst = max(worst, abs(d))
            w.writerow([r["rocprof_name"], r["grid"], r["launches_per_step"], round(avg, 2), r["avg_launch_us"], round(ms, 3),
                        r["ms_per_step"], round(d, 2)])
        print(f"# worst |delta| {worst:.2f} % of the rows that exceed {EVENT_US} us per launch in absolute terms (bound: 3 %)")
        if worst > 3.0:
            sys.exit(f"kernel_stats_by_shape: rocprofv3 and the bench's HIP events disagree by {worst:.2f} % (> 3 % and > {EVENT_US} us per launch)")


if __name__ == "__main__":
    main()
