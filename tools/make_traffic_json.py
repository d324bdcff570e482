#!/usr/bin/env python3
"""profiles/gemm_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --encode-only`.

HBM-side bytes per launch of every GEMM kernel, ONE ROW PER PROBLEM SHAPE: dispatches are grouped by (kernel name, grid
work-items), not by name alone -- one instantiation serves several shapes of the chain, and a per-name average over mixed shapes
is not a traffic figure of any of them (round 2's file had such rows, one of them below its own algorithmic bytes).
Corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (x2),
WRITE_SIZE is exact for 16-byte-per-lane stores; both are in KiB.

    python tools/make_traffic_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [<bench.json>]

<bench.json>: the JSON line of a `bench.py` run of the same build.  Every `roofline_per_kernel` row of it names a (kernel, grid)
and its algorithmic bytes per launch; a counter figure below that is refused (exit status 1), and the ratio is written next to
the bytes (`vs_algorithmic`: wasted re-reads show up as a ratio well above 1).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short_name(k: str) -> str:
    m = re.search(r"(gemm_\w+<[^>]*>)", k)
    return m.group(1) if m else k


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and "gemm_" in r["Kernel_Name"]:
                acc[(short_name(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    bench = sys.argv[4] if len(sys.argv) > 4 else None
    F, W = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(F) | set(W)):
        f = sum(F.get(k, [0])) / max(1, len(F.get(k, [])))
        w = sum(W.get(k, [0])) / max(1, len(W.get(k, [])))
        rows[f"{k[0]}@{k[1]}"] = {
            "kernel": k[0], "grid": k[1], "launches": max(len(F.get(k, [])), len(W.get(k, []))),
            "fetch_kib_raw": round(f, 1), "write_kib": round(w, 1),
            "read_bytes_per_launch": round(2.0 * f * 1024.0), "write_bytes_per_launch": round(w * 1024.0),
            "hbm_bytes_per_launch": round((2.0 * f + w) * 1024.0)}
    bad = []
    if bench:
        js = [ln for ln in open(bench).read().splitlines() if ln.startswith("{")]
        b = json.loads(js[-1])
        for r in b.get("roofline_per_kernel", []):
            key = f"{r['rocprof_name']}@{r['grid']}"
            if key not in rows:
                bad.append(f"{key}: no dispatch of this (kernel, grid) in the counter files")
                continue
            alg = r["algorithmic_bytes_per_launch"]
            rows[key]["algorithmic_bytes_per_launch"] = alg
            rows[key]["vs_algorithmic"] = round(rows[key]["hbm_bytes_per_launch"] / alg, 3)
            if rows[key]["hbm_bytes_per_launch"] < 0.98 * alg:
                bad.append(f"{key}: counter bytes {rows[key]['hbm_bytes_per_launch']} below the algorithmic bytes {alg}")
    json.dump({"note": "one row per (kernel, grid work-items) of `bench.py --encode-only --streams 1`; bytes = (2*FETCH_SIZE + WRITE_SIZE) * "
                       "1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request; Infinity-Cache hits are counted)",
               "per_kernel": rows}, open(out, "w"), indent=1)
    print(open(out).read())
    if bad:
        sys.exit("make_traffic_json: REFUSED\n  " + "\n  ".join(bad))


if __name__ == "__main__":
    main()
