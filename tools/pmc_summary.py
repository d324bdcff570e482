#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per (kernel, grid work-items) -- one problem shape per row -- the mean of
every counter per dispatch, the mean dispatch duration, and where the counters allow it:
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)      (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
  clock_ghz = GRBM_GUI_ACTIVE / 8 / duration                                      (reads high on dispatches < 0.3 ms, MI355X_MICROARCH.md)
    python tools/pmc_summary.py <counter_collection.csv> [--min-us 5]"""
import csv
import re
import sys
from collections import defaultdict


def short_name(k: str) -> str:
    k = k.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void\s+)?([\w:]+(?:<[^>]*>)?)", k)
    return (m.group(1) if m else k)[:60]


def main():
    path = sys.argv[1]
    min_us = float(sys.argv[sys.argv.index("--min-us") + 1]) if "--min-us" in sys.argv else 5.0
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    with open(path) as f:
        for r in csv.DictReader(f):
            key = (short_name(r["Kernel_Name"]), int(r["Grid_Size"]))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[key][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    rows = []
    for key, cs in acc.items():
        d = sum(dur[key].values()) / max(1, len(dur[key]))
        if d < min_us:
            continue
        rows.append((d * len(dur[key]), key, cs, d))
    for _, key, cs, d in sorted(rows, key=lambda x: -x[0]):
        mean = {c: sum(v) / len(v) for c, v in cs.items()}
        line = f"{key[0]} grid {key[1]}: {len(dur[key])} dispatches, {d:.1f} us"
        if "GRBM_GUI_ACTIVE" in mean:
            cyc = mean["GRBM_GUI_ACTIVE"] / 8.0
            line += f", clock {cyc / (d * 1e3):.2f} GHz"
            if "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
                line += f", mfma_util {mean['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc):.3f}"
            if "SQ_LDS_IDX_ACTIVE" in mean and "SQ_BUSY_CU_CYCLES" in mean:
                line += f", lds_busy {mean['SQ_LDS_IDX_ACTIVE'] / max(1.0, mean['SQ_BUSY_CU_CYCLES']):.3f}"
        print(line)
        for c, v in sorted(mean.items()):
            print(f"   {c:28s} mean={v:16.1f}")


if __name__ == "__main__":
    main()
