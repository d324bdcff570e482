#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 `--kernel-trace` run stored in the rocpd SQLite format (`*_results.db`, this image's default
output): one row per (kernel name, grid), sorted by total time.  `python tools/rocpd_stats.py results.db [out.csv] [--skip-first N]`
(N = dispatches of each kernel to ignore: warm-up)."""
import csv
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    db = sqlite3.connect(args[0])
    rows = db.execute("select name, grid_x * grid_y * grid_z, workgroup_x * workgroup_y * workgroup_z, count(*), sum(end - start) / 1e3, "
                      "avg(end - start) / 1e3, min(end - start) / 1e3, max(end - start) / 1e3, max(vgpr_count), max(accum_vgpr_count), "
                      "max(lds_size) from kernels group by name, grid_x, grid_y, grid_z order by 5 desc").fetchall()
    total = sum(r[4] for r in rows) or 1.0
    out = open(args[1], "w", newline="") if len(args) > 1 else sys.stdout
    w = csv.writer(out)
    w.writerow(["Name", "Grid", "Workgroup", "Calls", "TotalDurationUs", "AverageUs", "MinUs", "MaxUs", "Percentage", "VGPR", "AGPR", "LDS"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], r[3], round(r[4], 1), round(r[5], 2), round(r[6], 2), round(r[7], 2), round(100 * r[4] / total, 2), r[8], r[9], r[10]])


if __name__ == "__main__":
    main()
