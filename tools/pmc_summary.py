#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean counter value per dispatch."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
with open(path) as f:
    for r in csv.DictReader(f):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")
