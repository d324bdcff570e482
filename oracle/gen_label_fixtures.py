#!/usr/bin/env python3
"""tests/golden/labels_*.npz: integer class-label vectors parsed from the reference's image list files
(`<relative/path.jpg> <label>` per line; data/<dataset>/{database,test}.txt).  Data only -- no images exist in the
snapshot.  Run in the build container:  python oracle/gen_label_fixtures.py"""
import os
import numpy as np

REF = "/root/reference/data"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SETS = {"cub200": ("cub200_2011", "database.txt", "test.txt"),
        "nabirds": ("nabirds", "database.txt", "test.txt"),
        "cars196": ("cars196", "train.txt", "test.txt")}  # cars196/database.txt is among the missing large blobs


def labels(path):
    return np.array([int(line.rsplit(" ", 1)[1]) for line in open(path) if line.strip()], dtype=np.int16)


for name, (folder, db, te) in SETS.items():
    d, t = labels(os.path.join(REF, folder, db)), labels(os.path.join(REF, folder, te))
    np.savez_compressed(os.path.join(OUT, f"labels_{name}.npz"), db=d, test=t)
    print(name, d.shape, t.shape, int(d.max()) + 1)
