#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one directory per pass) into per-kernel averages per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import kernel_name  # noqa: E402

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        acc[kernel_name(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(acc):
    if not name.startswith("repet::"):
        continue
    print(name)
    for counter in sorted(acc[name]):
        vals = acc[name][counter]
        print(f"    {counter:32s} n={len(vals):3d} mean={sum(vals) / len(vals):.6g}")
