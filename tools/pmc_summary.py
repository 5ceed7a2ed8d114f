#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one directory per pass) into per-kernel averages per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if name.startswith("_ZN5repet"):                      # not demangled (anonymous namespace): keep the readable part
            for key in ("gram_f16_big_pipe_kernel", "gram_f16_big_kernel", "gram_f16_kernel", "split_f16_rows_kernel", "split_f16_kernel", "stft_reg_kernel", "istft_ola_reg_kernel"):
                if key in name:
                    name = "repet::" + key
                    break
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(acc):
    if not name.startswith("repet::"):
        continue
    print(name)
    for counter in sorted(acc[name]):
        vals = acc[name][counter]
        print(f"    {counter:32s} n={len(vals):3d} mean={sum(vals) / len(vals):.6g}")
