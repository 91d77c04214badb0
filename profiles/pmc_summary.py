#!/usr/bin/env python3
"""Summarises rocprofv3 counter_collection.csv files: per kernel, the mean counter value per dispatch.
Usage: python profiles/pmc_summary.py <pmc_out_dir>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    per_dispatch = defaultdict(float)
    names = {}
    for row in csv.DictReader(open(path)):
        key = (row["Dispatch_Id"], row["Counter_Name"])
        per_dispatch[key] += float(row["Counter_Value"])
        names[row["Dispatch_Id"]] = row["Kernel_Name"]
    for (disp, counter), v in per_dispatch.items():
        k = names[disp].split("(")[0].replace("void ", "")
        acc[k][counter].append(v)
for k in sorted(acc):
    if "nrphy" not in k and "fillBuffer" not in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-24s mean %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
