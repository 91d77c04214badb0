#!/usr/bin/env python3
"""Per-stage counters of the codeblock kernel from profiles/stage_pmc.sh output.  Usage: stage_summary.py <dir>"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
for st in [5, 6, 7, 1, 2, 3, 4, 0]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for path in glob.glob("%s/stage%d/*/*counter_collection.csv" % (root, st)):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"]
            if "codeblock_kernel" in k:
                k = "codeblock"
            elif "prologue" in k:
                k = "prologue"
            elif "ofdm_kernel" in k:
                k = "ofdm"
            else:
                continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[k].add(row["Dispatch_Id"])
    for k in sorted(acc):
        d = len(n[k])
        print("stage", st, k, {c: round(v / d / 1e6, 2) for c, v in sorted(acc[k].items())})
