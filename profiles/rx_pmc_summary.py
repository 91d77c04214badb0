#!/usr/bin/env python3
"""Condenses profiles/rx_pmc.sh output into the receive-chain entries of profiles/traffic.json (printed as JSON):
vector instructions of ldpc_decode_kernel per codeblock at 4 and at 8 iterations (no early stop), HBM bytes per codeblock (8 iterations).
Usage: python3 profiles/rx_pmc_summary.py <out_dir> [source label]"""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else root


def decoder_means(d):
    acc, waves = defaultdict(list), []
    for path in glob.glob("%s/%s/*/*counter_collection.csv" % (root, d)):
        per = defaultdict(float)
        for row in csv.DictReader(open(path)):
            if "ldpc_decode" in row["Kernel_Name"]:
                per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        for (disp, c), v in per.items():
            acc[c].append(v)
    return {c: sum(v) / len(v) for c, v in acc.items()}


out = {"rx_valu_insts_per_codeblock_at_4_iterations": {}, "rx_valu_insts_per_codeblock_at_8_iterations": {},
       "rx_hbm_bytes_per_codeblock": {}, "rx_source": label}
for leg, n_cb in (("bg1", 64 * 104), ("bg2", 64 * 8)):
    m8, m4 = decoder_means(leg + "_it8"), decoder_means(leg + "_it4")
    if "SQ_INSTS_VALU" not in m8 or "SQ_INSTS_VALU" not in m4:
        continue
    # (the first iteration runs a cheaper routine, so the count is not proportional to the iterations: two points are kept)
    out["rx_valu_insts_per_codeblock_at_4_iterations"][leg] = round(m4["SQ_INSTS_VALU"] / n_cb, 1)
    out["rx_valu_insts_per_codeblock_at_8_iterations"][leg] = round(m8["SQ_INSTS_VALU"] / n_cb, 1)
    f, w = decoder_means(leg + "_FETCH_SIZE"), decoder_means(leg + "_WRITE_SIZE")
    if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
        out["rx_hbm_bytes_per_codeblock"][leg] = round((f["FETCH_SIZE"] * 1024 * 2 + w["WRITE_SIZE"] * 1024) / n_cb, 1)
print(json.dumps(out, indent=1))
