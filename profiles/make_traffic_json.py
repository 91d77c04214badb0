#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of profiles/pmc.sh (default bench command, 1024 config-3 slots per launch):
per kernel the HBM bytes per launch (FETCH_SIZE, WRITE_SIZE: KiB units; FETCH_SIZE doubled, as MI355X_MICROARCH.md section
HBM prescribes for gfx950) and the vector instructions per launch (SQ_INSTS_VALU), keyed as bench.py names its kernels.
Usage: python3 profiles/make_traffic_json.py <pmc_out_dir> <source label> [rx_pmc_summary.txt] > profiles/traffic.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, label = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    per_dispatch, names = defaultdict(float), {}
    spans = {}
    for row in csv.DictReader(open(path)):
        per_dispatch[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        names[row["Dispatch_Id"]] = row["Kernel_Name"]
        if row.get("Start_Timestamp") and row.get("End_Timestamp"):
            spans[(row["Dispatch_Id"], row["Counter_Name"])] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    for (disp, counter), v in per_dispatch.items():
        acc[names[disp].split("(")[0].replace("void ", "")][counter].append(v)
        if counter == "SQ_BUSY_CYCLES" and (disp, counter) in spans:   # the dispatch's duration in the pass that counted its cycles
            acc[names[disp].split("(")[0].replace("void ", "")]["_busy_span_ns"].append(spans[(disp, counter)])


def key_of(kernel):
    if "codeblock_kernel" in kernel:
        return "codeblock_kernel"
    if "prologue_kernel" in kernel:
        return "prologue_kernel"
    if "ofdm_kernel<4096" in kernel:
        return "ofdm_kernel<4096, ci16>" if kernel.rstrip().endswith("true>") else "ofdm_kernel<4096>"   # (the wire-format instance: bench.py --wire)
    return None


out = {"slots": 1024, "source": label, "hbm_bytes_per_launch": {}, "hbm_read_bytes_per_launch": {},
       "hbm_write_bytes_per_launch": {}, "valu_insts_per_launch": {}, "salu_insts_per_launch": {},
       # kernels whose launch time follows their vector instruction count, not their bytes
       "valu_bound": ["codeblock_kernel"], "kernel_names": {}}
for kernel, counters in acc.items():
    k = key_of(kernel)
    if k is None:
        continue
    mean = {c: sum(v) / len(v) for c, v in counters.items()}
    out["kernel_names"][k] = kernel
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        rd, wr = int(mean["FETCH_SIZE"] * 1024 * 2), int(mean["WRITE_SIZE"] * 1024)
        out["hbm_read_bytes_per_launch"][k], out["hbm_write_bytes_per_launch"][k] = rd, wr
        out["hbm_bytes_per_launch"][k] = rd + wr
    if "SQ_INSTS_VALU" in mean:
        out["valu_insts_per_launch"][k] = int(mean["SQ_INSTS_VALU"])
    if "SQ_INSTS_SALU" in mean:
        out["salu_insts_per_launch"][k] = int(mean["SQ_INSTS_SALU"])
    if "SQ_BUSY_CYCLES" in mean and mean.get("_busy_span_ns"):
        # engine clock under the kernel: SQ_BUSY_CYCLES counts every shader engine (32 on MI355X) while a wave of the dispatch is resident
        out.setdefault("clock_ghz", {})[k] = round(mean["SQ_BUSY_CYCLES"] / 32.0 / mean["_busy_span_ns"], 3)
# the receive chain's entries: profiles/rx_pmc_summary.py output (third argument), else kept from the previous file
try:
    rx = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else json.load(
        open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json")))
    for k, v in rx.items():
        if k.startswith("rx_"):
            out[k] = v
except Exception:
    pass
# what the counters were measured on, and the issue-cost model of the same sources (bench.py refuses the file for other kernels)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import valu_issue_model
out["kernel_source_sha256"] = valu_issue_model.source_sha()
for k, v in out.items():
    if k.startswith(("valu_issue_model",)):
        pass
try:
    if os.environ.get("NRPHY_SKIP_ISA_MODEL") == "1":   # on the GPU box: the model needs no GPU, add it afterwards (valu_issue_model.py --update)
        raise RuntimeError("skipped: run python3 profiles/valu_issue_model.py --update")
    m = valu_issue_model.model()
    out["valu_issue_model"], out["valu_issue_model_source"] = m["valu_issue_model"], m["valu_issue_model_source"]
except Exception as e:
    out["valu_issue_model_error"] = str(e)
print(json.dumps(out, indent=1))
