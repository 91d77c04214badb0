#!/bin/bash
# Profiles of the receive-side chain (BASELINE config 5).  Run on the GPU box from the repository root:
#   bash profiles/collect_rx.sh <tag>
# Writes gpurun_out/<tag>_rx_kernel_stats.csv (rocprofv3 --kernel-trace --stats of profiles/rx_chain_bench.py) and
# gpurun_out/<tag>_rx_pmc.txt (SQ instruction counters of the decoder and dematcher, a separate pass).
set -u
TAG=$1
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_rx_stats" -- python3 "$ROOT/profiles/rx_chain_bench.py" > "$OUT/${TAG}_rx_stats.log" 2>&1
echo "stats rc=$?"
cp "$OUT/${TAG}_rx_stats"/*/*kernel_stats.csv "$OUT/${TAG}_rx_kernel_stats.csv"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT/${TAG}_rx_pmc" -- python3 "$ROOT/profiles/rx_chain_bench.py" --steps 2 --warmup 1 > "$OUT/${TAG}_rx_pmc.log" 2>&1
echo "pmc rc=$?"
cd "$ROOT"
python3 - "$OUT/${TAG}_rx_pmc" > "$OUT/${TAG}_rx_pmc.txt" <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for path in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if "nrphy::" not in k:
            continue
        k = k.split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
for k in sorted(acc):
    print(k, "dispatches", len(n[k]))
    for c, v in sorted(acc[k].items()):
        print("   %-20s per dispatch %14.1f" % (c, v / len(n[k])))
PY
echo "done"
