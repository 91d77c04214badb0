#!/bin/bash
# Kernel times and vector-instruction counts of ONE leg of the receive chain (BASELINE config 5), 8 fixed iterations.
# GPU box, repository root:  bash profiles/rx_leg_profile.sh <tag> <bg1|bg2>
# -> gpurun_out/<tag>_<leg>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and gpurun_out/<tag>_<leg>_pmc.txt (SQ counters, own pass)
set -u
TAG=$1; LEG=$2
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_${LEG}_stats" -- python3 "$ROOT/profiles/rx_chain_bench.py" --leg $LEG --no-early-stop --steps 4 --warmup 2 > "$OUT/${TAG}_${LEG}_stats.log" 2>&1
echo "stats rc=$?"
cp "$OUT/${TAG}_${LEG}_stats"/*/*kernel_stats.csv "$OUT/${TAG}_${LEG}_kernel_stats.csv"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/${TAG}_${LEG}_pmc" -- python3 "$ROOT/profiles/rx_chain_bench.py" --leg $LEG --no-early-stop --steps 2 --warmup 1 > "$OUT/${TAG}_${LEG}_pmc.log" 2>&1
echo "pmc rc=$?"
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, counter budget of the TCC block)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/${TAG}_${LEG}_pmc_$C" -- python3 "$ROOT/profiles/rx_chain_bench.py" --leg $LEG --no-early-stop --steps 2 --warmup 1 > "$OUT/${TAG}_${LEG}_pmc_$C.log" 2>&1
  echo "pmc $C rc=$?"
done
cd "$ROOT"
python3 - "$OUT/${TAG}_${LEG}_pmc" "$OUT/${TAG}_${LEG}_pmc_FETCH_SIZE" "$OUT/${TAG}_${LEG}_pmc_WRITE_SIZE" > "$OUT/${TAG}_${LEG}_pmc.txt" <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for d in sys.argv[1:]:
    for path in glob.glob(d + "/*/*counter_collection.csv"):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"]
            if "nrphy::" not in k:
                continue
            k = k.split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-22s per dispatch %16.0f  (%d dispatches)" % (c, v / len(n[(k, c)]), len(n[(k, c)])))
PY
rm -rf "$OUT/${TAG}_${LEG}_stats" "$OUT/${TAG}_${LEG}_pmc" "$OUT/${TAG}_${LEG}_pmc_FETCH_SIZE" "$OUT/${TAG}_${LEG}_pmc_WRITE_SIZE"
head -12 "$OUT/${TAG}_${LEG}_kernel_stats.csv"
grep -A12 "ldpc_decode" "$OUT/${TAG}_${LEG}_pmc.txt"
