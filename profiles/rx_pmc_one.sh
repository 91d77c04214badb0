#!/bin/bash
# usage: rx_pmc_one.sh <lib.so> <outdir>
LIB=$(realpath "$1"); OUT=$(realpath -m "$2"); mkdir -p "$OUT"; ROOT=$PWD
export NRPHY_LIB_SO=$LIB   # the loader's override: the product library in the tree is never overwritten
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$OUT/a" -- python3 "$ROOT/profiles/rx_chain_bench.py" --no-early-stop --steps 2 --warmup 1 --slots 64 > "$OUT/a.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM \
  --output-format csv -d "$OUT/b" -- python3 "$ROOT/profiles/rx_chain_bench.py" --no-early-stop --steps 2 --warmup 1 --slots 64 > "$OUT/b.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(path)):
        if "ldpc_decode" in row["Kernel_Name"]:
            per[(row["Dispatch_Id"], row["Counter_Name"], row["Kernel_Name"].split("(")[0])] += float(row["Counter_Value"])
    for (d, c, k), v in per.items():
        acc[k][c].append(v)
for k in acc:
    print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in sorted(acc[k].items())})
PY
rm -rf "$OUT"/*/*/*.db
