#!/bin/bash
# Collects the profiles DESIGN.md cites.  Run on the GPU box from the repository root:
#   bash profiles/collect.sh <tag>        e.g.  bash profiles/collect.sh r01_v6
# Writes gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the default bench command),
# gpurun_out/<tag>_pmc_summary.txt (FETCH_SIZE / WRITE_SIZE / SQ counters, separate passes) and
# gpurun_out/<tag>_bench.json.  Copy them into profiles/ afterwards.
set -u
TAG=$1
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
python3 bench.py > "$OUT/${TAG}_bench.log" 2>&1 && tail -1 "$OUT/${TAG}_bench.log" > "$OUT/${TAG}_bench.json"
echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary > "$OUT/${TAG}_stats.log" 2>&1
echo "stats rc=$?"
cp "$OUT/${TAG}_stats"/*/*kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv"
cd "$ROOT"
bash profiles/pmc.sh "$OUT/${TAG}_pmc" > "$OUT/${TAG}_pmc.log" 2>&1
python3 profiles/pmc_summary.py "$OUT/${TAG}_pmc" > "$OUT/${TAG}_pmc_summary.txt"
# profiles/traffic.json of THIS tree (bytes, instructions, clock, source sha): copy to profiles/traffic.json afterwards
NRPHY_SKIP_ISA_MODEL=1 python3 profiles/make_traffic_json.py "$OUT/${TAG}_pmc" "gpurun_out/${TAG}_pmc_summary.txt (profiles/collect.sh ${TAG}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes via profiles/pmc.sh, bench.py --slots 1024 --no-secondary; FETCH_SIZE doubled as MI355X_MICROARCH.md section HBM prescribes for gfx950; units KiB)" > "$OUT/${TAG}_traffic.json"
rm -rf "$OUT/${TAG}_pmc" "$OUT/${TAG}_stats"
echo "pmc done"
