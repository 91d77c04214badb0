#!/bin/bash
# Vector instructions of the LDPC decoder per codeblock and iteration, for the roofline entries of the receive chain
# (profiles/rx_chain_bench.py reads them from profiles/traffic.json): PMC passes of the chain with early stop off at 8 and
# at 4 iterations, for both legs; per iteration = (count at 8 - count at 4) / 4 per codeblock, the rest is the fixed part.
# Usage (GPU box, repository root): bash profiles/rx_pmc.sh <out_dir>   then   python3 profiles/rx_pmc_summary.py <out_dir>
set -u
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for leg in bg1 bg2; do
  for it in 8 4; do
    snr=32; [ $leg = bg2 ] && snr=-4
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
      --output-format csv -d "$OUT/${leg}_it$it" -- python3 "$ROOT/profiles/rx_chain_bench.py" --leg $leg --iterations $it --no-early-stop --allow-failures \
      --snr-db $snr --steps 2 --warmup 1 --slots 64 > "$OUT/${leg}_it$it.log" 2>&1
    echo "$leg it$it rc=$?"
  done
  for c in FETCH_SIZE WRITE_SIZE; do
    snr=32; [ $leg = bg2 ] && snr=-4
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/${leg}_$c" -- python3 "$ROOT/profiles/rx_chain_bench.py" \
      --leg $leg --iterations 8 --no-early-stop --allow-failures --snr-db $snr --steps 2 --warmup 1 --slots 64 > "$OUT/${leg}_$c.log" 2>&1
    echo "$leg $c rc=$?"
  done
done
