#!/bin/bash
# Slots per launch: kernel times per slot and whole-path rate as the batch shrinks (does a grid that fits the 256 MB
# memory-side cache make the OFDM launch's grid reads cheaper?).  Run on the GPU box:  bash profiles/slots_sweep.sh [wire]
set -u
EXTRA=${1:+--wire}
for S in 64 128 192 256 384 512 768 1024 2048; do
  STEPS=$(( 51200 / S ))
  python3 bench.py --slots $S --steps $STEPS --warmup $(( STEPS / 2 + 30 )) --no-cpu-baseline --no-secondary $EXTRA 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']; s=$S
print(f'slots {s:5d}  value {d[\"value\"]/1e6:.4f} M/s  us per slot: prologue {k[\"prologue_tbcrc_scrambling_seq\"]/s*1e3:.4f} codeblock {k[\"codeblock_dmrs_zerofill\"]/s*1e3:.4f} ofdm {k[\"ofdm\"]/s*1e3:.4f}  step {d[\"ms_per_step\"]/s*1e3:.4f}')
"
done
