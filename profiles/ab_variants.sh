#!/bin/bash
# A/B/C... of several builds of libmi355nrphy.so on ONE box (box-to-box variance is +-4 %): alternates bench.py runs.
# The variant is chosen through NRPHY_LIB_SO (the Python loader's override): the product library in the tree is never touched.
# Usage (GPU box, repository root): bash profiles/ab_variants.sh ROUNDS build/variants/a.so build/variants/b.so ...
ROUNDS=$1; shift
for i in $(seq $ROUNDS); do
  for v in "$@"; do
    NRPHY_LIB_SO=$PWD/$v python3 bench.py --no-cpu-baseline --no-secondary --steps 20 ${BENCH_ARGS:-} 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['kernel_ms'], round(d['value']), d.get('verified_vs_oracle'))"
  done
done
