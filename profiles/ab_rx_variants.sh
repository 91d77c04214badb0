#!/bin/bash
# A/B of builds of libmi355nrphy.so on ONE box with the receive-chain bench (profiles/rx_chain_bench.py): kernel times per variant.
# Usage (GPU box, repository root): bash profiles/ab_rx_variants.sh ROUNDS build/variants/a.so build/variants/b.so ...
ROUNDS=$1; shift
for i in $(seq $ROUNDS); do
  for v in "$@"; do
    NRPHY_LIB_SO=$PWD/$v python3 profiles/rx_chain_bench.py ${RX_ARGS:-} 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', {k: round(x, 4) for k, x in d['kernel_ms'].items()}, round(d['value']), d.get('verified'))"
  done
done
