#!/bin/bash
# The OFDM launch by parts (profiling variant of the library, NRPHY_OFDM_PROBE: buffer ranges set to zero so that loads / stores do
# not reach memory): 0 = everything, 1 = no IQ stores, 2 = no grid loads, 3 = neither (arithmetic + LDS only).
# Usage (GPU box): bash profiles/ofdm_parts.sh [--wire]     (NRPHY_LIB_SO = the probes variant, default build/variants/probes.so)
export NRPHY_LIB_SO=${NRPHY_LIB_SO:-$PWD/build/variants/probes.so}
for r in 1 2; do
for pr in ${PROBES:-0 1 2 3}; do
  NRPHY_OFDM_PROBE=$pr python3 bench.py --no-cpu-baseline --no-secondary --steps 20 "$@" 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('probe $pr  ofdm', d['kernel_ms']['ofdm'], 'step', d['ms_per_step'])"
done
done
