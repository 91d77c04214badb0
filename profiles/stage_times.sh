#!/bin/bash
# Wall time of the codeblock launch when its codeblock waves stop after stage n (see stage_pmc.sh for the stages; 10 = every
# wave returns at once, 11 = everything but the data-RE stores), at bench.py's default step counts (steady clocks).
# Usage (GPU box, repository root): bash profiles/stage_times.sh
# Needs the profiling variant of the library: bash profiles/make_variant.sh probes "pdsch_kernels.hip ofdm_kernels.hip nrphy_host.cpp" "-DNRPHY_PROBES"
export NRPHY_LIB_SO=${NRPHY_LIB_SO:-$PWD/build/variants/probes.so}
for st in ${STAGES:-10 5 7 1 2 4 11 0}; do
  NRPHY_PROFILE_STAGE=$st python3 bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage $st', d['kernel_ms'], round(d['value']))"
done
