#!/bin/bash
# Dynamic instruction counts of the codeblock kernel stage by stage: NRPHY_PROFILE_STAGE=n makes the codeblock waves
# return after stage n (5 work item only, 6 +graph staging, 7 +segmentation, 1 +CRC, 2 +LDPC, 3 +weights,
# 4 +rate matching/interleaving, 0 everything; 11 everything but the data-RE stores).  STAGES="5 6 7 1" selects a subset.
# Usage (GPU box, repository root): bash profiles/stage_pmc.sh <out_dir>
# Needs the profiling variant of the library (the product library has no stage stops):
#   bash profiles/make_variant.sh probes "pdsch_kernels.hip ofdm_kernels.hip nrphy_host.cpp" "-DNRPHY_PROBES"
set -u
export NRPHY_LIB_SO=${NRPHY_LIB_SO:-$PWD/build/variants/probes.so}
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for st in ${STAGES:-1 2 3 4 0}; do
  export NRPHY_PROFILE_STAGE=$st
  rocprofv3 --kernel-trace --pmc ${COUNTERS:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY} \
    --output-format csv -d "$OUT/stage$st" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --slots 256 > "$OUT/stage$st.log" 2>&1
  echo "stage $st rc=$?"
done
