#!/bin/bash
# Collects rocprofv3 PMC counters for the benchmark kernels in separate passes (kernel-trace only, no other trace
# domains), as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes.  Usage (on the GPU box, from the
# repository root):  bash profiles/pmc.sh <out_dir> [bench args]
set -u
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary $*"
run() { # name, counters
  rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$OUT/$1" -- python3 "$GRAFT_REPO_ROOT/bench.py" $ARGS > "$OUT/$1.log" 2>&1
  echo "$1 rc=$?"
}
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
run sq2 "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
