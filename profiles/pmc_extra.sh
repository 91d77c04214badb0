#!/bin/bash
# Second-level PMC passes for the codeblock kernel: instruction cycles, instruction cache, wave-launch (SPI) stalls,
# memory-pipeline back-pressure.  Usage (GPU box, repository root): bash profiles/pmc_extra.sh <out_dir>
set -u
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary $*"
run() {
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$OUT/$1" -- python3 "$GRAFT_REPO_ROOT/bench.py" $ARGS > "$OUT/$1.log" 2>&1
  rc=$?
  echo "$1 rc=$rc"
  # a pass that timed out or was killed ends the script: no further GPU step after one that did not finish
  if [ $rc -ge 124 ]; then exit $rc; fi
}
run cyc "SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LEVEL_WAVES SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA"
run icache "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL"
# (the SPI_RA_* wave-launch counters hang rocprofv3 on this pool: not collected)
# TCP and TCC lists: at most four counters of a block per pass (MI355X_MICROARCH.md, "rocprofv3 PMC slots": TCC has 4; the
# 8-counter lists of round 1 failed with error 38).
run l1a "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum"
run l1b "TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TA_BUSY_avr"
run l2a "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
run l2b "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_WRITEBACK_sum"
run mem "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS"
