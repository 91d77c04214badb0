#!/bin/bash
# A/B of two builds of libmi355nrphy.so on ONE box (box-to-box variance is +-4 %): alternates bench.py runs.
# Usage (GPU box, repository root): bash profiles/ab_lib.sh build/variants/old.so build/variants/new.so [rounds]
A=$1; B=$2; ROUNDS=${3:-3}
LIB=srsran-edgeric-5g_amd/csrc/libmi355nrphy.so
cp $LIB /tmp/keep.so
for i in $(seq $ROUNDS); do
  for v in $A $B; do
    cp $v $LIB
    python3 bench.py --no-cpu-baseline --no-secondary --steps 20 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['kernel_ms'], round(d['value']))"
  done
done
cp /tmp/keep.so $LIB
