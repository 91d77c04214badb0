#!/bin/bash
# A/B/C... of several builds of libmi355nrphy.so on ONE box (box-to-box variance is +-4 %): alternates bench.py runs.
# Usage (GPU box, repository root): ROUNDS=3 bash profiles/ab_multi.sh build/variants/a.so build/variants/b.so ...
ROUNDS=${ROUNDS:-3}
LIB=srsran-edgeric-5g_amd/csrc/libmi355nrphy.so
cp $LIB /tmp/keep.so
for i in $(seq $ROUNDS); do
  for v in "$@"; do
    cp $v $LIB
    python3 bench.py --no-cpu-baseline --no-secondary ${BENCH_ARGS:-} 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['kernel_ms'], round(d['value']), d.get('verified_vs_oracle'))"
  done
done
cp /tmp/keep.so $LIB
