#!/bin/bash
# A/B of an environment knob on ONE box: alternates runs of bench.py with and without it.
# Usage (GPU box, repository root): bash profiles/ab_env.sh NRPHY_NO_DIAG_PRECODING [rounds]
KNOB=$1; ROUNDS=${2:-3}
for i in $(seq $ROUNDS); do
  for on in 0 1; do
    if [ $on = 1 ]; then export $KNOB=1; else unset $KNOB; fi
    python3 bench.py --no-cpu-baseline --no-secondary --steps 20 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$KNOB=$on', d['kernel_ms'], round(d['value']))"
  done
done
