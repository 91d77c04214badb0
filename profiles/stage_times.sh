#!/bin/bash
# Wall time of the codeblock launch when its codeblock waves stop after stage n (see stage_pmc.sh for the stages).
# Usage (GPU box, repository root): bash profiles/stage_times.sh
for st in 5 6 7 1 2 3 4 0; do
  NRPHY_PROFILE_STAGE=$st python3 bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 2 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage $st', d['kernel_ms'], round(d['value']))"
done
