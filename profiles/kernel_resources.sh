#!/bin/bash
# Register / spill / occupancy report of one HIP source of the library, from the compiler's own remarks
# (-Rpass-analysis=kernel-resource-usage), one line per kernel.  Runs without a GPU.
#   bash profiles/kernel_resources.sh pdsch_kernels.hip [extra hipcc flags] > profiles/rNN_resources_pdsch.txt
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$1; shift
CONTRACT=-ffp-contract=off
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$ROOT/srsran-edgeric-5g_amd/csrc" $CONTRACT "$@" \
  -x hip -c "$ROOT/srsran-edgeric-5g_amd/csrc/$SRC" -o "$TMP/out.o" -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
        continue
    m = re.search(r":\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", line)
    if m and cur is not None:
        cur[m.group(1)] = int(m.group(2))
print("%-64s %5s %5s %7s %9s %10s %10s %8s" % ("kernel", "SGPR", "VGPR", "scratch", "occupancy", "SGPR spill", "VGPR spill", "LDS"))
for r in rows:
    print("%-64s %5d %5d %7d %9d %10d %10d %8d" % (r["name"][-64:], r.get("TotalSGPRs", -1), r.get("VGPRs", -1), r.get("ScratchSize", -1),
          r.get("Occupancy", -1), r.get("SGPRs Spill", -1), r.get("VGPRs Spill", -1), r.get("LDS Size", -1)))
'
rm -rf "$TMP"
