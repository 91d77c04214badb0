#!/usr/bin/env python3
"""Adds the wire-format modulator's PMC figures to a traffic table:  python3 profiles/wire_traffic_update.py <traffic.json> <pmc_out_dir>
where <pmc_out_dir> is what `bash profiles/pmc.sh <pmc_out_dir> --wire` wrote (same passes as the default command, bench.py --wire)."""
import json
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
path, pmc = sys.argv[1], sys.argv[2]
env = dict(os.environ, NRPHY_SKIP_ISA_MODEL="1")
wire = json.loads(subprocess.run([sys.executable, os.path.join(here, "make_traffic_json.py"), pmc, "wire"], capture_output=True, text=True,
                                 env=env, check=True).stdout)
t = json.load(open(path))
key = "ofdm_kernel<4096, ci16>"
for field in ("hbm_bytes_per_launch", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "valu_insts_per_launch", "salu_insts_per_launch",
              "clock_ghz", "kernel_names"):
    if key in wire.get(field, {}):
        t.setdefault(field, {})[key] = wire[field][key]
if key not in t.setdefault("valu_bound", []):
    t["valu_bound"].append(key)   # 0.8 of its issue roof, 0.58 of the HBM roof at four workgroups per CU
t["wire_source"] = "profiles/pmc.sh <dir> --wire on the same sources (bench.py --wire --slots 1024: the modulator writes complex int16)"
json.dump(t, open(path, "w"), indent=1)
print({f: t[f].get(key) for f in ("hbm_bytes_per_launch", "valu_insts_per_launch", "clock_ghz")})
