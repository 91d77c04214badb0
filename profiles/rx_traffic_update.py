#!/usr/bin/env python3
"""Folds the receive-chain PMC summaries of profiles/rx_leg_profile.sh into a traffic table:
  python3 profiles/rx_traffic_update.py <traffic.json> <tag>      reads gpurun_out/<tag>_bg1_pmc.txt and <tag>_bg2_pmc.txt (or profiles/)
Sets rx_valu_insts_per_codeblock_at_8_iterations, rx_hbm_bytes_per_codeblock (FETCH_SIZE doubled, KiB units) and rx_source for the
legs' decoder kernels (1024 slots: 104 codeblocks per slot on the BG1 leg, 8 on the BG2 leg)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path, tag = sys.argv[1], sys.argv[2]
t = json.load(open(path))
t["rx_valu_insts_per_codeblock_at_8_iterations"], t["rx_hbm_bytes_per_codeblock"] = {}, {}
t.pop("rx_valu_insts_per_codeblock_at_4_iterations", None)
for leg, kernel, n_cb in (("bg1", "ldpc_decode_msg_bg1_kernel", 1024 * 104), ("bg2", "ldpc_decode_msg_bg2_slot_kernel", 1024 * 8)):
    name = "%s_%s_pmc.txt" % (tag, leg)
    f = next(p for p in (os.path.join(ROOT, "gpurun_out", name), os.path.join(ROOT, "profiles", name)) if os.path.exists(p))
    block = re.search(r"nrphy::%s\n((?:   .*\n)+)" % kernel, open(f).read()).group(1)
    c = {m.group(1): float(m.group(2)) for m in re.finditer(r"(\w+)\s+per dispatch\s+(\d+)", block)}
    t["rx_valu_insts_per_codeblock_at_8_iterations"][leg] = round(c["SQ_INSTS_VALU"] / n_cb, 1)
    t["rx_hbm_bytes_per_codeblock"][leg] = round((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / n_cb, 1)
t["rx_source"] = ("profiles/%s_bg1_pmc.txt / %s_bg2_pmc.txt (profiles/rx_leg_profile.sh %s: rocprofv3 --pmc SQ_* / FETCH_SIZE / WRITE_SIZE in separate "
                  "passes of profiles/rx_chain_bench.py --leg bg1|bg2 --no-early-stop, 8 iterations, 1024 slots; FETCH_SIZE doubled as for the downlink "
                  "kernels; BG1: ldpc_decode_msg_bg1_kernel (messages per edge in LDS), BG2: ldpc_decode_msg_bg2_slot_kernel (messages per edge in "
                  "the scratch slot)" % (tag, tag, tag))
json.dump(t, open(path, "w"), indent=1)
print({k: t[k] for k in ("rx_valu_insts_per_codeblock_at_8_iterations", "rx_hbm_bytes_per_codeblock")})
