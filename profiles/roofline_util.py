#!/usr/bin/env python3
"""Shared by bench.py and profiles/rx_chain_bench.py: profiles/traffic.json (with the check that it was measured on the kernel
sources in the tree) and the vector-issue roofline figures."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Vector-issue roof.  Until round 3: one 64-lane vector instruction per FOUR cycles and SIMD -- a convention, not a measured peak
# (profiles/r01_valu_rate.txt: 2.57 cycles for the plain VOP2 integer instructions, 3.44 for v_fma_f32, 4.2-4.4 for most VOP3,
# 4.76-5.42 for packed FP32).  Now a kernel is priced by its own mix: issue cycles = vector instructions of the PMC profile x the
# average issue cost of the kernel's instruction mix (profiles/valu_issue_model.py: static ISA histogram x the measured costs),
# against what 1,024 SIMDs offer in the launch's time at the 2.4 GHz the probe's cycles are defined on.
NOF_SIMDS = 1024
NOMINAL_GHZ = 2.4

_PROFILE = None


def kernel_source_sha():
    """sha256 over the device sources, as profiles/valu_issue_model.py / make_traffic_json.py record it."""
    import hashlib
    csrc = os.path.join(ROOT, "srsran-edgeric-5g_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h", ".inc")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()


def profile_table():
    """profiles/traffic.json: per kernel, HBM bytes and vector instructions per launch from the PMC passes of the default
    command (profiles/pmc.sh) -- measured in the profile run named there, not in this process -- and the issue-cost model of the
    kernels.  Used only when it was measured on THESE kernel sources (kernel_source_sha256); otherwise the line keeps to the
    HBM figures and says so."""
    global _PROFILE
    if _PROFILE is None:
        try:
            _PROFILE = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            have = _PROFILE.get("kernel_source_sha256")
            if have != kernel_source_sha():
                _PROFILE = {"stale": "profiles/traffic.json was measured on other kernel sources (%s...): PMC-derived figures withheld" % (
                    (have or "no sha recorded")[:12])}
        except Exception:
            _PROFILE = {}
    return _PROFILE


def valu_roof(valu, ms, avg_cost, clock_ghz=None):
    """Vector-issue figures of a launch: `valu` wavefront-instructions in `ms` at `avg_cost` issue cycles each."""
    issue_cycles = valu * avg_cost
    offered = NOF_SIMDS * ms * 1e-3 * NOMINAL_GHZ * 1e9
    out = {"bound": "valu", "achieved": round(issue_cycles / (ms * 1e-3) / 1e9, 1), "peak": round(NOF_SIMDS * NOMINAL_GHZ, 1),
           "unit": "G issue-cycles/s", "frac": round(issue_cycles / offered, 4), "issue_cycles": int(issue_cycles),
           "valu_insts_per_launch": int(valu), "avg_issue_cycles_per_instruction": avg_cost,
           "frac_at_4_cycles_per_instruction": round(valu * 4.0 / offered, 4)}
    if clock_ghz:
        out["clock_ghz"] = clock_ghz
    return out


