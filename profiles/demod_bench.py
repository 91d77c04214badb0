#!/usr/bin/env python3
"""Secondary measurement ("next" row): OFDM demodulator throughput at the config-3 numerology
(FFT 4096, 273 PRB, 4 ports), inputs resident in HBM.  Usage (GPU box, repository root):
python3 profiles/demod_bench.py [--slots 1024]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=1024)
    args = ap.parse_args()
    import torch
    import backends
    import cases
    lib = backends.pkg.lib
    ctx = lib.Context(0)
    _, ports, subc, cfg = cases.baseline_config(3)
    plan = lib.OfdmPlan(ctx, cfg, ports)
    slots = args.slots
    d_iq = torch.randn((slots, ports, plan.slot_stride, 2), dtype=torch.float32, device="cuda")
    d_grid = torch.zeros((slots, ports, 14, subc), dtype=torch.int32, device="cuda")
    d_slot = torch.tensor([i % 2 for i in range(slots)], dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()  # an explicit stream: a null handle would select the context's own stream
    torch.cuda.synchronize()
    for _ in range(3):
        plan.demod_run(slots, d_iq, d_grid, d_slot_index=d_slot, stream=s.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record(s)
    for _ in range(n):
        plan.demod_run(slots, d_iq, d_grid, d_slot_index=d_slot, stream=s.cuda_stream)
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    nbytes = slots * ports * (plan.slot_stride * 8 + 14 * subc * 4)
    print("ofdm_demod_kernel<4096>: %.4f ms per %d slots -> %.0f k slots/s, %.1f GB/s (%.1f %% of 8 TB/s)" % (
        ms, slots, slots / ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80.0))


if __name__ == "__main__":
    main()
