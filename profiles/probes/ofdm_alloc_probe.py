#!/usr/bin/env python3
"""Probe: does the OFDM launch time depend on where its buffers were allocated?  Re-allocates the grid and IQ buffers several times
(with a padding allocation of varying size in between) and times 30 launches per placement."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import backends, cases
lib = backends.pkg.lib
ctx = lib.Context(0)
_, ports, subc, cfg = cases.baseline_config(3)
plan = lib.OfdmPlan(ctx, cfg, ports)
slots = 1024
s = torch.cuda.Stream()
keep = []
for trial in range(12):
    pad = torch.empty(((trial * 37) % 11) * 1024 * 1024 + 256 * (trial % 7), dtype=torch.uint8, device="cuda")
    d_grid = torch.randint(0, 2 ** 31 - 1, (slots, ports, 14, subc), dtype=torch.int32, device="cuda")
    d_iq = torch.zeros((slots, ports, plan.slot_stride, 2), dtype=torch.float32, device="cuda")
    d_slot = torch.tensor([i % 2 for i in range(slots)], dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        plan.run(slots, d_grid, d_iq, d_slot_index=d_slot, stream=s.cuda_stream)
    times = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10):
            plan.run(slots, d_grid, d_iq, d_slot_index=d_slot, stream=s.cuda_stream)
        e1.record(s)
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 10)
    print("trial %2d grid %x iq %x (iq-grid mod 2MiB %7d, mod 64KiB %5d): %s ms" % (
        trial, d_grid.data_ptr(), d_iq.data_ptr(), (d_iq.data_ptr() - d_grid.data_ptr()) % (2 << 20),
        (d_iq.data_ptr() - d_grid.data_ptr()) % 65536, " ".join("%.4f" % t for t in times)), flush=True)
    if trial % 3 == 0:
        keep.append((d_grid, d_iq))   # hold some allocations so that the next ones land elsewhere
    del pad
