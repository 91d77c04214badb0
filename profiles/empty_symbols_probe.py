"""Wire-format OFDM launch of 1024 config-3-sized grids by the share of EMPTY symbols (rows of zeros): what skipping their transform is
worth for lightly loaded cells.  python3 profiles/empty_symbols_probe.py   (GPU box; NRPHY_LIB_SO selects a variant build)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import backends
import cases

lib, abi = backends.pkg.lib, backends.pkg.abi
ctx = lib.Context(0)
_, ports, subc, ofdm = cases.baseline_config(3)
slots = 1024
plan = lib.OfdmPlan(ctx, ofdm, ports)
wire = abi.IqWireCfg(abi.AmplitudeCfg(0, 1, -14.0, 1.0, -1.0), 32767.0)
d_iq = torch.zeros((slots, ports, plan.slot_stride, 2), dtype=torch.int16, device="cuda")
d_stats = torch.zeros((slots * ports, 4), dtype=torch.int32, device="cuda")
d_slot = torch.tensor([i % 2 for i in range(slots)], dtype=torch.int32, device="cuda")
gen = torch.Generator(device="cuda")
gen.manual_seed(5)
full = torch.randint(0x3C003C00, 0x3F003F00, (slots, ports, 14, subc), dtype=torch.int32, device="cuda", generator=gen)   # bf16 pairs around 0.01-0.5
for occupied in (14, 12, 7, 3, 1, 0):
    grid = full.clone()
    grid[:, :, occupied:, :] = 0
    torch.cuda.synchronize()
    lib.ORDER_AFTER_TORCH = False
    for _ in range(30):
        plan.run_ci16(slots, grid, wire, d_iq, d_slot_index=d_slot, d_stats=d_stats)
    ctx.synchronize()
    plan.enable_timing(20, stride=1)
    for _ in range(20):
        plan.run_ci16(slots, grid, wire, d_iq, d_slot_index=d_slot, d_stats=d_stats)
    ctx.synchronize()
    ms, _ = plan.kernel_time()
    lib.ORDER_AFTER_TORCH = True
    print("%2d of 14 symbols occupied: %.4f ms per 1024 slots" % (occupied, ms), flush=True)
