"""Probe: does a 1024-slot step get faster when the codeblock and OFDM launches alternate over sub-batches whose grids fit
the memory-side cache (256 MB)?  K plans of 1024 / K slots, each with its own slice of the transport blocks, grids and IQ
buffers, run as  pdsch(0) ofdm(0) pdsch(1) ofdm(1) ...  on one stream.  Every plan runs a prologue of its own here (a fixed
cost of 0.02-0.03 ms each, reported from the HIP events), which a single entry for the whole batch would pay once:
`net` = step - (K prologues) + (the prologue of the one-plan run).   python3 profiles/chunk_probe.py   (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import backends
import cases

lib, abi = backends.pkg.lib, backends.pkg.abi

SLOTS, STEPS, WARM = 1020, 40, 40
ctx = lib.Context(0)
nof_ports, nof_subc, ofdm = cases.baseline_config(3)[1:]
pdus = [cases.baseline_config(3, slot_index=i % 20)[0] for i in range(SLOTS)]
stride = (pdus[0].tb_size_bytes + 255) & ~255
gen = torch.Generator(device="cuda")
gen.manual_seed(7)
tb_sets = [torch.randint(0, 256, (SLOTS * stride,), dtype=torch.uint8, device="cuda", generator=gen) for _ in range(4)]
oplan = lib.OfdmPlan(ctx, ofdm, nof_ports)
d_grid = torch.zeros((SLOTS, nof_ports, 14, nof_subc), dtype=torch.int32, device="cuda")
d_iq = torch.zeros((SLOTS, nof_ports, oplan.slot_stride, 2), dtype=torch.float32, device="cuda")
d_slot = torch.tensor([i % 2 for i in range(SLOTS)], dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
lib.ORDER_AFTER_TORCH = False
base_prologue = None
for K in (1, 2, 3, 4, 5, 6, 1, 3):
    n = SLOTS // K
    plans = [lib.PdschPlan(ctx, pdus[c * n:(c + 1) * n], [i * stride for i in range(n)], list(range(n)), n, nof_ports, nof_subc)
             for c in range(K)]
    no = [0]

    def step():
        tb = tb_sets[no[0] % 4]
        for c in range(K):
            plans[c].run(tb[c * n * stride:], d_grid[c * n:], zero_grids=True)
            oplan.run(n, d_grid[c * n:], d_iq[c * n:], d_slot_index=d_slot[c * n:])
        no[0] += 1

    for _ in range(WARM):
        step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        step()
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / STEPS * 1e3
    # prologue time of one plan, from a few event-timed steps after the measurement
    plans[0].enable_timing(8, stride=1)
    for _ in range(8):
        step()
    ctx.synchronize()
    (ms_crc, ms_cb, _, _), _ = plans[0].kernel_times()
    if K == 1 and base_prologue is None:
        base_prologue = ms_crc
    net = ms - K * ms_crc + base_prologue
    print("K %d  (%4d slots per sub-batch, grids %5.1f MB)  step %.4f ms = %.4f M slots/s | prologue %.4f ms each, codeblock %.4f | "
          "net of the extra prologues %.4f ms = %.4f M slots/s" % (K, n, n * nof_ports * 14 * nof_subc * 4 / 1e6, ms, K * n / ms / 1e3,
                                                              ms_crc, ms_cb, net, K * n / net / 1e3), flush=True)
    for p in plans:
        p.close()
