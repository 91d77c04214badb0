#!/usr/bin/env python3
"""Probe: device memory before and after many create / run / destroy cycles of every object kind of the library."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import backends, cases
lib = backends.pkg.lib
o = backends.oracle()


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2 ** 20


rng = np.random.default_rng(0)
pdu, ports, subc, ofdm = cases.baseline_config(2)
tb = cases.random_tb(rng, pdu)
base = None
for cycle in range(6):
    ctx = lib.Context(0)
    for i in range(40):
        plan = lib.PdschPlan(ctx, [pdu] * 4, [0, 16384, 32768, 49152], [0, 1, 2, 3], 4, ports, subc)
        d_tb = torch.zeros(70000, dtype=torch.uint8, device="cuda")
        d_grid = torch.zeros((4, ports, 14, subc), dtype=torch.int32, device="cuda")
        plan.run(d_tb, d_grid)
        op = lib.OfdmPlan(ctx, ofdm, ports)
        d_iq = torch.zeros((4, ports, op.slot_stride, 2), dtype=torch.float32, device="cuda")
        op.run(4, d_grid, d_iq)
        ctx.pdsch_process_host(pdu, tb, ports, subc)
        q = lib.PdschAsyncQueue(ctx, 2, ports, subc, pdu.tb_size_bytes)
        q.submit(pdu, tb, lambda status, grid: None)
        q.wait()
        q.close()
        pool = lib.DlSlotPool(ctx, ofdm, ports, 2, pdu.tb_size_bytes + 64)   # the downlink slot pipeline: open, write, modulate, close
        sid = pool.open()
        assert pool.pdsch(sid, [pdu], [tb]) == 0 and pool.modulate(sid, 0) == 0 and pool.wait(sid) == 0
        pool.close(sid)
        pool.destroy()
        ctx.synchronize()
        plan.close()
        op.close()
        del d_tb, d_grid, d_iq
    ctx.close()
    torch.cuda.empty_cache()
    f = free_mb()
    base = f if base is None else base
    print("cycle %d: free device memory %.1f MiB (change since the first cycle %+.1f MiB)" % (cycle, f, f - base), flush=True)
