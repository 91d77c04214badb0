#!/usr/bin/env python3
"""Workgroup timeline of the prologue launch (transport-block CRC + scrambling / DM-RS sequences) of 1024 config-3 slots.
Needs a variant build that records it (never the product library):
    bash profiles/make_variant.sh trace "pdsch_kernels.hip" "-DNRPHY_WG_TRACE"                       (build container)
    NRPHY_LIB_SO=build/variants/trace.so python3 profiles/prologue_trace.py [--ofdm] > out.txt       (GPU box)
Thread 0 of every workgroup stores the 100 MHz wall clock at its start, at a mid point (CRC role: after the per-thread
Horner chains) and at its end, plus HW_ID / XCC_ID.  Prints, per role: when the workgroups start and end relative to the
first start, how long they take, how many are resident per compute unit over time.  --ofdm runs an OFDM launch in front of
every PDSCH run, as in the bench step (the prologue then starts behind 2 GB of IQ stores)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import backends
    import cases
    lib = backends.pkg.lib
    h = lib.load()
    ctx = lib.Context(0)
    slots = 1024
    pdu0, ports, subc, ofdm = cases.baseline_config(3)
    tb_stride = (pdu0.tb_size_bytes + 255) & ~255
    pdus = [cases.baseline_config(3, slot_index=i % 20)[0] for i in range(slots)]
    plan = lib.PdschPlan(ctx, pdus, [i * tb_stride for i in range(slots)], list(range(slots)), slots, ports, subc)
    oplan = lib.OfdmPlan(ctx, ofdm, ports)
    d_tb = torch.randint(0, 256, (slots * tb_stride,), dtype=torch.uint8, device="cuda")
    d_grid = torch.zeros((slots, ports, 14, subc), dtype=torch.int32, device="cuda")
    d_iq = torch.zeros((slots, ports, oplan.slot_stride, 2), dtype=torch.float32, device="cuda")
    d_slot = torch.tensor([i % 2 for i in range(slots)], dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with_ofdm = "--ofdm" in sys.argv
    for _ in range(12):
        if with_ofdm:
            oplan.run(slots, d_grid, d_iq, d_slot_index=d_slot)
        plan.run(d_tb, d_grid, zero_grids=True)
    ctx.synchronize()
    torch.cuda.synchronize()
    nblk = 16384
    buf = (C.c_uint64 * (8 * nblk))()
    h.nrphy_debug_wg_trace.argtypes = [C.POINTER(C.c_uint64), C.c_uint32]
    rc = h.nrphy_debug_wg_trace(buf, nblk)
    assert rc == 0, rc
    t = np.frombuffer(buf, dtype=np.uint64).reshape(nblk, 8)
    used = t[:, 0] != 0
    t = t[used]
    role = (t[:, 7] >> np.uint64(60)).astype(int)
    xcc = ((t[:, 7] >> np.uint64(32)) & np.uint64(0xF)).astype(int)
    hw = (t[:, 7] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    cuid = xcc * 64 + se * 16 + cu
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0).astype(np.int64) / 100.0  # microseconds
    mid = (t[:, 1].astype(np.int64) - int(t0)) / 100.0
    end = (t[:, 6] - t0).astype(np.int64) / 100.0
    print("# %d workgroups traced (%s), launch spans %.1f us from the first start to the last end; %d distinct (xcc, se, cu)"
          % (len(t), "behind an OFDM launch" if with_ofdm else "PDSCH runs back to back", end.max(), len(set(cuid.tolist()))))
    m = role == 3
    if m.any():  # the codeblock launch ran last and its workgroups overwrote the prologue's records
        marks = [(t[m, k].astype(np.int64) - t[m, 0].astype(np.int64)) / 100.0 for k in (1, 2, 3, 6)]
        names = ("codeblock built", "encoded", "rate matched", "mapped and stored")
        print("codeblock waves (wave 0 of the first %d workgroups), us after the wave's start:" % m.sum())
        prev = None
        for nm, x in zip(names, marks):
            d = x if prev is None else x - prev
            print("  %-18s at p50 %6.2f p90 %6.2f | stage p50 %6.2f p90 %6.2f us" % (nm, np.percentile(x, 50), np.percentile(x, 90),
                                                                                  np.percentile(d, 50), np.percentile(d, 90)))
            prev = x
        life = (end[m] - start[m])
        print("  wave life p50 %.2f p90 %.2f us; launch spans %.1f us for these workgroups" % (np.percentile(life, 50), np.percentile(life, 90), end[m].max()))
        return
    for r, name in ((1, "sequence"), (2, "tb crc")):
        m = role == r
        if not m.any():
            continue
        d = end[m] - start[m]
        print("%-9s n=%5d  start: min %.1f p50 %.1f p90 %.1f max %.1f us | end: p50 %.1f p90 %.1f max %.1f us | "
              "duration: min %.1f p50 %.1f p90 %.1f max %.1f us" %
              (name, m.sum(), start[m].min(), np.percentile(start[m], 50), np.percentile(start[m], 90), start[m].max(),
               np.percentile(end[m], 50), np.percentile(end[m], 90), end[m].max(),
               d.min(), np.percentile(d, 50), np.percentile(d, 90), d.max()))
        if r == 1:
            marks = [(t[m, k].astype(np.int64) - t[m, 0].astype(np.int64)) / 100.0 for k in (1, 2, 3, 4, 6)]
            print("          sequence wave, us after its start (p50): state jump %.1f | 31 head words %.1f | seed rows (LDS doubling) %.1f | "
                  "seed rows stored %.1f | all rows stored %.1f" % tuple(np.percentile(x, 50) for x in marks))
        if r == 2:
            a = mid[m] - start[m]
            b = end[m] - mid[m]
            print("          tb crc phases: loads + tables + chains p50 %.1f p90 %.1f us | fold + share p50 %.1f p90 %.1f us"
                  % (np.percentile(a, 50), np.percentile(a, 90), np.percentile(b, 50), np.percentile(b, 90)))
            late = m & (start > 0.6 * start[m].max())  # steady state, the sequence workgroups gone
            marks = [(t[late, k].astype(np.int64) - t[late, 0].astype(np.int64)) / 100.0 for k in (3, 2, 1, 6)]
            print("          tb crc workgroup in steady state, us after its start (p50): descriptors %.1f | words + tables there %.1f | "
                  "chains done %.1f | share stored %.1f" % tuple(np.percentile(x, 50) for x in marks))
    # residency over time
    print("# resident workgroups (whole device) every 5 us: sequence / tb crc")
    for x in np.arange(0.0, end.max() + 5.0, 5.0):
        live = (start <= x) & (end > x)
        print("  t=%5.1f us  %5d / %5d" % (x, (live & (role == 1)).sum(), (live & (role == 2)).sum()))
    # duration of tb-crc workgroups by start time (does a workgroup take longer early in the launch?)
    m = role == 2
    if m.any():
        print("# tb crc workgroup duration by start time")
        for lo in np.arange(0.0, start[m].max() + 10.0, 10.0):
            sel = m & (start >= lo) & (start < lo + 10.0)
            if sel.any():
                d = end[sel] - start[sel]
                print("  started %5.1f-%5.1f us: n=%5d  p50 %.1f  p90 %.1f us" % (lo, lo + 10.0, sel.sum(), np.percentile(d, 50), np.percentile(d, 90)))


if __name__ == "__main__":
    main()
