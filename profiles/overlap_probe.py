#!/usr/bin/env python3
"""One bounded experiment: can the vector-bound codeblock launch and the memory-bound OFDM launch
share the device?  Sub-batch software pipelining on two streams restricted to disjoint sets of compute units
(hipExtStreamCreateWithCUMask): the PDSCH launches (prologue + codeblock kernel) of sub-batch k + 1 on the one set while
the OFDM launch of sub-batch k runs on the other; sub-batches of 256 config-3 slots, four grid buffers in rotation, events
both ways.  Baselines on the same box: one stream with 1024-slot batches (bench.py's form) and one stream with the same
256-slot sub-batches.  Prints a table: ms per 1024 slots for every split.
Usage (GPU box, repository root): python3 profiles/overlap_probe.py > profiles/r03_overlap.txt"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import backends
    import cases
    lib = backends.pkg.lib
    ctx = lib.Context(0)
    hip = C.CDLL("libamdhip64.so")
    nof_cu = torch.cuda.get_device_properties(0).multi_processor_count
    sub, nsub, steps, warm = 256, 4, 40, 20
    pdu0, ports, subc, ofdm = cases.baseline_config(3)
    oplan = lib.OfdmPlan(ctx, ofdm, ports)
    tb_stride = (pdu0.tb_size_bytes + 255) & ~255

    def make_plan(slots):
        pdus = [cases.baseline_config(3, slot_index=i % 20)[0] for i in range(slots)]
        return lib.PdschPlan(ctx, pdus, [i * tb_stride for i in range(slots)], list(range(slots)), slots, ports, subc)

    d_tb = torch.randint(0, 256, (4, 1024 * tb_stride), dtype=torch.uint8, device="cuda")
    d_grid = torch.zeros((1024, ports, 14, subc), dtype=torch.int32, device="cuda")
    d_iq = torch.zeros((1024, ports, oplan.slot_stride, 2), dtype=torch.float32, device="cuda")
    d_slot = torch.tensor([i % 2 for i in range(1024)], dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    lib.ORDER_AFTER_TORCH = False

    def masked_stream(cus):
        words = (nof_cu + 31) // 32
        mask = (C.c_uint32 * words)()
        for cu in cus:
            mask[cu // 32] |= 1 << (cu % 32)
        s = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, mask)
        assert rc == 0, rc
        return s

    def event():
        e = C.c_void_p()
        assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0  # hipEventDisableTiming
        return e

    def timed(run_step):
        for _ in range(warm):
            run_step()
        torch.cuda.synchronize()
        hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_step()
        hip.hipDeviceSynchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    rows = []
    # baseline 1: bench.py's form, one stream, 1024 slots per launch
    plan1024 = make_plan(1024)
    n = [0]

    def step_1024():
        plan1024.run(d_tb[n[0] % 4], d_grid, zero_grids=True)
        oplan.run(1024, d_grid, d_iq, d_slot_index=d_slot)
        n[0] += 1
    rows.append(("one stream, 1024-slot batches (bench.py)", timed(step_1024)))
    plan1024.close()
    # baseline 2: one stream, four 256-slot sub-batches
    plans = [make_plan(sub) for _ in range(2)]

    def views(k):
        return (d_tb[n[0] % 4][k * sub * tb_stride:(k + 1) * sub * tb_stride], d_grid[k * sub:(k + 1) * sub],
                d_iq[k * sub:(k + 1) * sub], d_slot[k * sub:(k + 1) * sub])

    def step_sub():
        for k in range(nsub):
            tb, g, iq, sl = views(k)
            plans[0].run(tb, g, zero_grids=True)
            oplan.run(sub, g, iq, d_slot_index=sl)
        n[0] += 1
    rows.append(("one stream, 256-slot sub-batches", timed(step_sub)))

    # the experiment: PDSCH on stream A (its compute units), OFDM on stream B (the others)
    def pipelined(cus_a, cus_b, label):
        sa, sb = masked_stream(cus_a), masked_stream(cus_b)
        done_a = [event() for _ in range(nsub)]
        done_b = [event() for _ in range(nsub)]
        first = [True]

        def step():
            for k in range(nsub):
                tb, g, iq, sl = views(k)
                if not first[0]:
                    hip.hipStreamWaitEvent(sa, done_b[k], 0)      # grid buffer k is free again
                plans[k % 2].run(tb, g, zero_grids=True, stream=sa)  # two plans: their sequence scratch alternates
                hip.hipEventRecord(done_a[k], sa)
                hip.hipStreamWaitEvent(sb, done_a[k], 0)
                oplan.run(sub, g, iq, d_slot_index=sl, stream=sb)
                hip.hipEventRecord(done_b[k], sb)
            first[0] = False
            n[0] += 1
        ms = timed(step)
        hip.hipStreamSynchronize(sa)
        hip.hipStreamSynchronize(sb)
        hip.hipStreamDestroy(sa)
        hip.hipStreamDestroy(sb)
        rows.append((label, ms))

    all_cus = list(range(nof_cu))
    pipelined(all_cus, all_cus, "two streams, no masks (both on all %d CUs)" % nof_cu)
    for n_ofdm in (64, 96, 128, 160):
        # contiguous mask bits
        pipelined(all_cus[n_ofdm:], all_cus[:n_ofdm], "CU masks, contiguous bits: PDSCH %d CUs | OFDM %d CUs" % (nof_cu - n_ofdm, n_ofdm))
        # ... and strided ones: how mask bits number the compute units of the eight XCDs is not documented for this part; one
        # of the two patterns gives every XCD the same split, the other hands whole XCDs to one side
        per = n_ofdm // 8
        b2 = [cu for cu in all_cus if (cu % 32) < per]
        a2 = [cu for cu in all_cus if (cu % 32) >= per]
        pipelined(a2, b2, "CU masks, strided bits (cu %% 32 < %d): PDSCH %d CUs | OFDM %d CUs" % (per, len(a2), len(b2)))
    lib.ORDER_AFTER_TORCH = True
    base = rows[0][1]
    print("# %s, %d compute units; config 3, ms per 1024 slots (prologue + codeblock + OFDM), %d timed steps after %d" % (
        torch.cuda.get_device_name(0), nof_cu, steps, warm))
    for label, ms in rows:
        print("%-78s %7.3f ms  %6.0f k slots/s  %+5.1f %%" % (label, ms, 1024 / ms, 100 * (base / ms - 1)))


if __name__ == "__main__":
    main()
