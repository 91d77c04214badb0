#!/usr/bin/env python3
"""Through-the-shim timing: the host-span entry points the reference's adaptors call, one PDU /
one slot at a time with host buffers -- plan creation, H2D, kernels, D2H and synchronisation included.
Usage (GPU box, repository root): python3 profiles/shim_latency.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import backends
    import cases
    lib = backends.pkg.lib
    ctx = lib.Context(0)
    rng = np.random.default_rng(0)
    for cfg in (1, 2, 3):
        pdu, ports, subc, ofdm = cases.baseline_config(cfg)
        tb = cases.random_tb(rng, pdu)
        oplan = lib.OfdmPlan(ctx, ofdm, ports)
        grid = ctx.pdsch_process_host(pdu, tb, ports, subc)
        for _ in range(3):
            ctx.pdsch_process_host(pdu, tb, ports, subc, grid=grid)
            oplan.modulate_slot_host(grid, 0)
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.pdsch_process_host(pdu, tb, ports, subc, grid=grid)
        t1 = time.perf_counter()
        for _ in range(n):
            oplan.modulate_slot_host(grid, 0)
        t2 = time.perf_counter()
        ms_p, ms_o = 1e3 * (t1 - t0) / n, 1e3 * (t2 - t1) / n
        print("config %d: pdsch_process_host %.3f ms/PDU, ofdm modulate_slot_host (all ports) %.3f ms/slot -> %.0f slots/s "
              "through the shim, one slot in flight" % (cfg, ms_p, ms_o, 1e3 / (ms_p + ms_o)), flush=True)
        # The asynchronous seam (nrphy_pdsch_async_*: what the asynchronous pdsch_processor drop-in uses): `depth` PDUs in
        # flight, transport block in and whole grid out over PCIe for every PDU, completions counted by the library's
        # counting handler on the runtime's threads.
        import ctypes as C
        h = ctx.lib
        done_fn = C.cast(h.nrphy_pdsch_async_count_done, C.c_void_p)
        for depth in (1, 4, 8):
            q = C.c_void_p()
            assert h.nrphy_pdsch_async_create(ctx.handle, depth, ports, subc, pdu.tb_size_bytes, C.byref(q)) == 0
            count = C.c_uint64(0)
            total = 400

            def pump(n_submit):
                sent = 0
                while sent < n_submit:
                    rc = h.nrphy_pdsch_async_submit(q, C.byref(pdu), tb.ctypes.data, done_fn, C.byref(count))
                    if rc == 0:
                        sent += 1
                    elif rc != 4:   # NRPHY_ERR_CAPACITY: queue full, try again
                        raise RuntimeError(rc)
                h.nrphy_pdsch_async_wait(q)

            pump(20)
            count.value = 0
            t0 = time.perf_counter()
            pump(total)
            dt = time.perf_counter() - t0
            assert count.value == total, (count.value, total)
            print("config %d: asynchronous seam, %d in flight: %.0f PDUs/s (%.3f ms per PDU)" % (cfg, depth, total / dt, 1e3 * dt / total),
                  flush=True)
            h.nrphy_pdsch_async_destroy(q)
        # Several submitting threads on one queue (the reference runs one processor instance per downlink thread,
        # pdsch_processor_pool.h:51-58): ctypes releases the interpreter lock during the call, so the submits run in parallel.
        import threading
        for nthreads in (2, 4):
            q = C.c_void_p()
            assert h.nrphy_pdsch_async_create(ctx.handle, 8, ports, subc, pdu.tb_size_bytes, C.byref(q)) == 0
            count = C.c_uint64(0)
            total = 800

            def worker(n_submit):
                sent = 0
                while sent < n_submit:
                    rc = h.nrphy_pdsch_async_submit(q, C.byref(pdu), tb.ctypes.data, done_fn, C.byref(count))
                    if rc == 0:
                        sent += 1
                    elif rc != 4:
                        raise RuntimeError(rc)

            def pump_threads(n_submit):
                ts = [threading.Thread(target=worker, args=(n_submit // nthreads,)) for _ in range(nthreads)]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
                h.nrphy_pdsch_async_wait(q)

            pump_threads(40)
            count.value = 0
            t0 = time.perf_counter()
            pump_threads(total)
            dt = time.perf_counter() - t0
            assert count.value == total, (count.value, total)
            print("config %d: asynchronous seam, 8 in flight, %d submitting threads: %.0f PDUs/s (%.3f ms per PDU)" % (
                cfg, nthreads, total / dt, 1e3 * dt / total), flush=True)
            h.nrphy_pdsch_async_destroy(q)

    # A whole cell-slot of BASELINE config 4 (four PDUs, 68 PRB each) through the asynchronous seam, 4 operations in flight:
    # PDU by PDU (four operations, four grids back) against nrphy_pdsch_async_submit_slot (one operation, one grid back).
    import ctypes as C
    abi = backends.abi
    h = ctx.lib
    done_fn = C.cast(h.nrphy_pdsch_async_count_done, C.c_void_p)
    pdus, ports, subc = cases.mixed_cell(0)
    tbs = [cases.random_tb(rng, q) for q in pdus]
    arr = (abi.PdschPdu * 4)(*pdus)
    ptrs = (C.c_void_p * 4)(*[t.ctypes.data for t in tbs])
    total_tb = sum(((q.tb_size_bytes + 7) & ~3) for q in pdus)
    for label, per_slot in (("PDU by PDU", 4), ("one operation per slot", 1)):
        q = C.c_void_p()
        assert h.nrphy_pdsch_async_create(ctx.handle, 4, ports, subc, total_tb, C.byref(q)) == 0
        count = C.c_uint64(0)

        def pump(n_slots):
            for _ in range(n_slots):
                if per_slot == 1:
                    while True:
                        rc = h.nrphy_pdsch_async_submit_slot(q, 4, arr, ptrs, done_fn, C.byref(count))
                        if rc == 0:
                            break
                        assert rc == 4, rc
                else:
                    for k in range(4):
                        while True:
                            rc = h.nrphy_pdsch_async_submit(q, C.byref(pdus[k]), tbs[k].ctypes.data, done_fn, C.byref(count))
                            if rc == 0:
                                break
                            assert rc == 4, rc
            h.nrphy_pdsch_async_wait(q)

        pump(10)
        count.value = 0
        n_slots = 200
        t0 = time.perf_counter()
        pump(n_slots)
        dt = time.perf_counter() - t0
        assert count.value == n_slots * per_slot
        print("config 4 cell-slot (4 PDUs), asynchronous seam, 4 in flight, %s: %.0f cell-slots/s (%.3f ms per slot)" % (
            label, n_slots / dt, 1e3 * dt / n_slots), flush=True)
        h.nrphy_pdsch_async_destroy(q)


if __name__ == "__main__":
    main()
