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


def live_traffic():
    """The asynchronous seam on LIVE traffic: every submit brings a PDU that differs from the one before, as a gNB's does
    (the reference derives per-PDU state on every call, pdsch_processor_concurrent_impl.cpp:55-207, and every slot brings a
    new pdu_t, downlink_processor_single_executor_impl.cpp:98-142).  A pool of 64 config-3-sized PDUs: slot_index cycles
    0-19, RNTI and scrambling identities are drawn, the allocation starts at PRB 0-3 and has 262-270 PRB (64 different
    shapes), the MCS is 256-QAM at a rate drawn from 700-948 / 1024 -- transport blocks of 640-870 kbit, 4 layers on the
    100 MHz grid.  A second pool is ONE shape with everything else drawn (what a cell at full load repeats)."""
    import ctypes as C
    import backends
    import cases
    abi, lib = backends.abi, backends.pkg.lib
    ctx = lib.Context(0)
    h = ctx.lib
    done_fn = C.cast(h.nrphy_pdsch_async_count_done, C.c_void_p)
    rng = np.random.default_rng(7)
    w = cases.codebook("four_layer_four_ports_0_0")
    ports, subc = 4, 273 * 12

    def pool(vary_shape):
        out = []
        for i in range(64):
            n_prb = int(rng.integers(262, 271)) if vary_shape else 270
            start = int(rng.integers(0, 273 - n_prb + 1)) if vary_shape else 0
            rate = float(rng.uniform(700, 948))
            tb_bits = cases.tbs(12, 36, 8, rate, 4, n_prb)
            pdu = abi.make_pdu(slot_index=i % 20, rnti=int(rng.integers(1, 65520)), n_id=int(rng.integers(0, 1024)),
                               scrambling_id=int(rng.integers(0, 65536)), bwp_start_rb=0, bwp_size_rb=273, qm=8,
                               dmrs_symbols=(2, 7, 11), nof_cdm_groups_without_data=2, prb_start=start, prb_count=n_prb,
                               start_symbol=0, nof_symbols=12, base_graph=1, precoding=w, tb_size_bytes=tb_bits // 8)
            out.append((pdu, cases.random_tb(rng, pdu)))
        return out

    for label, vary in (("64 shapes", True), ("one shape", False)):
        items = pool(vary)
        max_tb = max(p.tb_size_bytes for p, _ in items)
        # parity of the live path first: a few PDUs of the pool through the queue against the blocking call
        q = lib.PdschAsyncQueue(ctx, 4, ports, subc, max_tb)
        got = {}
        for k in range(8):
            while not q.submit(items[k][0], items[k][1], (lambda st, g, k=k: got.__setitem__(k, (st, g)))):
                q.wait_slot()
        q.wait()
        for k in range(8):
            want = ctx.pdsch_process_host(items[k][0], items[k][1], ports, subc)
            assert got[k][0] == 0 and np.array_equal(got[k][1], want), "async grid differs from the blocking call (PDU %d)" % k
        q.close()
        for depth, nthreads in ((1, 1), (4, 1), (8, 1), (8, 2), (8, 4)):
            qh = C.c_void_p()
            assert h.nrphy_pdsch_async_create(ctx.handle, depth, ports, subc, max_tb, C.byref(qh)) == 0
            count = C.c_uint64(0)
            refs = [(C.byref(p), tb.ctypes.data) for p, tb in items]

            def worker(first, n_submit):
                k = first
                for _ in range(n_submit):
                    pr, tbp = refs[k % 64]
                    k += 1
                    while True:
                        rc = h.nrphy_pdsch_async_submit(qh, pr, tbp, done_fn, C.byref(count))
                        if rc == 0:
                            break
                        if rc != 4:
                            raise RuntimeError(rc)
                        h.nrphy_pdsch_async_wait_slot(qh)

            def pump(n_submit):
                import threading
                ts = [threading.Thread(target=worker, args=(17 * t, n_submit // nthreads)) for t in range(nthreads)]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
                h.nrphy_pdsch_async_wait(qh)

            pump(128)
            count.value = 0
            total = 1600
            t0 = time.perf_counter()
            pump(total)
            dt = time.perf_counter() - t0
            assert count.value == total, (count.value, total)
            print("live traffic (%s, every submit a different PDU), %d in flight, %d submitting thread(s): %.0f PDUs/s (%.3f ms per PDU)"
                  % (label, depth, nthreads, total / dt, 1e3 * dt / total), flush=True)
            h.nrphy_pdsch_async_destroy(qh)


if __name__ == "__main__":
    if "--live" in sys.argv:
        live_traffic()
    else:
        main()
        live_traffic()
