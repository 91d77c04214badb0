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


def live_pdu_pool(rng, vary_shape=True, n=64):
    """64 config-3-sized PDUs that all differ (see live_traffic)."""
    import backends
    import cases
    abi = backends.abi
    w = cases.codebook("four_layer_four_ports_0_0")
    out = []
    for i in range(n):
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


def dl_slot_pipeline(ctx=None, depths=(1, 2, 4, 8), total=600, verbose=True, check=True):
    """Seams A + C through the downlink slot pipeline (nrphy_dl_slots_*), config-3-sized live traffic (every slot another
    PDU): per slot the transport block goes down (one copy with the plan's tables), the grid stays in HBM, the whole slot is
    modulated when the grid is handed over and the IQ comes up into pinned memory -- float32 as pdxch_processor_baseband hands
    it to the radio buffers, or int16 after the amplitude controller.  `k in flight` = slots opened before the oldest is
    waited for.  Returns a list of {"leg", "in_flight", "slots_per_sec", "ms_per_slot", "pcie_bytes_down", "pcie_bytes_up"}."""
    import ctypes as C
    import backends
    import cases
    abi, lib = backends.abi, backends.pkg.lib
    ctx = ctx or lib.Context(0)
    h = ctx.lib
    rng = np.random.default_rng(11)
    items = live_pdu_pool(rng)
    _, ports, subc, ofdm = cases.baseline_config(3)
    max_tb = max(p.tb_size_bytes for p, _ in items)
    refs = [(C.byref(p), (C.c_void_p * 1)(tb.ctypes.data)) for p, tb in items]
    wire = abi.IqWireCfg(abi.AmplitudeCfg(0, 1, -(10.0 * float(np.log10(subc)) + 12.0), 1.0, -0.1), 32767.0)   # the reference's radio unit setting (bench.py)
    results = []
    for leg, wire_cfg, sample_bytes in (("tb_in_f32_iq_out", None, 8), ("tb_in_ci16_iq_out", wire, 4)):
        if check:
            # parity of the leg first: one slot against the blocking host-span calls of the same library
            pool = lib.DlSlotPool(ctx, ofdm, ports, 1, max_tb, wire_cfg=wire_cfg)
            sid = pool.open()
            assert pool.pdsch(sid, [items[0][0]], [items[0][1]]) == 0 and pool.modulate(sid, 1) == 0 and pool.wait(sid) == 0
            grid = ctx.pdsch_process_host(items[0][0], items[0][1], ports, subc)
            assert np.array_equal(pool.read_grid(sid), grid)
            if wire_cfg is None:
                oplan = lib.OfdmPlan(ctx, ofdm, ports)
                want = oplan.modulate_slot_host(grid, 1)
                got = np.stack([pool.iq(sid, p) for p in range(ports)])
                assert np.array_equal(got, want.reshape(ports, -1)), "pipeline IQ differs from the blocking call"
                oplan.close()
            pool.close(sid)
            pool.destroy()
        slot_samples = lib.slot_size(ofdm, 0)
        for depth in depths:
            pool = lib.DlSlotPool(ctx, ofdm, ports, depth, max_tb, wire_cfg=wire_cfg)
            ph = pool.handle
            sid = C.c_uint32()

            spans = {"wait": 0.0, "pdsch": 0.0, "modulate": 0.0}   # host time inside the calls (seconds, timed region)

            def pump(n_slots):
                ring = []
                clock = time.perf_counter
                for k in range(n_slots):
                    if len(ring) == depth:
                        old = ring.pop(0)
                        t = clock()
                        assert h.nrphy_dl_slot_wait(ph, old) == 0
                        spans["wait"] += clock() - t
                        assert h.nrphy_dl_slot_close(ph, old) == 0
                    assert h.nrphy_dl_slot_open(ph, C.byref(sid)) == 0
                    pr, tbp = refs[k % len(refs)]
                    t = clock()
                    assert h.nrphy_dl_slot_pdsch(ph, sid.value, 1, pr, tbp) == 0
                    t1 = clock()
                    assert h.nrphy_dl_slot_modulate(ph, sid.value, k % 2, None, None) == 0
                    spans["pdsch"] += t1 - t
                    spans["modulate"] += clock() - t1
                    ring.append(sid.value)
                for old in ring:
                    assert h.nrphy_dl_slot_wait(ph, old) == 0
                    assert h.nrphy_dl_slot_close(ph, old) == 0

            pump(64)
            for key in spans:
                spans[key] = 0.0
            t0 = time.perf_counter()
            pump(total)
            dt = time.perf_counter() - t0
            pool.destroy()
            r = {"leg": leg, "in_flight": depth, "slots_per_sec": round(total / dt, 1), "ms_per_slot": round(1e3 * dt / total, 4),
                 "pcie_bytes_down": int(np.mean([p.tb_size_bytes for p, _ in items])) + 4096,
                 "pcie_bytes_up": ports * slot_samples * sample_bytes,
                 "host_us_per_slot": {key: round(1e6 * v / total, 1) for key, v in spans.items()}}
            results.append(r)
            if verbose:
                print("slot pipeline %s, live traffic, %d in flight: %.0f slots/s (%.3f ms per slot); PCIe per slot: %d B down "
                      "(transport block + plan tables), %d B up (IQ); host time inside the calls, us per slot: %s" % (
                          leg, depth, r["slots_per_sec"], r["ms_per_slot"], r["pcie_bytes_down"], r["pcie_bytes_up"],
                          r["host_us_per_slot"]), flush=True)
    return results


def dl_slot_pipeline_threads(ctx=None, thread_counts=(1, 2, 4), depth=4, total=1200, verbose=True):
    """The wire-format leg of dl_slot_pipeline with several submitting threads on ONE pool (a DU runs one upper-PHY thread per
    cell): every thread opens, fills, modulates and waits for its own slots, `depth` of them in flight per thread.  The calls
    release Python's interpreter lock, so the threads really overlap inside the library and the HIP runtime."""
    import ctypes as C
    import threading
    import backends
    import cases
    abi, lib = backends.abi, backends.pkg.lib
    ctx = ctx or lib.Context(0)
    h = ctx.lib
    rng = np.random.default_rng(11)
    items = live_pdu_pool(rng)
    _, ports, subc, ofdm = cases.baseline_config(3)
    max_tb = max(p.tb_size_bytes for p, _ in items)
    refs = [(C.byref(p), (C.c_void_p * 1)(tb.ctypes.data)) for p, tb in items]
    wire = abi.IqWireCfg(abi.AmplitudeCfg(0, 1, -(10.0 * float(np.log10(subc)) + 12.0), 1.0, -0.1), 32767.0)   # the reference's radio unit setting (bench.py)
    results = []
    for n_threads in thread_counts:
        pool = lib.DlSlotPool(ctx, ofdm, ports, depth * n_threads, max_tb, wire_cfg=wire)
        ph = pool.handle
        errors = []

        def pump(n_slots, first):
            sid = C.c_uint32()
            ring = []
            try:
                for k in range(n_slots):
                    if len(ring) == depth:
                        old = ring.pop(0)
                        assert h.nrphy_dl_slot_wait(ph, old) == 0 and h.nrphy_dl_slot_close(ph, old) == 0
                    assert h.nrphy_dl_slot_open(ph, C.byref(sid)) == 0
                    pr, tbp = refs[(first + k) % len(refs)]
                    assert h.nrphy_dl_slot_pdsch(ph, sid.value, 1, pr, tbp) == 0
                    assert h.nrphy_dl_slot_modulate(ph, sid.value, k % 2, None, None) == 0
                    ring.append(sid.value)
                for old in ring:
                    assert h.nrphy_dl_slot_wait(ph, old) == 0 and h.nrphy_dl_slot_close(ph, old) == 0
            except AssertionError as e:   # (reported by the caller: an exception in a thread would otherwise go unnoticed)
                errors.append(e)

        def run(n_each):
            threads = [threading.Thread(target=pump, args=(n_each, 17 * t)) for t in range(n_threads)]
            t0 = time.perf_counter()
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            return time.perf_counter() - t0

        run(64)
        dt = run(total // n_threads)
        pool.destroy()
        assert not errors, errors
        done = (total // n_threads) * n_threads
        r = {"leg": "tb_in_ci16_iq_out", "submitting_threads": n_threads, "in_flight_per_thread": depth,
             "slots_per_sec": round(done / dt, 1), "ms_per_slot": round(1e3 * dt / done, 4)}
        results.append(r)
        if verbose:
            print("slot pipeline tb_in_ci16_iq_out, live traffic, %d submitting thread(s) x %d in flight: %.0f slots/s (%.3f ms per slot)"
                  % (n_threads, depth, r["slots_per_sec"], r["ms_per_slot"]), flush=True)
    return results


def seams_apart(ctx=None, total=300, verbose=True):
    """What round 3's adaptors did per slot, for comparison: seam A through the asynchronous queue (transport block down, grid
    up), then seam C through the blocking host-span modulator (grid down, IQ up), one slot in flight."""
    import backends
    import cases
    lib = backends.pkg.lib
    ctx = ctx or lib.Context(0)
    rng = np.random.default_rng(11)
    items = live_pdu_pool(rng)
    _, ports, subc, ofdm = cases.baseline_config(3)
    oplan = lib.OfdmPlan(ctx, ofdm, ports)
    q = lib.PdschAsyncQueue(ctx, 1, ports, subc, max(p.tb_size_bytes for p, _ in items))
    box = {}

    def one(k):
        pdu, tb = items[k % len(items)]
        assert q.submit(pdu, tb, lambda st, g: box.__setitem__("g", g))
        q.wait()
        oplan.modulate_slot_host(box["g"], k % 2)

    for k in range(20):
        one(k)
    t0 = time.perf_counter()
    for k in range(total):
        one(k)
    dt = time.perf_counter() - t0
    q.close()
    oplan.close()
    grid_bytes = ports * 14 * subc * 4
    r = {"leg": "seam_a_async_then_seam_c_host", "in_flight": 1, "slots_per_sec": round(total / dt, 1), "ms_per_slot": round(1e3 * dt / total, 4),
         "pcie_bytes_down": int(np.mean([p.tb_size_bytes for p, _ in items])) + 4096 + grid_bytes,
         "pcie_bytes_up": grid_bytes + ports * lib.slot_size(ofdm, 0) * 8}
    if verbose:
        print("seams apart (asynchronous seam A, grid to the host, blocking seam C), live traffic, 1 in flight: %.0f slots/s (%.3f ms per "
              "slot); PCIe per slot: %d B down, %d B up" % (r["slots_per_sec"], r["ms_per_slot"], r["pcie_bytes_down"], r["pcie_bytes_up"]),
              flush=True)
    return r


if __name__ == "__main__":
    if "--live" in sys.argv:
        live_traffic()
    elif "--pipeline" in sys.argv:
        dl_slot_pipeline()
        dl_slot_pipeline_threads()
        seams_apart()
    else:
        main()
        live_traffic()
        dl_slot_pipeline()
        dl_slot_pipeline_threads()
        seams_apart()
