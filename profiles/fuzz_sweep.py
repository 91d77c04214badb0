#!/usr/bin/env python3
"""Larger randomised sweeps than the test-suite runs, GPU against the oracle through the C ABI (the numbers DESIGN.md section
2 quotes).  Usage (GPU box, repository root): python3 profiles/fuzz_sweep.py [pdsch] [mutated] [mutated_ctrl] [encode] [batched] [slot] [dlslot] [async] [ctrl_batched] [plan] [pusch] [rx] [ofdm] [wire] [csi] [dlctrl] [demod] [lower]   (default: all)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import backends  # noqa: E402
import cases  # noqa: E402

# NRPHY_FUZZ_SEED shifts every generator seed: further sweeps over fresh configurations (the default 0 gives the quoted runs).
BASE = int(os.environ.get("NRPHY_FUZZ_SEED", "0"))
abi, lib = backends.abi, backends.pkg.lib
o = backends.oracle()
ctx = lib.Context(0) if "--oracle-only" not in sys.argv else None


def f32(raw):
    return (raw.astype(np.uint32) << 16).view(np.float32)


def pdsch():
    bad = n = 0
    for seed in range(6):
        rng = np.random.default_rng(BASE + 1000 + seed)
        for pdu, P, S in cases.random_pdus(o.tbs, rng, 80):
            if o.validate(pdu) != 0 or o.derive(pdu)["nof_re"] == 0:
                continue
            tb = cases.random_tb(rng, pdu)
            d = o.derive(pdu)
            want, orm, oscr = o.pdsch_process(pdu, tb, P, S, taps=True, codeword_bits=d["codeword_bits"])
            rng.integers(0, 3)  # the CPU leg of this sweep draws which reference processor to use here; keep the streams aligned
            got, rm, scr = ctx.pdsch_process_host(pdu, tb, P, S, taps=True)
            n += 1
            if not (np.array_equal(got, want) and np.array_equal(rm, orm) and np.array_equal(scr, oscr)):
                bad += 1
                print("PDSCH MISMATCH seed", seed, d, flush=True)
    print("pdsch: %d random PDUs, %d mismatches" % (n, bad), flush=True)
    return bad


def same_grid(a, b):
    """Equal cbf16 grids; where a PDU's own parameters make values not-a-number (a NaN power ratio), NaN must meet NaN -- which
    NaN the host's and the device's arithmetic produce is not defined."""
    nan_a = ((a & 0x7F80) == 0x7F80) & ((a & 0x7F) != 0)
    nan_b = ((b & 0x7F80) == 0x7F80) & ((b & 0x7F) != 0)
    return np.array_equal(nan_a, nan_b) and np.array_equal(np.where(nan_a, 0, a), np.where(nan_b, 0, b))


def mutated(device=True):
    """Valid random PDUs with one or two fields overwritten at random (profiles/fuzz_validators_cpu.py: mutate); whatever both
    validators still accept -- unusual but legal corners: odd identities, power ratios, reserved patterns, symbol ranges -- must
    come out of the device exactly as out of the oracle.  device=False runs the oracle side alone (for the sanitizer build)."""
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import fuzz_validators_cpu as fv
    rng = np.random.default_rng(BASE + 31337)
    bad = n = tried = 0
    while n < 150 and tried < 4000:
        for pdu, P, S in cases.random_pdus(o.tbs, rng, 10):
            tried += 1
            fields = [fv.mutate(rng, pdu) for _ in range(int(rng.integers(1, 3)))]
            if lib.validate(pdu) != 0 or o.validate(pdu) != 0:
                continue
            # the caller's side of the contract: a weight array of the size the (possibly overwritten) counts announce
            cases.attach_weights(pdu, (rng.standard_normal((pdu.nof_prg, pdu.nof_ports, pdu.nof_layers, 2)) / 2).astype(np.float32))
            d = o.derive(pdu)
            hi = max((i for i in range(abi.MAX_RB) if (pdu.prb_mask[i // 64] >> (i % 64)) & 1), default=-1)
            if d["nof_re"] == 0 or d["nof_codeblocks"] > 40 or 12 * (hi + 1) > S or pdu.nof_ports > P:
                continue
            tb = cases.random_tb(rng, pdu)
            try:
                want, orm, oscr = o.pdsch_process(pdu, tb, P, S, taps=True, codeword_bits=d["codeword_bits"])
            except AssertionError:   # refused after the derivation (more codeblocks than resource elements): the device must refuse too
                if device:
                    try:
                        ctx.pdsch_process_host(pdu, tb, P, S, taps=True)
                        bad += 1
                        print("MUTATED PDU: the oracle refuses, the device accepts; overwritten:", fields, d, flush=True)
                    except lib.NrphyError:
                        pass
                continue
            n += 1
            if device:
                got, rm, scr = ctx.pdsch_process_host(pdu, tb, P, S, taps=True)
                if not (same_grid(got, want) and np.array_equal(rm, orm) and np.array_equal(scr, oscr)):
                    bad += 1
                    print("MUTATED PDU MISMATCH", n, "overwritten:", fields, "grid", np.array_equal(got, want), "rm", np.array_equal(rm, orm),
                          "scrambled", np.array_equal(scr, oscr), d, flush=True)
    print("mutated pdsch: %d PDUs with overwritten fields that stay valid (of %d tried), %d mismatches" % (n, tried, bad), flush=True)
    return bad


def mutated_ctrl(device=True):
    """The same for the other grid writers: PDCCH, SS/PBCH and NZP-CSI-RS descriptors with overwritten fields that both validators
    still accept, into grids full of other data."""
    import ctypes as C_
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import fuzz_validators_cpu as fv
    h = lib.load()
    rng = np.random.default_rng(BASE + 27182)
    total = 0

    def weights(obj, count):
        f = (rng.standard_normal(2 * count) / 2).astype(np.float32)
        obj._keepalive = f
        obj.precoding = f.ctypes.data_as(C_.POINTER(C_.c_float))

    for kind in ("pdcch", "ssb", "csi"):
        bad = n = tried = 0
        while n < 150 and tried < 6000:
            tried += 1
            if kind == "pdcch":
                obj = cases.random_pdcch(rng)
            elif kind == "ssb":
                obj = cases.random_ssb(rng, int(rng.integers(24, 107)), int(rng.integers(1, 5)))
            else:
                obj = cases.csi_rs_cases(rng)[int(rng.integers(0, 4))][1]
            fields = [fv.mutate(rng, obj) for _ in range(int(rng.integers(1, 3)))]
            if kind == "pdcch":
                if h.nrphy_pdcch_validate(C_.byref(obj)) != 0 or o.pdcch_validate(obj) != 0:
                    continue
                weights(obj, obj.nof_prg * obj.nof_ports)
                ports, subc = 4, 12 * (obj.bwp_start_rb + obj.bwp_size_rb)
            elif kind == "ssb":
                if h.nrphy_ssb_validate(C_.byref(obj)) != 0 or o.ssb_validate(obj) != 0:
                    continue
                ports, subc = 4, 12 * cases.ssb_grid_rb(obj)
            else:
                if h.nrphy_csi_rs_validate(C_.byref(obj)) != 0 or o.csi_rs_validate(obj) != 0:
                    continue
                weights(obj, obj.nof_ports * obj.nof_ports)
                ports, subc = 4, 12 * (obj.start_rb + obj.nof_rb)
            if subc > 12 * abi.MAX_RB:
                continue
            grid = (rng.standard_normal((ports, 14, subc, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
            want = {"pdcch": o.pdcch_process, "ssb": o.ssb_process, "csi": o.csi_rs_map}[kind](obj, grid)
            n += 1
            if device:
                got = {"pdcch": ctx.pdcch_process_host, "ssb": ctx.ssb_process_host, "csi": ctx.csi_rs_map_host}[kind](obj, grid)
                if not same_grid(got, want):
                    bad += 1
                    print("MUTATED", kind.upper(), "MISMATCH", n, "overwritten:", fields, flush=True)
        print("mutated %s: %d descriptors with overwritten fields that stay valid (of %d tried), %d mismatches" % (kind, n, tried, bad),
              flush=True)
        total += bad
    return total


def encode():
    """Seam B (nrphy_pdsch_encode_host: segmentation, LDPC, rate matching, interleaving from the encoder configuration alone): random
    base graph, redundancy version, modulation, layers, limited-buffer size, transport-block and codeword sizes; and nrphy_dft_run on
    every size of the reference's generic DFT with random batches."""
    import torch
    rng = np.random.default_rng(BASE + 14142)
    bad = n = 0
    while n < 150:
        bg, rv, qm, layers = int(rng.integers(1, 3)), int(rng.integers(0, 4)), int(rng.choice([2, 4, 6, 8])), int(rng.integers(1, 5))
        tb_bytes = int(rng.choice([rng.integers(3, 60), rng.integers(60, 1200), rng.integers(1200, 20000)]))
        if bg == 2 and tb_bytes * 8 > 3824 and rng.integers(0, 2):
            continue
        rate = float(rng.uniform(0.08, 0.93))
        nsym = max(layers, int(tb_bytes * 8 / rate / qm) // layers * layers)
        if nsym * qm > 1200000:
            continue
        nref = int(rng.choice([0, 0, int(rng.integers(3000, 30000))]))
        tb = rng.integers(0, 256, tb_bytes, dtype=np.uint8)
        try:
            want = o.pdsch_encode_cfg(bg, rv, qm, nref, layers, nsym, tb)
        except AssertionError:
            continue   # a configuration the oracle refuses (e.g. more codeblocks than bits): nothing to compare
        bits, packed = ctx.pdsch_encode_host(bg, rv, qm, nref, layers, nsym, tb)
        n += 1
        if not (np.array_equal(packed, want[: packed.size]) and np.array_equal(np.packbits(bits), packed) and bits.size == nsym * qm):
            bad += 1
            print("ENCODER MISMATCH", bg, rv, qm, nref, layers, nsym, tb_bytes, flush=True)
    print("encoder seam: %d random configurations, %d mismatches" % (n, bad), flush=True)
    bad2 = 0
    sizes = [128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 4608, 6144, 9216, 12288, 18432, 24576, 36864, 49152]
    for t in range(40):
        size, inverse, batch = int(rng.choice(sizes)), int(rng.integers(0, 2)), int(rng.integers(1, 8))
        x = ((rng.standard_normal((batch, size)) + 1j * rng.standard_normal((batch, size))) * rng.uniform(0.1, 10)).astype(np.complex64)
        d_in = torch.from_numpy(x.view(np.float32)).cuda()
        d_out = torch.zeros_like(d_in)
        ctx.dft(size, inverse, batch, d_in, d_out)
        ctx.synchronize()
        out = d_out.cpu().numpy().view(np.complex64)
        for i in range(batch):
            want = o.dft(x[i], inverse)
            if np.abs(out[i] - want).max() / np.abs(want).max() >= 1e-5:
                bad2 += 1
                print("DFT MISMATCH", size, inverse, batch, i, flush=True)
                break
    print("dft: 40 random (size, direction, batch), %d mismatches" % bad2, flush=True)
    return bad + bad2


def batched():
    """The device-pointer forms with several rows per call and padded strides: soft demodulator spans, soft-bit descrambling of
    several codewords (also in place), amplitude controller and cf32 -> ci16 on strided buffers, fronthaul compression rows."""
    import torch
    rng = np.random.default_rng(BASE + 17320)
    bad = n = 0
    for t in range(60):
        modulation = int(rng.choice([0, 1, 2, 4, 6, 8]))
        qm = max(modulation, 1)
        spans, length = int(rng.integers(1, 9)), int(rng.integers(1, 3000))
        sym, noise = cases.demod_inputs(rng, modulation, spans * length, int(rng.integers(0, 3)))
        d_llr = torch.zeros(spans * length * qm + 64, dtype=torch.int8, device="cuda")
        ctx.demodulate_soft(modulation, spans, length, torch.from_numpy(np.ascontiguousarray(sym).view(np.float32)).cuda(),
                            torch.from_numpy(noise).cuda(), d_llr)
        ctx.synchronize()
        got = d_llr.cpu().numpy()
        ok = not got[spans * length * qm:].any()
        for r_ in range(spans):
            want = o.demodulate_soft(modulation, sym[r_ * length:(r_ + 1) * length], noise[r_ * length:(r_ + 1) * length])
            ok = ok and np.array_equal(got[r_ * length * qm:(r_ + 1) * length * qm], want)
        n += 1
        if not ok:
            bad += 1
            print("BATCHED DEMOD MISMATCH", modulation, spans, length, flush=True)
    for t in range(40):
        n_cw, length = int(rng.integers(1, 7)), int(rng.choice([rng.integers(1, 300), rng.integers(1, 70000), rng.integers(65000, 140000)]))
        stride = length + int(rng.integers(0, 40))
        c_init = rng.integers(0, 1 << 31, n_cw).astype(np.uint32)
        llr = rng.integers(-128, 128, (n_cw, stride)).astype(np.int8)
        d_in = torch.from_numpy(llr.copy()).cuda()
        in_place = bool(rng.integers(0, 2))
        d_out = d_in if in_place else torch.full((n_cw, stride), 99, dtype=torch.int8, device="cuda")
        ctx.llr_descramble(torch.from_numpy(c_init.view(np.int32)).cuda(), n_cw, length, d_in, stride, d_out, stride)
        ctx.synchronize()
        got = d_out.cpu().numpy()
        ok = True
        for r_ in range(n_cw):
            ok = ok and np.array_equal(got[r_, :length], o.prg_apply_xor_llr(int(c_init[r_]), 0, llr[r_, :length]))
            ok = ok and np.array_equal(got[r_, length:], llr[r_, length:] if in_place else np.full(stride - length, 99, np.int8))
        n += 1
        if not ok:
            bad += 1
            print("BATCHED DESCRAMBLER MISMATCH", n_cw, length, stride, in_place, flush=True)
    for t in range(40):
        n_buf, ns = int(rng.integers(1, 9)), int(rng.integers(1, 5000))
        stride = ns + int(rng.integers(0, 33))
        x = ((rng.standard_normal((n_buf, stride)) + 1j * rng.standard_normal((n_buf, stride))) * rng.uniform(0.05, 2.0)).astype(np.complex64)
        cfg = abi.AmplitudeCfg(int(rng.integers(0, 2)), int(rng.integers(0, 2)), float(rng.uniform(-12, 6)), float(rng.uniform(0.5, 2)),
                               float(rng.uniform(-12, -0.1)))
        scale = float(rng.choice([32767.0, 1000.0, 20000.0]))
        d_x = torch.from_numpy(x.view(np.float32).copy()).cuda()
        d_y = torch.zeros_like(d_x)
        d_st = torch.zeros((n_buf, 4), dtype=torch.int32, device="cuda")
        d16 = torch.zeros((n_buf, stride, 2), dtype=torch.int16, device="cuda")
        ctx.amplitude_control(cfg, n_buf, ns, d_x, d_y, d_st, in_stride=stride, out_stride=stride)
        ctx.iq_convert_ci16(n_buf, ns, d_y, scale, d16, in_stride=stride, out_stride=stride)
        ctx.synchronize()
        y, st, y16 = d_y.cpu().numpy().view(np.complex64).reshape(n_buf, stride), d_st.cpu().numpy(), d16.cpu().numpy()
        ok = True
        for r_ in range(n_buf):
            want, wm = o.amplitude_control(cfg, x[r_, :ns])
            ok = ok and np.array_equal(y[r_, :ns].view(np.uint32), want.view(np.uint32)) and not y[r_, ns:].any()
            ok = ok and np.array_equal(y16[r_, :ns].reshape(-1), o.iq_convert_ci16(want, scale)) and not y16[r_, ns:].any()
            if cfg.kind == 0:   # the clipping implementation measures
                ok = ok and st[r_, 2] == wm["nof_clipped"] and st[r_, 1].view(np.float32) == np.float32(wm["stats"].peak_power)
        n += 1
        if not ok:
            bad += 1
            print("BATCHED AMPLITUDE / CI16 MISMATCH", n_buf, ns, stride, cfg.kind, cfg.enable_clipping, flush=True)
    for t in range(40):
        rows, nprb = int(rng.integers(1, 9)), int(rng.integers(1, 120))
        cfg = abi.OfhCompressionCfg(int(rng.integers(0, 2)), int(rng.integers(8, 17)), float(rng.uniform(0.2, 1.5)))
        rec = ctx.lib.nrphy_ofh_compressed_prb_bytes(C.byref(cfg))
        rstride, ostride = 12 * nprb + 12 * int(rng.integers(0, 3)), rec * nprb + int(rng.integers(0, 9))
        prbs = (((rng.standard_normal((rows, rstride, 2)) * rng.uniform(0.01, 1.5)).astype(np.float32).view(np.uint32)) >> 16).astype(np.uint16)
        d_p = torch.from_numpy(prbs.view(np.uint32).reshape(rows, rstride).view(np.int32).copy()).cuda()
        d_o = torch.full((rows, ostride), 0x5A, dtype=torch.uint8, device="cuda")
        ctx.ofh_compress(cfg, rows, nprb, d_p, d_o, row_stride=rstride, out_row_stride=ostride)
        ctx.synchronize()
        got = d_o.cpu().numpy()
        ok = True
        for r_ in range(rows):
            want = o.ofh_compress(cfg, prbs[r_, : 12 * nprb].reshape(nprb, 12, 2))
            ok = ok and np.array_equal(got[r_, : rec * nprb], want) and np.all(got[r_, rec * nprb:] == 0x5A)
        n += 1
        if not ok:
            bad += 1
            print("BATCHED OFH MISMATCH", rows, nprb, cfg.type, cfg.data_width, flush=True)
    print("batched device forms: %d random calls (demodulator, descrambler, amplitude / ci16, fronthaul), %d mismatches" % (n, bad), flush=True)
    return bad


def slot():
    """nrphy_pdsch_process_slot_host: two to six PDUs with disjoint PRB ranges, different users, modulations, rates, layer counts and
    symbol ranges into ONE grid (the FAPI slot batch); the oracle processes them one by one, the union of their resource elements is
    the expected grid (each writes zeros elsewhere)."""
    rng = np.random.default_rng(BASE + 22360)
    bad = n = 0
    while n < 30:
        ports = int(rng.integers(1, 5))
        bwp = int(rng.integers(24, 273))
        k = int(rng.integers(2, 7))
        cuts = np.sort(rng.choice(np.arange(1, bwp), k - 1, replace=False)) if bwp > k else None
        if cuts is None:
            continue
        edges = [0] + [int(c) for c in cuts] + [bwp]
        pdus, tbs_ = [], []
        for u in range(k):
            lo, hi = edges[u], edges[u + 1]
            if hi - lo > 2 and rng.integers(0, 2):
                lo += 1   # a gap between two users
            layers = int(rng.integers(1, ports + 1))
            qm, rate = int(rng.choice([2, 4, 6, 8])), float(rng.uniform(100, 940))
            start = int(rng.integers(0, 3))
            nsym = int(rng.integers(6, 14 - start + 1))
            dmrs = sorted(int(x) for x in rng.choice(np.arange(start, start + nsym), int(rng.integers(1, 4)), replace=False))
            groups = int(rng.integers((layers + 1) // 2, 3))
            tb_bits = o.tbs(nsym, len(dmrs) * 6 * groups, 0, qm, rate, layers, hi - lo)
            if tb_bits < 24:
                break
            r_ = rate / 1024
            bg = 2 if (tb_bits <= 292 or (tb_bits <= 3824 and r_ <= 0.67) or r_ <= 0.25) else 1
            w = ((rng.standard_normal((1, ports, layers)) + 1j * rng.standard_normal((1, ports, layers))) / 2).astype(np.complex64)
            pdus.append(abi.make_pdu(slot_index=int(rng.integers(0, 20)), rnti=int(rng.integers(1, 65535)), n_id=int(rng.integers(0, 1024)),
                                     bwp_size_rb=bwp, qm=qm, dmrs_symbols=tuple(dmrs), prb_start=lo, prb_count=hi - lo, start_symbol=start,
                                     nof_symbols=nsym, base_graph=bg, precoding=w, tb_size_bytes=tb_bits // 8,
                                     nof_cdm_groups_without_data=groups, rv=int(rng.integers(0, 4))))
            tbs_.append(None)
        if len(pdus) != k or any(o.validate(q) != 0 or o.derive(q)["nof_re"] == 0 for q in pdus):
            continue
        tbs_ = [cases.random_tb(rng, q) for q in pdus]
        want = np.zeros((ports, 14, bwp * 12, 2), np.uint16)
        for q, tb in zip(pdus, tbs_):
            want |= o.pdsch_process(q, tb, ports, bwp * 12)
        # (an empty grid: what other channels wrote before is kept by the call -- tests/test_gpu_parity.py covers that -- and an element a
        # PDU maps with the value zero could not be told from one it leaves alone)
        got = ctx.pdsch_process_slot_host(pdus, tbs_, np.zeros((ports, 14, bwp * 12, 2), np.uint16))
        n += 1
        if not np.array_equal(got, want):
            bad += 1
            print("SLOT MISMATCH", n, ports, bwp, k, int(np.count_nonzero(got != want)), flush=True)
    print("pdsch slots: %d random slots of 2-6 PDUs in one grid, %d mismatches" % (n, bad), flush=True)
    return bad


def dlslot():
    """The downlink slot pipeline (nrphy_dl_slots_*): random slots on a 106-PRB / four-port cell -- one to four PDSCH PDUs with disjoint
    PRB ranges arriving in one or several calls, a random PDCCH before or after them, sparse puts, sometimes a host grid loaded instead --
    several slots open at once, the grid read back and the IQ of the slot's pinned buffer against the oracle (float and wire format)."""
    rng = np.random.default_rng(BASE + 31415)
    bad = n = 0
    nof_rb, ports = 106, 4
    subc = 12 * nof_rb
    wire = abi.IqWireCfg(abi.AmplitudeCfg(0, 1, -10.0, 1.0, -6.0), 32767.0)
    for use_wire in (False, True):
        ocfg = abi.OfdmConfig(1, nof_rb, 2048, 0, 1.0 / np.sqrt(2048), 3.5e9)
        pool = lib.DlSlotPool(ctx, ocfg, ports, 3, 400000, wire_cfg=wire if use_wire else None)
        open_slots = []

        def finish(entry):
            nonlocal bad, n
            sid, want, slot_index = entry
            assert pool.wait(sid) == 0
            n += 1
            ok = np.array_equal(pool.read_grid(sid), want)
            if not ok:
                print("   grid differs in", int(np.count_nonzero(pool.read_grid(sid) != want)), "halves", flush=True)
            ref_iq = o.ofdm_slot(ocfg, want, slot_index)
            for port in range(ports):
                got = pool.iq(sid, port)
                if use_wire:
                    y, _ = o.amplitude_control(wire.amplitude, ref_iq[port])
                    w16 = o.iq_convert_ci16(y, wire.ci16_scale).reshape(-1, 2).astype(np.int32)
                    if got.shape != w16.shape or int(np.abs(got.astype(np.int32) - w16).max()) > 1:
                        print("   wire IQ port", port, got.shape, w16.shape, int(np.abs(got.astype(np.int32) - w16[:got.shape[0]]).max()) if got.shape[0] <= w16.shape[0] else -1,
                              "slot_index", slot_index, flush=True)
                    ok = ok and got.shape == w16.shape and int(np.abs(got.astype(np.int32) - w16).max()) <= 1
                else:
                    scale = float(np.abs(ref_iq[port]).max())
                    ok = ok and got.shape == ref_iq[port].shape and (scale == 0 or float(np.abs(got - ref_iq[port]).max()) / scale < 1e-5)
            if not ok:
                bad += 1
                print("DL-SLOT MISMATCH", n, "wire" if use_wire else "f32", flush=True)
            pool.close(sid)

        for t in range(24):
            if len(open_slots) == 3:
                finish(open_slots.pop(0))
            sid = pool.open()
            want = np.zeros((ports, 14, subc, 2), np.uint16)
            slot_index = int(rng.integers(0, 2))
            if rng.integers(0, 6) == 0:   # seam C alone: a grid computed elsewhere
                want = (((rng.standard_normal((ports, 14, subc, 2)) * 0.5).astype(np.float32).view(np.uint32)) >> 16).astype(np.uint16)
                pool.load_grid(sid, want)
            else:
                pdcch = cases.random_pdcch(rng, nof_ports_max=ports, nof_rb_grid=nof_rb)
                first = bool(rng.integers(0, 2))
                if first:
                    pool.pdcch(sid, [pdcch])
                    want = o.pdcch_process(pdcch, want)
                k = int(rng.integers(1, 5))
                edges = [0] + sorted(int(c) for c in rng.choice(np.arange(8, nof_rb - 8), k - 1, replace=False)) + [nof_rb]
                pdus, tbs_ = [], []
                for u in range(k):
                    lo, hi = edges[u], edges[u + 1]
                    layers = int(rng.integers(1, 5))
                    qm, rate = int(rng.choice([2, 4, 6, 8])), float(rng.uniform(100, 940))
                    nsym = int(rng.integers(8, 13))
                    groups = int(rng.integers((layers + 1) // 2, 3))
                    tb_bits = o.tbs(nsym, 12 * groups, 0, qm, rate, layers, hi - lo)
                    r_ = rate / 1024
                    bg = 2 if (tb_bits <= 292 or (tb_bits <= 3824 and r_ <= 0.67) or r_ <= 0.25) else 1
                    w = ((rng.standard_normal((1, ports, layers)) + 1j * rng.standard_normal((1, ports, layers))) / 2).astype(np.complex64)
                    q = abi.make_pdu(slot_index=int(rng.integers(0, 20)), rnti=int(rng.integers(1, 65535)), n_id=int(rng.integers(0, 1024)),
                                     bwp_size_rb=nof_rb, qm=qm, dmrs_symbols=(2, 2 + nsym - 4), prb_start=lo, prb_count=hi - lo, start_symbol=2,
                                     nof_symbols=nsym, base_graph=bg, precoding=w, tb_size_bytes=max(3, tb_bits // 8),
                                     nof_cdm_groups_without_data=groups, rv=int(rng.integers(0, 4)))
                    if tb_bits < 24 or o.validate(q) != 0 or o.derive(q)["nof_re"] == 0:
                        continue
                    pdus.append(q)
                    tbs_.append(cases.random_tb(rng, q))
                cut = int(rng.integers(0, len(pdus) + 1))   # the PDUs arrive in up to two calls
                for part_pdus, part_tbs in ((pdus[:cut], tbs_[:cut]), (pdus[cut:], tbs_[cut:])):
                    if part_pdus:
                        assert pool.pdsch(sid, part_pdus, part_tbs) == 0
                for q, tb in zip(pdus, tbs_):
                    part = o.pdsch_process(q, tb, ports, subc)
                    mask = part.view(np.uint32) != 0
                    want.view(np.uint32)[mask] = part.view(np.uint32)[mask]
                if not first:
                    pool.pdcch(sid, [pdcch])
                    want = o.pdcch_process(pdcch, want)
                entries = []
                for _ in range(int(rng.integers(0, 9))):   # a channel the library does not generate: a few finite bf16 pairs on symbol 13
                    v = (rng.standard_normal(2) * 0.5).astype(np.float32).view(np.uint32) >> 16
                    entries.append(abi.GridRe(int(rng.integers(0, ports)), 13, int(rng.integers(0, subc)), int(v[0] | (v[1] << 16))))
                if entries:
                    pool.put(sid, entries)
                    for e in entries:
                        want.view(np.uint32).reshape(ports, 14, subc)[e.port, e.symbol, e.subc] = e.value
            assert pool.modulate(sid, slot_index) == 0
            open_slots.append((sid, want, slot_index))
        while open_slots:
            finish(open_slots.pop(0))
        pool.destroy()
    print("downlink slot pipeline: %d random slots (float and wire format), %d mismatches" % (n, bad), flush=True)
    return bad


def async_queue():
    """nrphy_pdsch_async_*: 60 random PDUs through a queue of three operations in flight (random shapes: the plan cache of a slot misses
    and evicts), every completion once with the PDU's grid, in the three staging modes (NRPHY_ASYNC_ZERO_COPY unset, 1, 3)."""
    import threading
    import time
    bad = n = 0
    for mode in (None, "1", "3"):
        if mode is None:
            os.environ.pop("NRPHY_ASYNC_ZERO_COPY", None)
        else:
            os.environ["NRPHY_ASYNC_ZERO_COPY"] = mode
        rng = np.random.default_rng(BASE + 26457 + (0 if mode is None else int(mode)))
        drawn = [x for x in cases.random_pdus(o.tbs, rng, 24) if o.validate(x[0]) == 0 and o.derive(x[0])["nof_re"] > 0][:20]
        nof_ports, nof_subc = 4, max(x[2] for x in drawn)
        q = lib.PdschAsyncQueue(ctx, 3, nof_ports, nof_subc, max(x[0].tb_size_bytes for x in drawn))
        results, lock = {}, threading.Lock()
        jobs = [(x[0], cases.random_tb(rng, x[0])) for x in drawn]
        for i, (pdu, tb) in enumerate(jobs):
            def on_done(status, grid, i=i):
                with lock:
                    results.setdefault(i, []).append((status, grid))
            while not q.submit(pdu, tb, on_done):
                time.sleep(0.0005)
        q.wait()
        for i, (pdu, tb) in enumerate(jobs):
            n += 1
            got = results.get(i, [])
            if len(got) != 1 or got[0][0] != 0 or not np.array_equal(got[0][1], o.pdsch_process(pdu, tb, nof_ports, nof_subc)):
                bad += 1
                print("ASYNC MISMATCH mode", mode, "job", i, len(got), flush=True)
        q.close()
    os.environ.pop("NRPHY_ASYNC_ZERO_COPY", None)
    print("asynchronous seam: %d random PDUs, three in flight, three staging modes, %d mismatches" % (n, bad), flush=True)
    return bad


def ctrl_batched():
    """The device-pointer forms of the other grid writers: batches of 4-16 NZP-CSI-RS signals, DCIs and SS/PBCH blocks in one call each,
    every item into a device grid of its own (full of other data), plus sparse host writes (nrphy_grid_put) with repeated positions."""
    import torch
    rng = np.random.default_rng(BASE + 31622)
    bad = n = 0
    ports, subc = 4, 106 * 12
    for t in range(12):
        m = int(rng.integers(4, 17))
        kind = ("csi", "pdcch", "ssb")[t % 3]
        items = []
        while len(items) < m:
            if kind == "csi":
                name, cfg, p_, s_ = cases.csi_rs_cases(rng)[int(rng.integers(0, 4))]
                if o.csi_rs_validate(cfg) == 0 and p_ <= ports and 12 * (cfg.start_rb + cfg.nof_rb) <= subc:
                    items.append(cfg)
            elif kind == "pdcch":
                pdu = cases.random_pdcch(rng)
                if 12 * (pdu.bwp_start_rb + pdu.bwp_size_rb) <= subc:
                    items.append(pdu)
            else:
                pdu = cases.random_ssb(rng, 106, int(rng.integers(1, 5)))
                if o.ssb_validate(pdu) == 0 and cases.ssb_grid_rb(pdu) <= 106:
                    items.append(pdu)
        grids = (rng.standard_normal((m, ports, 14, subc, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        d_grid = torch.from_numpy(grids.view(np.uint32).reshape(m, ports, 14, subc).view(np.int32).copy()).cuda()
        call = {"csi": ctx.csi_rs_map, "pdcch": ctx.pdcch_process, "ssb": ctx.ssb_process}[kind]
        call(items, list(range(m)), d_grid, ports, subc)
        ctx.synchronize()
        got = d_grid.cpu().numpy().view(np.uint16).reshape(m, ports, 14, subc, 2)
        ref_call = {"csi": o.csi_rs_map, "pdcch": o.pdcch_process, "ssb": o.ssb_process}[kind]
        for i in range(m):
            n += 1
            if not np.array_equal(got[i], ref_call(items[i], grids[i])):
                bad += 1
                print("BATCHED", kind.upper(), "MISMATCH", t, i, flush=True)
    for t in range(10):
        grid = (rng.standard_normal((ports, 14, subc, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        k = int(rng.integers(1, 400))
        entries = [(int(rng.integers(0, ports)), int(rng.integers(0, 14)), int(rng.integers(0, subc)), int(rng.integers(0, 1 << 32))) for _ in range(k)]
        entries += entries[: k // 4]   # repeated positions: the later entry wins, here with the same value
        want = grid.copy().view(np.uint32).reshape(ports, 14, subc)
        for p_, l_, s_, v_ in entries:
            want[p_, l_, s_] = v_
        d_grid = torch.from_numpy(grid.view(np.uint32).reshape(ports, 14, subc).view(np.int32).copy()).cuda()
        ctx.grid_put(d_grid, ports, subc, entries)
        ctx.synchronize()
        n += 1
        if not np.array_equal(d_grid.cpu().numpy().view(np.uint32), want):
            bad += 1
            print("GRID PUT MISMATCH", t, k, flush=True)
    print("batched grid writers: %d items (CSI-RS, PDCCH, SS/PBCH in device grids; sparse writes), %d mismatches" % (n, bad), flush=True)
    return bad


def plan():
    """The batched path (what bench.py times): groups of 24 random PDUs in ONE plan, each into its own grid of a common shape, run
    twice on the same grids with new transport blocks (the second run must overwrite everything the first one wrote)."""
    import torch
    bad = n = 0
    for group in range(4):
        rng = np.random.default_rng(BASE + 90210 + group)
        drawn = [x for x in cases.random_pdus(o.tbs, rng, 28) if o.validate(x[0]) == 0 and o.derive(x[0])["nof_re"] > 0][:24]
        nof_ports, nof_subc = 4, max(x[2] for x in drawn)
        pdus = [x[0] for x in drawn]
        m = len(pdus)
        pl = None
        d_grid = torch.full((m, nof_ports, 14, nof_subc), 0x7FFF7FFF, dtype=torch.int32, device="cuda")
        for run in range(2):
            offs, tbs, pos = [], [], 0
            for q in pdus:
                tb = cases.random_tb(rng, q)
                offs.append(pos)
                tbs.append(tb)
                pos += (len(tb) + 15) & ~15
            buf = np.zeros(pos + 16, np.uint8)
            for off, tb in zip(offs, tbs):
                buf[off:off + len(tb)] = tb
            if pl is None:
                pl = lib.PdschPlan(ctx, pdus, offs, list(range(m)), m, nof_ports, nof_subc)
            d_rm = torch.zeros(pl.codeword_bits // 8 + 8, dtype=torch.uint8, device="cuda")
            pl.run(torch.from_numpy(buf).cuda(), d_grid, d_cw_rm=d_rm, zero_grids=True)
            ctx.synchronize()
            grids = d_grid.cpu().numpy().view(np.uint16).reshape(m, nof_ports, 14, nof_subc, 2)
            rm = d_rm.cpu().numpy()
            for i, q in enumerate(pdus):
                d = o.derive(q)
                want, orm, _ = o.pdsch_process(q, tbs[i], nof_ports, nof_subc, taps=True, codeword_bits=d["codeword_bits"])
                cw = pl.codeword_offset(i) // 8
                n += 1
                if not (np.array_equal(grids[i], want) and np.array_equal(rm[cw:cw + len(orm)], orm)):
                    bad += 1
                    print("PLAN MISMATCH group", group, "run", run, "pdu", i, d, flush=True)
        pl.close()
    print("pdsch plans: %d PDU runs in plans of 24, %d mismatches" % (n, bad), flush=True)
    return bad


def pusch():
    """The transport-block decoder (nrphy_pusch_decode_batch): random sizes, base graphs, modulations, layers, limited-buffer sizes and
    redundancy-version orders, up to three transmissions each, against pusch_decoder_impl restated on the oracle's codeblock functions
    (tests/cases.py: pusch_decode_expected): verdict, counters, the whole soft buffer after every transmission, the transport block."""
    import torch
    rng = np.random.default_rng(BASE + 5252)
    done = bad = 0
    while done < 40:
        layers, qm = int(rng.integers(1, 5)), int(rng.choice([2, 4, 6, 8]))
        n_prb, nsym = int(rng.integers(2, 60)), int(rng.integers(6, 14))
        rate = float(rng.uniform(100, 800))
        tb_bits = o.tbs(nsym, 12, 0, qm, rate, layers, n_prb)
        if tb_bits < 40 or tb_bits > 120000:
            continue
        r = rate / 1024
        bg = 2 if (tb_bits <= 292 or (tb_bits <= 3824 and r <= 0.67) or r <= 0.25) else 1
        pdu = abi.make_pdu(bwp_size_rb=n_prb, qm=qm, dmrs_symbols=(2,), prb_start=0, prb_count=n_prb, start_symbol=0,
                           nof_symbols=nsym, precoding=abi.identity_precoding(layers), tb_size_bytes=tb_bits // 8, base_graph=bg,
                           tbs_lbrm_bytes=int(rng.choice([3168, 20000, abi.TBS_LBRM_DEFAULT])), nof_cdm_groups_without_data=2)
        if o.validate(pdu) != 0:
            continue
        d = o.derive(pdu)
        nof_sys = d["segment_length"] - 2 * d["lifting_size"]
        if d["nof_re"] == 0 or d["nof_codeblocks"] > 12 or d["n_cb"] <= nof_sys:   # (a buffer within the systematic part: refused on the receive side)
            continue
        C_, n, G = d["nof_codeblocks"], d["full_length"], d["codeword_bits"]
        tb = cases.random_tb(rng, pdu)
        cfg0 = abi.PuschDecoderCfg(bg, qm, 0, layers, d["n_ref"], pdu.tb_size_bytes, G // qm, 6, 1, 1)
        soft_bytes, state_bytes, _ = ctx.pusch_decoder_sizes(cfg0, 1)
        d_soft = torch.full((1, soft_bytes), -7, dtype=torch.int8, device="cuda")
        d_state = torch.full((state_bytes,), 0xA5, dtype=torch.uint8, device="cuda")
        d_tb = torch.zeros((1, pdu.tb_size_bytes + 3), dtype=torch.uint8, device="cuda")
        d_res = torch.zeros((1, 4), dtype=torch.int32, device="cuda")
        soft = np.full((C_, n), -7, np.int8)
        cb_ok = np.ones(C_, np.uint8)
        cb_msg = np.zeros((C_, d["segment_length"]), np.uint8)
        sigma, early = float(rng.uniform(3.0, 9.0)), int(rng.integers(0, 2))
        ok = True
        for tx, rv in enumerate([0] + [int(x) for x in rng.permutation([1, 2, 3])[:2]]):
            cfg = abi.PuschDecoderCfg(bg, qm, rv, layers, d["n_ref"], pdu.tb_size_bytes, G // qm, 6, early, 1 if tx == 0 else 0)
            pdu.rv = rv
            _, rm, _ = o.pdsch_process(pdu, tb, layers, 12 * n_prb, taps=True, codeword_bits=G)
            pdu.rv = 0
            bits = np.unpackbits(rm)[:G].astype(np.float64)
            llr = np.clip(np.rint((1 - 2 * bits) * 8.0 + rng.normal(0, sigma, G)), -120, 120).astype(np.int8)
            want = cases.pusch_decode_expected(o, d, cfg, llr, soft, cb_ok, cb_msg)
            ctx.pusch_decode_batch(cfg, 1, torch.from_numpy(llr[None, :].copy()).cuda(), G, d_soft, d_state, d_tb, d_tb.shape[1], d_res)
            ctx.synchronize()
            res = tuple(int(x) for x in d_res.cpu().numpy()[0])
            ok = ok and res == (int(want[0]), want[1], want[2], want[3]) and np.array_equal(d_soft.cpu().numpy().reshape(C_, n), soft)
            if want[0]:
                ok = ok and np.array_equal(d_tb.cpu().numpy()[0, : pdu.tb_size_bytes], tb)
                break
        done += 1
        if not ok:
            bad += 1
            print("PUSCH DECODER MISMATCH", done, bg, qm, layers, d["n_ref"], C_, d["lifting_size"], flush=True)
    print("ul-sch decoder: %d random transport blocks (up to three transmissions each), %d mismatches" % (done, bad), flush=True)
    return bad


def rx():
    rng = np.random.default_rng(BASE + 424242)
    sizes = cases.LIFTING_SIZES
    bad = 0
    for t in range(200):
        bg = int(rng.integers(1, 3))
        zc = int(rng.choice(sizes[3:]))
        kb, n_short = (22, 66) if bg == 1 else (10, 50)
        k = kb * zc
        crc_id = int(rng.choice([16, 0x24A, 0x24B]))
        crc_len = 16 if crc_id == 16 else 24
        if k - crc_len - 2 <= 0:
            continue
        nf = int(rng.integers(0, max(1, min(k - crc_len - 2, (kb - 2) * zc - 1) // 2)))
        nof_llr = int(rng.integers(k + 2 * zc, n_short * zc + 1))
        _, llr = cases.make_ldpc_llrs(o, rng, bg, zc, nof_llr, crc_id, nf, float(rng.uniform(6, 30)), float(rng.uniform(2, 14)))
        llr[rng.integers(0, nof_llr, max(1, nof_llr // 50))] = 127
        llr[rng.integers(0, nof_llr, max(1, nof_llr // 50))] = 0
        if t % 7 == 0:
            llr[rng.integers(0, nof_llr, max(1, nof_llr // 40))] = -127
        iters, scaling = int(rng.integers(1, 11)), float(rng.choice([0.5, 0.625, 0.75, 0.8, 0.9, 0.99]))
        crc = crc_id if rng.integers(0, 4) else 0
        want = o.ldpc_decode(bg, zc, nf, crc, iters, scaling, llr)
        got = ctx.ldpc_decode_host(bg, zc, nf, crc, iters, scaling, llr)
        if got[0] != want[0] or not np.array_equal(got[1], want[1]):
            bad += 1
            print("DECODER MISMATCH", bg, zc, nf, hex(crc), iters, scaling, nof_llr, flush=True)
    print("decoder: 200 random configurations, %d mismatches" % bad, flush=True)
    total = bad
    bad = 0
    for t in range(300):
        bg = int(rng.integers(1, 3))
        zc = int(rng.choice(sizes))
        n = (66 if bg == 1 else 50) * zc
        nof_sys = ((22 if bg == 1 else 10) - 2) * zc
        qm = int(rng.choice([1, 2, 4, 6, 8]))
        e = qm * int(rng.integers(1, max(2, min(4 * n, 70000) // qm)))
        nf = int(rng.integers(0, max(1, nof_sys // 2)))
        nref = int(rng.choice([0, 0, int(rng.integers(nof_sys + 1, n + 1))]))
        rv, new_data = int(rng.integers(0, 4)), int(rng.integers(0, 2))
        llr = rng.integers(-127, 128, e).astype(np.int8)
        old = rng.integers(-127, 128, n).astype(np.int8)
        want = o.ldpc_rate_dematch(bg, zc, rv, qm, nref, nf, new_data, llr, old)
        got = ctx.ldpc_rate_dematch_host(bg, zc, rv, qm, nref, nf, new_data, llr, old)
        if not np.array_equal(got, want):
            bad += 1
            print("DEMATCHER MISMATCH", bg, zc, e, rv, qm, nref, nf, new_data, flush=True)
    print("rate dematcher: 300 random configurations, %d mismatches" % bad, flush=True)
    total += bad
    bad = 0
    for t in range(300):
        n = int(rng.choice([rng.integers(1, 200), rng.integers(1, 70000), rng.integers(65000, 66100), rng.integers(1, 1 << 21)]))
        c_init = int(rng.integers(0, 1 << 31))
        llr = rng.integers(-128, 128, n).astype(np.int8)
        if not np.array_equal(ctx.llr_descramble_host(c_init, llr), o.prg_apply_xor_llr(c_init, 0, llr)):
            bad += 1
            print("DESCRAMBLER MISMATCH", c_init, n, flush=True)
    print("soft-bit descrambler: 300 random (c_init, length) pairs, %d mismatches" % bad, flush=True)
    return total + bad


def ofdm():
    rng = np.random.default_rng(BASE + 8088)
    bad = 0
    for t in range(100):
        n = int(rng.choice([128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 4608, 6144]))
        mu, ext = int(rng.integers(0, 5)), int(rng.integers(0, 4) == 0)
        bw, ports = int(rng.integers(1, min(275, (n - 1) // 12) + 1)), int(rng.integers(1, 5))
        cfg = abi.OfdmConfig(mu, bw, n, ext, float(rng.uniform(0.01, 2.0)), float(rng.choice([0.0, 7e8, 2.4e9, 3.5e9, 28e9, 39e9])))
        slot = int(rng.integers(0, 1 << mu))
        grid = (rng.standard_normal((ports, 14, bw * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        plan = lib.OfdmPlan(ctx, cfg, ports)
        iq = plan.modulate_slot_host(grid, slot)
        want = o.ofdm_slot(cfg, grid, slot)
        e1 = np.abs(iq - want).max() / np.abs(want).max()
        wo = int(rng.integers(0, (144 * n) // 2048))
        rxs = (rng.standard_normal(want.shape) + 1j * rng.standard_normal(want.shape)).astype(np.complex64)
        got, ref = plan.demodulate_slot_host(rxs, slot, wo), o.ofdm_demod_slot(cfg, rxs, slot, wo)
        ns = 12 if ext else 14
        a, b = f32(got[:, :ns]), f32(ref[:, :ns])
        nb = int((np.abs(a - b) > np.maximum(np.abs(b), 1e-3 * np.abs(b).max()) * 2.0 ** -7).sum())
        if e1 >= 1e-5 or nb or np.mean(got[:, :ns] == ref[:, :ns]) < 0.99:
            bad += 1
            print("OFDM MISMATCH", n, mu, ext, bw, ports, slot, wo, "modulator err %.1e" % e1, "demodulator bad", nb, flush=True)
        plan.close()
    print("ofdm: 100 random configurations, %d mismatches" % bad, flush=True)
    return bad


def wire():
    """nrphy_ofdm_run_ci16 (modulator + amplitude controller + int16 conversion fused) on random configurations and amplitude
    settings: equal to the device modulator followed by the oracle's amplitude controller and conversion up to one LSB at rounding
    ties (> 99.99 % identical), clipped-sample count, processed count and peak power exact, power sum to 1e-4."""
    import torch
    rng = np.random.default_rng(BASE + 16180)
    bad = n = 0
    for t in range(40):
        size = int(rng.choice([256, 512, 1024, 1536, 2048, 4096, 4608, 6144]))
        mu, ext = int(rng.integers(0, 4)), int(rng.integers(0, 5) == 0)
        bw, ports, slots = int(rng.integers(1, min(275, (size - 1) // 12) + 1)), int(rng.integers(1, 4)), int(rng.integers(1, 4))
        ocfg = abi.OfdmConfig(mu, bw, size, ext, float(rng.uniform(0.5, 2.0)) / np.sqrt(size), float(rng.choice([0.0, 2.4e9, 3.5e9])))
        amp = abi.AmplitudeCfg(0, int(rng.integers(0, 2)), float(rng.uniform(-20, 6)), float(rng.choice([1.0, 2.0])), float(rng.uniform(-12, -0.5)))
        wire_cfg = abi.IqWireCfg(amp, float(rng.choice([32767.0, 20000.0, 40000.0])))
        grid = ((rng.standard_normal((slots, ports, 14, bw * 12, 2)) * rng.uniform(0.2, 1.0)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        pl = lib.OfdmPlan(ctx, ocfg, ports)
        d_grid = torch.from_numpy(grid.view(np.uint32).reshape(slots, ports, 14, bw * 12).view(np.int32)).cuda()
        d_slot = torch.from_numpy((np.arange(slots, dtype=np.uint32) % (1 << mu)).view(np.int32)).cuda()
        d_iq16 = torch.zeros((slots, ports, pl.slot_stride, 2), dtype=torch.int16, device="cuda")
        d_stats = torch.zeros((slots * ports, 4), dtype=torch.int32, device="cuda")
        d_iq = torch.zeros((slots, ports, pl.slot_stride, 2), dtype=torch.float32, device="cuda")
        pl.run_ci16(slots, d_grid, wire_cfg, d_iq16, d_slot_index=d_slot, d_stats=d_stats)
        pl.run(slots, d_grid, d_iq, d_slot_index=d_slot)
        ctx.synchronize()
        fused, stats = d_iq16.cpu().numpy(), d_stats.cpu().numpy()
        iq = d_iq.cpu().numpy().view(np.complex64).reshape(slots, ports, -1)
        ok = True
        for s_ in range(slots):
            ssz = lib.slot_size(ocfg, int(s_ % (1 << mu)))
            for p_ in range(ports):
                y, m = o.amplitude_control(wire_cfg.amplitude, iq[s_, p_, :ssz])
                # The fused kernel converts every sample the way the reference's vector loop does (saturating); the reference's scalar
                # tail -- the last (2 n mod 16) floats of a call, where an out-of-range product wraps -- has no counterpart in a fused
                # slot (mi355_nrphy.h, nrphy_ofdm_run_ci16): pad the oracle's call so that every sample lies in its vector part.
                want = o.iq_convert_ci16(np.concatenate([y, np.zeros(8, np.complex64)]), wire_cfg.ci16_scale).reshape(-1, 2)[:ssz]
                st = stats[s_ * ports + p_]
                ok = ok and np.abs(fused[s_, p_, :ssz].astype(np.int32) - want.astype(np.int32)).max() <= 1
                ok = ok and np.mean(fused[s_, p_, :ssz] == want) > 0.9999 and st[3] == ssz and st[2] == m["nof_clipped"]
                ok = ok and st[1].view(np.float32) == np.float32(m["stats"].peak_power)
                ok = ok and abs(st[0].view(np.float32) - m["stats"].sum_power) <= 1e-4 * max(m["stats"].sum_power, 1e-30)
        pl.close()
        n += 1
        if not ok:
            bad += 1
            print("WIRE MISMATCH", size, mu, ext, bw, ports, slots, amp.enable_clipping, amp.input_gain_dB, amp.ceiling_dBFS, wire_cfg.ci16_scale, flush=True)
    print("wire-format ofdm: %d random configurations, %d mismatches" % (n, bad), flush=True)
    return bad


def csi():
    rng = np.random.default_rng(BASE + 8088)
    bad = 0
    for t in range(200):
        row = int(rng.integers(1, 6))
        ports = abi.CSI_ROW_PORTS[row]
        dens = {1: ["three"], 2: ["one", "dot5_even", "dot5_odd"], 3: ["one", "dot5_even", "dot5_odd"], 4: ["one"], 5: ["one"]}[row]
        kmax = {1: 3, 2: 11, 3: 10, 4: 8, 5: 10}[row]
        start = int(rng.integers(0, 200))
        nrb = int(rng.integers(1, 275 - start))
        cp = int(rng.integers(0, 2))
        lmax = (12 if cp else 14) - (2 if row == 5 else 1)
        w = None
        if rng.integers(0, 2):
            w = ((rng.standard_normal((1, ports, ports)) + 1j * rng.standard_normal((1, ports, ports))) / 2).astype(np.complex64)
        cfg = abi.make_csi_rs(row=row, start_rb=start, nof_rb=nrb, k0=int(rng.integers(0, kmax + 1)), l0=int(rng.integers(0, lmax + 1)),
                              density=str(rng.choice(dens)), slot_index=int(rng.integers(0, 20)), cp=cp,
                              scrambling_id=int(rng.integers(0, 1024)), amplitude=float(rng.uniform(0.1, 2)), precoding=w)
        P, S = int(rng.integers(ports, 5)), 12 * (start + nrb)
        grid = (rng.standard_normal((P, 14, S, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        if o.csi_rs_validate(cfg) != 0 or ctx.lib.nrphy_csi_rs_validate(C.byref(cfg)) != 0:
            bad += 1
            continue
        if not np.array_equal(ctx.csi_rs_map_host(cfg, grid), o.csi_rs_map(cfg, grid)):
            bad += 1
            print("CSI-RS MISMATCH", row, start, nrb, cp, flush=True)
    print("csi-rs: 200 random configurations, %d mismatches" % bad, flush=True)
    return bad


def dlctrl():
    """PDCCH and SS/PBCH block processors: random PDUs into grids full of other data."""
    rng = np.random.default_rng(BASE + 515151)
    bad = n = 0
    for t in range(1500):
        pdu = cases.random_pdcch(rng)
        grid = (rng.standard_normal((4, 14, 52 * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        n += 1
        if not np.array_equal(ctx.pdcch_process_host(pdu, grid), o.pdcch_process(pdu, grid)):
            bad += 1
            print("PDCCH MISMATCH", t, flush=True)
    print("pdcch: %d random PDUs, %d mismatches" % (n, bad), flush=True)
    n2 = bad2 = 0
    for t in range(600):
        nrb = int(rng.integers(24, 107))
        ports = int(rng.integers(1, 5))
        pdu = cases.random_ssb(rng, nrb, ports)
        grid = (rng.standard_normal((ports, 14, nrb * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        if o.ssb_validate(pdu) != 0:
            continue
        n2 += 1
        if not np.array_equal(ctx.ssb_process_host(pdu, grid), o.ssb_process(pdu, grid)):
            bad2 += 1
            print("SSB MISMATCH", t, flush=True)
    print("ssb: %d random PDUs, %d mismatches" % (n2, bad2), flush=True)
    return bad + bad2


def demod():
    """Soft demodulator: random span lengths, the three input kinds of tests/cases.py, every modulation."""
    rng = np.random.default_rng(BASE + 626262)
    bad = n = 0
    for t in range(1200):
        modulation = int(rng.choice([0, 1, 2, 4, 6, 8]))
        length = int(rng.integers(1, 5000))
        sym, noise = cases.demod_inputs(rng, modulation, length, int(rng.integers(0, 3)))
        n += 1
        if not np.array_equal(ctx.demodulate_soft_host(modulation, sym, noise), o.demodulate_soft(modulation, sym, noise)):
            bad += 1
            print("DEMOD MISMATCH", modulation, length, flush=True)
    print("soft demodulator: %d random spans, %d mismatches" % (n, bad), flush=True)
    return bad


def lower():
    """Amplitude controller, cf32 -> ci16, OFH compression on random buffers and parameters."""
    rng = np.random.default_rng(BASE + 737373)
    bad = n = 0
    for t in range(400):
        nsamp = int(rng.integers(1, 20000))
        x = ((rng.standard_normal(nsamp) + 1j * rng.standard_normal(nsamp)) * rng.uniform(0.05, 2.0)).astype(np.complex64)
        cfg = abi.AmplitudeCfg(int(rng.integers(0, 2)), int(rng.integers(0, 2)), float(rng.uniform(-12, 6)), float(rng.uniform(0.5, 2)),
                               float(rng.uniform(-12, -0.1)))
        y, m = ctx.amplitude_control_host(cfg, x)
        oy, om = o.amplitude_control(cfg, x)
        scale = float(rng.choice([32767.0, 1000.0, 40000.0]))
        n += 1
        ok = np.array_equal(y.view(np.uint32), oy.view(np.uint32)) and np.array_equal(ctx.iq_convert_ci16_host(y, scale),
                                                                                     o.iq_convert_ci16(oy, scale))
        if not ok:
            bad += 1
            print("LOWER-PHY MISMATCH amplitude/ci16", t, flush=True)
    for t in range(400):
        nprb = int(rng.integers(1, 274))
        prbs = (((rng.standard_normal((nprb, 12, 2)) * rng.uniform(0.01, 1.5)).astype(np.float32).view(np.uint32)) >> 16).astype(np.uint16)
        cfg = abi.OfhCompressionCfg(int(rng.integers(0, 2)), int(rng.integers(8, 17)), float(rng.uniform(0.2, 1.5)))
        n += 1
        if not np.array_equal(ctx.ofh_compress_host(cfg, prbs), o.ofh_compress(cfg, prbs)):
            bad += 1
            print("OFH MISMATCH", cfg.type, cfg.data_width, nprb, flush=True)
    print("lower-PHY tail: %d random cases, %d mismatches" % (n, bad), flush=True)
    return bad


if __name__ == "__main__":
    legs = {"pdsch": pdsch, "mutated": mutated, "mutated_ctrl": mutated_ctrl, "encode": encode, "batched": batched, "slot": slot, "dlslot": dlslot, "async": async_queue, "ctrl_batched": ctrl_batched, "plan": plan, "pusch": pusch, "rx": rx, "ofdm": ofdm, "wire": wire, "csi": csi, "dlctrl": dlctrl, "demod": demod, "lower": lower}
    if "--oracle-only" in sys.argv:   # the oracle side of the mutated leg alone, for the CPU sanitizer build
        sys.exit(mutated(device=False) + mutated_ctrl(device=False))
    which = sys.argv[1:] or list(legs)
    total = sum(legs[w]() for w in which)
    sys.exit(1 if total else 0)
