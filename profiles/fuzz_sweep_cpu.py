#!/usr/bin/env python3
"""The CPU leg of the randomised sweeps: the ORACLE against the COMPILED REFERENCE (oracle/_ref/libsrsref.so, build container only)
on the configurations profiles/fuzz_sweep.py draws for the device -- same generators, same seeds, NRPHY_FUZZ_SEED shifts them.
Usage (build container, repository root): python3 profiles/fuzz_sweep_cpu.py [pdsch] [rx] [csi] [dlctrl] [demod] [lower]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import backends  # noqa: E402
import cases  # noqa: E402

BASE = int(os.environ.get("NRPHY_FUZZ_SEED", "0"))
abi = backends.abi
o, r = backends.oracle(), backends.ref()
assert r is not None, "compiled reference not built (make -C oracle ref)"


def grid_verdict(got, want):
    """'exact'; or 'ulp' when the cbf16 grids differ at a handful of elements by a last-place rounding -- at most one bf16 unit in the
    last place of the value, or of a thousandth of the grid's largest value where layers cancel: the reference's scalar tail loop
    (channel_precoder_avx2.cpp:119-127), whose imaginary part this build of the reference contracts with the other product rounded
    first than in its vector loop; or 'wrong'."""
    if np.array_equal(got, want):
        return "exact", 0
    where = got != want
    a = (got[where].astype(np.uint32) << 16).view(np.float32)
    b = (want[where].astype(np.uint32) << 16).view(np.float32)
    scale = float(np.abs((want.astype(np.uint32) << 16).view(np.float32)).max())
    if a.size <= 8 and np.all(np.abs(a - b) <= np.maximum(np.abs(b), 1e-3 * scale) * 2.0 ** -7):
        return "ulp", int(a.size)
    return "wrong", int(a.size)


def pdsch():
    bad = n = skipped = ulp = 0
    for seed in range(6):
        rng = np.random.default_rng(BASE + 1000 + seed)
        for pdu, P, S in cases.random_pdus(o.tbs, rng, 80):
            if o.validate(pdu) != 0 or o.derive(pdu)["nof_re"] == 0:
                continue
            tb = cases.random_tb(rng, pdu)
            d = o.derive(pdu)
            impl = int(rng.integers(0, 3))
            fs = d["segment_length"] - 2 * d["lifting_size"] - d["nof_filler_bits"]
            if d["nof_filler_bits"] and fs < d["n_cb"] < fs + d["nof_filler_bits"]:
                skipped += 1   # the circular buffer ends inside the filler bits: the reference reads out of bounds
                continue
            want = o.pdsch_process(pdu, tb, P, S)
            n += 1
            verdict, count = grid_verdict(r.pdsch_process(pdu, tb, P, S, impl=impl), want)
            if verdict == "ulp":
                ulp += 1
                print("pdsch: seed %d processor %d: %d element(s) of %d one bf16 ulp apart" % (seed, impl, count, want.size), flush=True)
            elif verdict != "exact":
                bad += 1
                print("PDSCH MISMATCH seed", seed, "processor", impl, count, d, flush=True)
    print("pdsch: %d random PDUs oracle = reference (a random one of its three processors each), %d mismatches, %d with elements one "
          "ulp apart, %d skipped (reference undefined)" % (n, bad, ulp, skipped), flush=True)
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
        want = r.ldpc_decode(bg, zc, nf, crc, iters, scaling, llr, simd=0)
        got = o.ldpc_decode(bg, zc, nf, crc, iters, scaling, llr)
        if got[0] != want[0] or not np.array_equal(got[1], want[1]):
            bad += 1
            print("DECODER MISMATCH", bg, zc, nf, hex(crc), iters, scaling, nof_llr, flush=True)
    print("decoder: 200 random configurations oracle = reference (generic decoder), %d mismatches" % bad, flush=True)
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
        want = r.ldpc_rate_dematch(bg, zc, rv, qm, nref, nf, new_data, llr, old, simd=0)
        if not np.array_equal(o.ldpc_rate_dematch(bg, zc, rv, qm, nref, nf, new_data, llr, old), want):
            bad += 1
            print("DEMATCHER MISMATCH", bg, zc, e, rv, qm, nref, nf, new_data, flush=True)
    print("rate dematcher: 300 random configurations oracle = reference (generic), %d mismatches" % bad, flush=True)
    total += bad
    bad = 0
    for t in range(300):
        n = int(rng.choice([rng.integers(1, 200), rng.integers(1, 70000), rng.integers(65000, 66100), rng.integers(1, 1 << 21)]))
        c_init = int(rng.integers(0, 1 << 31))
        llr = rng.integers(-128, 128, n).astype(np.int8)
        if not np.array_equal(o.prg_apply_xor_llr(c_init, 0, llr), r.prg_apply_xor_llr(c_init, 0, llr)):
            bad += 1
            print("DESCRAMBLER MISMATCH", c_init, n, flush=True)
    print("soft-bit descrambler: 300 random (c_init, length) pairs oracle = reference, %d mismatches" % bad, flush=True)
    return total + bad


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
        if o.csi_rs_validate(cfg) != 0:
            bad += 1
            continue
        if not np.array_equal(o.csi_rs_map(cfg, grid), r.csi_rs_map(cfg, grid, simd=int(t % 2))):
            bad += 1
            print("CSI-RS MISMATCH", row, start, nrb, cp, flush=True)
    print("csi-rs: 200 random configurations oracle = reference, %d mismatches" % bad, flush=True)
    return bad


def dlctrl():
    rng = np.random.default_rng(BASE + 515151)
    bad = n = ulp = 0
    for t in range(1500):
        pdu = cases.random_pdcch(rng)
        grid = (rng.standard_normal((4, 14, 52 * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        n += 1
        verdict, count = grid_verdict(r.pdcch_process(pdu, grid, simd=int(t % 2)), o.pdcch_process(pdu, grid))
        if verdict == "ulp":
            ulp += 1
            print("pdcch: PDU %d (simd %d): %d element(s) one bf16 ulp apart" % (t, t % 2, count), flush=True)
        elif verdict != "exact":
            bad += 1
            print("PDCCH MISMATCH", t, count, flush=True)
    print("pdcch: %d random PDUs oracle = reference, %d mismatches, %d with elements one ulp apart" % (n, bad, ulp), flush=True)
    n2 = bad2 = 0
    for t in range(600):
        nrb = int(rng.integers(24, 107))
        ports = int(rng.integers(1, 5))
        pdu = cases.random_ssb(rng, nrb, ports)
        grid = (rng.standard_normal((ports, 14, nrb * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        if o.ssb_validate(pdu) != 0:
            continue
        n2 += 1
        if not np.array_equal(o.ssb_process(pdu, grid), r.ssb_process(pdu, grid)):
            bad2 += 1
            print("SSB MISMATCH", t, flush=True)
    print("ssb: %d random PDUs oracle = reference, %d mismatches" % (n2, bad2), flush=True)
    return bad + bad2


def demod():
    rng = np.random.default_rng(BASE + 626262)
    bad = n = 0
    for t in range(1200):
        modulation = int(rng.choice([0, 1, 2, 4, 6, 8]))
        length = int(rng.integers(1, 5000))
        sym, noise = cases.demod_inputs(rng, modulation, length, int(rng.integers(0, 3)))
        n += 1
        if not np.array_equal(o.demodulate_soft(modulation, sym, noise), r.demodulate_soft(modulation, sym, noise)):
            bad += 1
            print("DEMOD MISMATCH", modulation, length, flush=True)
    print("soft demodulator: %d random spans oracle = reference, %d mismatches" % (n, bad), flush=True)
    return bad


def lower():
    rng = np.random.default_rng(BASE + 737373)
    bad = n = 0
    for t in range(400):
        nsamp = int(rng.integers(1, 20000))
        x = ((rng.standard_normal(nsamp) + 1j * rng.standard_normal(nsamp)) * rng.uniform(0.05, 2.0)).astype(np.complex64)
        cfg = abi.AmplitudeCfg(int(rng.integers(0, 2)), int(rng.integers(0, 2)), float(rng.uniform(-12, 6)), float(rng.uniform(0.5, 2)),
                               float(rng.uniform(-12, -0.1)))
        oy, _ = o.amplitude_control(cfg, x)
        ry, _ = r.amplitude_control(cfg, x)
        scale = float(rng.choice([32767.0, 1000.0, 40000.0]))
        n += 1
        if not (np.array_equal(oy.view(np.uint32), ry.view(np.uint32)) and np.array_equal(o.iq_convert_ci16(oy, scale), r.iq_convert_ci16(ry, scale))):
            bad += 1
            print("LOWER-PHY MISMATCH amplitude/ci16", t, flush=True)
    for t in range(400):
        nprb = int(rng.integers(1, 274))
        prbs = (((rng.standard_normal((nprb, 12, 2)) * rng.uniform(0.01, 1.5)).astype(np.float32).view(np.uint32)) >> 16).astype(np.uint16)
        cfg = abi.OfhCompressionCfg(int(rng.integers(0, 2)), int(rng.integers(8, 17)), float(rng.uniform(0.2, 1.5)))
        n += 1
        if not np.array_equal(o.ofh_compress(cfg, prbs), r.ofh_compress(cfg, prbs, 1)):
            bad += 1
            print("OFH MISMATCH", cfg.type, cfg.data_width, nprb, flush=True)
    print("lower-PHY tail: %d random cases oracle = reference (vector compressor), %d mismatches" % (n, bad), flush=True)
    return bad


if __name__ == "__main__":
    legs = {"pdsch": pdsch, "rx": rx, "csi": csi, "dlctrl": dlctrl, "demod": demod, "lower": lower}
    which = sys.argv[1:] or list(legs)
    total = sum(legs[w]() for w in which)
    sys.exit(1 if total else 0)
