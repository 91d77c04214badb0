"""CPU tests (no GPU): pin the C oracle.

1. Against the golden vectors generated from the compiled reference (tests/golden/*.npz) -- always runs.
2. Against the compiled reference itself (oracle/_ref/libsrsref.so) on fresh random inputs -- runs wherever
   `make -C oracle ref` was possible (this container); skipped on machines without /root/reference.
3. Against known-answer restatements that need no vectors (bit-serial Gold generator, polynomial long division,
   TS 38.211 modulation formulae, numpy FFT), as the reference's own unit tests do
   (pseudo_random_generator_test.cpp:101-123, crc_calculator_generic_impl.cpp:59-85, dft_processor_test.cpp:43-91).
"""
import hashlib
import os

import numpy as np
import pytest

import backends
import cases

abi = backends.abi
lib = backends.pkg.lib

LIFTING_SIZES = [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 44, 48, 52,
                 56, 60, 64, 72, 80, 88, 96, 104, 112, 120, 128, 144, 160, 176, 192, 208, 224, 240, 256, 288, 320, 352,
                 384]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---------------------------------------------------------------------------------------------------------------------
# 1. golden vectors from the reference
# ---------------------------------------------------------------------------------------------------------------------
def test_oracle_ldpc_encoder_golden(oracle):
    g = np.load(os.path.join(cases.GOLDEN, "ldpc_encoder.npz"))
    assert len(g["keys"]) == 102
    for bg, zc in g["keys"]:
        out_bits = (66 if bg == 1 else 50) * int(zc)
        got = oracle.ldpc_encode(int(bg), int(zc), g["msg_%d_%d" % (bg, zc)], out_bits)
        assert np.array_equal(got, g["out_%d_%d" % (bg, zc)]), (bg, zc)


def test_oracle_pdsch_processor_golden(oracle):
    g = np.load(os.path.join(cases.GOLDEN, "pdsch_processor.npz"))
    items = [("cfg%d" % c,) + cases.baseline_config(c)[:3] for c in (1, 2, 3)]
    for i, pdu in enumerate(cases.unit_test_like_pdus(np.random.default_rng(2024))):
        items.append(("unit%02d" % i, pdu, 4, 26 * 12))
    for name, pdu, nof_ports, nof_subc in items:
        tb = np.random.default_rng(int(g[name + "_tb_seed"])).integers(0, 256, pdu.tb_size_bytes, dtype=np.uint8)
        d = oracle.derive(pdu)
        grid, rm, _ = oracle.pdsch_process(pdu, tb, nof_ports, nof_subc, taps=True, codeword_bits=d["codeword_bits"])
        assert sha(rm) == str(g[name + "_cw_sha"]), name
        assert np.array_equal(rm[:64], g[name + "_cw_head"]), name
        assert sha(grid) == str(g[name + "_grid_sha"]), name
        g32 = grid.view(np.uint32).reshape(nof_ports, -1)
        for p in range(nof_ports):
            assert np.array_equal(g32[p][g["%s_p%d_idx" % (name, p)]], g["%s_p%d_val" % (name, p)]), (name, p)


@pytest.mark.parametrize("name", ["n4096", "n2048", "n1024", "n1536", "n3072", "n768", "n384", "x512", "x2048", "n6144",
                                  "n4608"])
def test_oracle_ofdm_golden(oracle, name):
    g = np.load(os.path.join(cases.GOLDEN, "ofdm_sizes.npz" if name in ("n6144", "n4608") else "ofdm_modulator.npz"))
    mu, bw, n, fc, slot = g[name + "_cfg"][:5]
    ext = int(g[name + "_cfg"][5]) if len(g[name + "_cfg"]) > 5 else 0   # x...: extended cyclic prefix, 12 symbols
    cfg = abi.OfdmConfig(int(mu), int(bw), int(n), ext, 1.0 / np.sqrt(n), float(fc))
    iq = oracle.ofdm_slot(cfg, g[name + "_grid"], int(slot))
    want = g[name + "_iq"]
    # The reference's own tolerance is |err| / sqrt(N) < 5e-5 (ofdm_modulator_vectortest.cpp:30); ours: 1e-5 relative.
    assert np.abs(iq - want).max() / np.abs(want).max() < 1e-5


def bf16_to_f32(raw):
    return (raw.astype(np.uint32) << 16).view(np.float32)


def assert_bf16_grids_close(got, want, min_exact=0.99):
    """cbf16 grids that went through different float FFTs: at most one bf16 ulp apart, almost all identical."""
    a, b = bf16_to_f32(got), bf16_to_f32(want)
    scale = np.abs(b).max()
    assert np.all(np.abs(a - b) <= np.maximum(np.abs(b), 1e-3 * scale) * 2.0 ** -7)
    assert np.mean(got == want) >= min_exact


@pytest.mark.parametrize("case", [(1, 273, 4096, 3.5e9, 1, 0), (0, 106, 2048, 2.4e9, 0, 9), (0, 52, 1024, 2.4e9, 0, 0),
                                  (0, 106, 1536, 2.4e9, 0, 5), (1, 51, 768, 3.6e9, 1, 0), (0, 25, 512, 2.4e9, 0, 3),
                                  (2, 48, 1024, 3.5e9, 2, 11, 1), (2, 192, 4096, 28e9, 1, 0, 1)])
def test_oracle_ofdm_demodulator_vs_reference(oracle, ref, case):
    """ofdm_slot_demodulator_impl (compiled reference) against the C restatement on random IQ, and the
    modulate -> demodulate round trip of ofdm_modulator_unittest / ofdm_demodulator_unittest style."""
    if ref is None:
        pytest.skip("compiled reference not available")
    mu, bw, n, fc, slot, wo = case[:6]
    ext = case[6] if len(case) > 6 else 0   # extended cyclic prefix
    rng = np.random.default_rng(n + wo)
    cfg = abi.OfdmConfig(mu, bw, n, ext, 1.0 / np.sqrt(n), fc)
    size = lib.slot_size(cfg, slot)
    iq = (rng.standard_normal((2, size)) + 1j * rng.standard_normal((2, size))).astype(np.complex64)
    assert_bf16_grids_close(oracle.ofdm_demod_slot(cfg, iq, slot, wo), ref.ofdm_demod_slot(cfg, iq, slot, wo))
    # Round trip: a bf16 grid modulated and demodulated with scales whose product is 1 / N comes back (bf16 rounding).
    grid = (rng.standard_normal((1, 14, bw * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    if ext:
        grid[:, 12:] = 0   # a slot has 12 symbols: grid rows 12 and 13 are neither sent nor received
    back = oracle.ofdm_demod_slot(cfg, oracle.ofdm_slot(cfg, grid, slot), slot, 0)
    assert_bf16_grids_close(back, grid, min_exact=0.9)


@pytest.mark.parametrize("name", ["d4096", "d2048w", "d1536", "d512w", "dx1024w", "d6144", "d4608w"])
def test_oracle_ofdm_demodulator_golden(oracle, name):
    g = np.load(os.path.join(cases.GOLDEN, "ofdm_sizes.npz" if name in ("d6144", "d4608w") else "ofdm_demodulator.npz"))
    mu, bw, n, fc, slot, wo = g[name + "_cfg"][:6]
    ext = int(g[name + "_cfg"][6]) if len(g[name + "_cfg"]) > 6 else 0
    cfg = abi.OfdmConfig(int(mu), int(bw), int(n), ext, 1.0 / np.sqrt(n), float(fc))
    assert_bf16_grids_close(oracle.ofdm_demod_slot(cfg, g[name + "_iq"], int(slot), int(wo)), g[name + "_grid"])


@pytest.mark.parametrize("case", cases.LDPC_DECODE_CASES)
def test_oracle_ldpc_decoder_vs_reference(oracle, ref, case):
    """Layered scaled min-sum against the C restatement.  The oracle follows ldpc_decoder_generic: identical hard bits and
    iteration counts, with and without CRC early stop.  The reference's AVX2 decoder uses different intermediate
    arithmetic and differs from its own generic decoder at the bit level before convergence (and by an iteration now and
    then); it is only required to deliver the same message once both have passed the CRC."""
    bg, zc, extra, tail, crc_id, filler, amp, sigma = case
    rng = np.random.default_rng(zc * 1000 + extra)
    nof_llr = cases.ldpc_decode_nof_llr(case)
    msg, llr = cases.make_ldpc_llrs(oracle, rng, bg, zc, nof_llr, crc_id, filler, amp, sigma)
    for crc in (crc_id, 0):
        for iters in (1, 8):
            it_o, bits_o = oracle.ldpc_decode(bg, zc, filler, crc, iters, 0.8, llr)
            it_g, bits_g = ref.ldpc_decode(bg, zc, filler, crc, iters, 0.8, llr, simd=0)
            it_a, bits_a = ref.ldpc_decode(bg, zc, filler, crc, iters, 0.8, llr, simd=1)
            assert it_o == it_g and np.array_equal(bits_o, bits_g), (crc, iters)
            if it_o >= 1 and it_a >= 1:
                assert np.array_equal(bits_o, bits_a), (crc, iters)
    it, bits = oracle.ldpc_decode(bg, zc, filler, crc_id, 8, 0.8, llr)
    if sigma < amp:
        assert it >= 1 and np.array_equal(bits[: msg.size - filler], msg[: msg.size - filler])
    # all-zero input: nothing to decode (ldpc_decoder_impl.cpp:88-97)
    zero = np.zeros(nof_llr, np.int8)
    assert oracle.ldpc_decode(bg, zc, filler, crc_id, 8, 0.8, zero)[0] == 0
    assert ref.ldpc_decode(bg, zc, filler, crc_id, 8, 0.8, zero)[0] == 0


@pytest.mark.parametrize("case", cases.LDPC_DEMATCH_CASES)
def test_oracle_ldpc_rate_dematcher_vs_reference(oracle, ref, case):
    """Rate dematcher against the compiled reference: new data and HARQ combining, arbitrary int8 soft bits (infinities
    included) against the generic implementation bit for bit.  The reference's AVX2 implementation saturates an
    infinite operand like a finite one, so it is compared on finite inputs only."""
    bg, zc, e, rv, qm, nref, nf = case
    rng = np.random.default_rng(e * 7 + rv)
    n = (66 if bg == 1 else 50) * zc
    for finite in (False, True):
        lim = 120 if finite else 127
        llr = rng.integers(-lim, lim + 1, e).astype(np.int8)
        llr[rng.integers(0, e, e // 10)] = 0
        old = rng.integers(-lim, lim + 1, n).astype(np.int8)
        for new_data in (1, 0):
            want = ref.ldpc_rate_dematch(bg, zc, rv, qm, nref, nf, new_data, llr, old, simd=0)
            got = oracle.ldpc_rate_dematch(bg, zc, rv, qm, nref, nf, new_data, llr, old)
            assert np.array_equal(got, want), (finite, new_data, int(np.count_nonzero(got != want)))
            if finite:
                avx = ref.ldpc_rate_dematch(bg, zc, rv, qm, nref, nf, new_data, llr, old, simd=1)
                if new_data:
                    # the filler bits are the only infinities the AVX2 path meets: it may only differ where a wrap adds to them
                    nof_sys = ((22 if bg == 1 else 10) - 2) * zc
                    keep = np.ones(n, bool)
                    keep[nof_sys - nf:nof_sys] = False
                    assert np.array_equal(avx[keep], want[keep])
    # the reference's own test property (ldpc_rm_test.cpp:183-211): match -> LLR -> dematch -> hard -> match is the identity
    cb = rng.integers(0, 2, n, dtype=np.uint8)
    nof_sys = ((22 if bg == 1 else 10) - 2) * zc
    cb[nof_sys - nf:nof_sys] = 0
    matched = np.unpackbits(oracle.rate_match(bg, zc, rv, qm, nref, nf, np.packbits(cb), e))[:e]
    soft = oracle.ldpc_rate_dematch(bg, zc, rv, qm, nref, nf, 1, (1 - 2 * matched.astype(np.int8)), np.zeros(n, np.int8))
    hard = (soft < 0).astype(np.uint8)
    again = np.unpackbits(oracle.rate_match(bg, zc, rv, qm, nref, nf, np.packbits(hard), e))[:e]
    assert np.array_equal(again, matched)
    assert np.all(soft[nof_sys - nf:nof_sys] == 127)


def test_oracle_vs_ref_diagonal_precoding_variants(oracle, ref):
    """Weight matrices full of exact zeros (signed-zero products in the layer sum), oracle against the compiled reference
    (its three processors, AVX2 and generic precoders): bit-exact grids."""
    rng = np.random.default_rng(607)
    for name, w in cases.diagonal_precoding_variants():
        layers = w.shape[2]
        tb_bits = oracle.tbs(12, 12, 0, 8, 700.0, layers, 30)
        pdu = abi.make_pdu(bwp_size_rb=30, qm=8, dmrs_symbols=(2, 11), prb_start=0, prb_count=30, start_symbol=1,
                           nof_symbols=13, precoding=w, tb_size_bytes=tb_bits // 8, ratio_data_dB=1.5,
                           nof_cdm_groups_without_data=2, scrambling_id=77, n_id=5, rnti=4321)
        tb = cases.random_tb(rng, pdu)
        want = oracle.pdsch_process(pdu, tb, w.shape[1], 30 * 12)
        for impl in range(3):
            got = ref.pdsch_process(pdu, tb, w.shape[1], 30 * 12, impl=impl)
            assert np.array_equal(got, want), (name, impl)


@pytest.mark.parametrize("bg", [1, 2])
def test_oracle_ldpc_decoder_reference_unit_tests(oracle, ref, bg):
    """The reference's LDPCDecTest / LDPCDecTestZeroLLR / LDPCDecTestAlmostZeroLLR (ldpc_enc_dec_test.cpp:287-358) on
    the restatement: every lifting size, noiseless codeblocks at the test's four lengths decode to the message in the
    default 6 iterations without a CRC; all-zero and almost-zero inputs give all ones.  Every fourth lifting size is
    also compared bit for bit with the compiled reference's generic decoder."""
    rng = np.random.default_rng(300 + bg)
    kb = 22 if bg == 1 else 10
    for n, zc in enumerate(cases.LIFTING_SIZES):
        for length in cases.ldpc_dec_test_lengths(bg, zc):
            msg, llr = cases.noiseless_llrs(oracle, rng, bg, zc, zc // 3, length)
            it, bits = oracle.ldpc_decode(bg, zc, zc // 3, 0, 6, 0.8, llr)
            assert it == 0 and np.array_equal(bits, msg), (zc, length)
            if n % 4 == 0:
                assert np.array_equal(ref.ldpc_decode(bg, zc, zc // 3, 0, 6, 0.8, llr, simd=0)[1], bits)
        full = (66 if bg == 1 else 50) * zc
        zero = np.zeros(full, np.int8)
        it, bits = oracle.ldpc_decode(bg, zc, 0, 0, 6, 0.8, zero)
        assert it == 0 and bits.all()
        lo = (24 if bg == 1 else 12) * zc
        for i in range(lo + 2, full, 3):
            zero[i] = 1 if i % 2 == 0 else -1
        assert oracle.ldpc_decode(bg, zc, 0, 0, 6, 0.8, zero)[1].all()
        if n % 4 == 0:
            assert ref.ldpc_decode(bg, zc, 0, 0, 6, 0.8, zero, simd=0)[1].all()
    assert kb * zc == bits.size


def test_oracle_vs_ref_extended_cyclic_prefix_pdsch(oracle, ref):
    """PDSCH with extended cyclic prefix (12 symbols per slot: DM-RS c_init, symbol bounds): validator and grid against
    the reference's three processors; a DM-RS symbol beyond the slot is refused by both."""
    rng = np.random.default_rng(1212)
    for pdu, nof_ports, nof_subc in cases.extended_cp_pdus(oracle.tbs):
        assert oracle.validate(pdu) == 0 and ref.validate(pdu) == 0
        tb = cases.random_tb(rng, pdu)
        want = oracle.pdsch_process(pdu, tb, nof_ports, nof_subc)
        assert not want[:, 12:].any(), "symbols 12 and 13 do not exist with extended cyclic prefix"
        for impl in range(3):
            assert np.array_equal(ref.pdsch_process(pdu, tb, nof_ports, nof_subc, impl=impl), want), impl
    bad = cases.extended_cp_pdus(oracle.tbs)[0][0]
    bad.dmrs_symbol_mask |= 1 << 12
    bad.nof_symbols = 13
    assert oracle.validate(bad) != 0 and ref.validate(bad) != 0


@pytest.mark.parametrize("shape", ["cfg2", "cfg1", "bg2_multi"])
@pytest.mark.parametrize("early_stop", [1, 0])
def test_pusch_decoder_restatement_vs_reference(oracle, ref, shape, early_stop):
    """The transport-block decoder the GPU tests compare with (cases.pusch_decode_expected: pusch_decoder_impl restated on
    the oracle's codeblock functions) against the compiled reference's pusch_decoder_impl with its rx_buffer_pool, generic
    rate dematcher and generic LDPC decoder: two HARQ transmissions (rv 0 new data, rv 2 combined), same CRC verdict,
    number of decoder runs, iteration sum and maximum, and transport block."""
    rng = np.random.default_rng({"cfg2": 21, "cfg1": 22, "bg2_multi": 23}[shape] + early_stop)
    pdu, nof_ports, nof_subc, amp, sigma = cases.pusch_decoder_shape(oracle, shape)
    d = oracle.derive(pdu)
    C, n, G = d["nof_codeblocks"], d["full_length"], d["codeword_bits"]
    tb = cases.random_tb(rng, pdu)
    soft = np.zeros((C, n), np.int8)
    cb_ok = np.ones(C, np.uint8)
    cb_msg = np.zeros((C, d["segment_length"]), np.uint8)
    harq_id = {"cfg2": 1, "cfg1": 2, "bg2_multi": 3}[shape] + 4 * early_stop
    for tx, rv in enumerate((0, 2)):
        cfg = abi.PuschDecoderCfg(pdu.ldpc_base_graph, pdu.qm, rv, pdu.nof_layers, d["n_ref"], pdu.tb_size_bytes, G // pdu.qm, 6,
                                  early_stop, 1 if tx == 0 else 0)
        pdu.rv = rv
        _, rm, _ = oracle.pdsch_process(pdu, tb, nof_ports, nof_subc, taps=True, codeword_bits=G)
        pdu.rv = 0
        bits = np.unpackbits(rm)[:G].astype(np.float64)
        llr = np.clip(np.rint((1 - 2 * bits) * amp + rng.normal(0, sigma, G)), -120, 120).astype(np.int8)
        skipped = int(cb_ok.sum()) if tx else 0
        tb_ok, n_ok, it_sum, it_max, got_tb = cases.pusch_decode_expected(oracle, d, cfg, llr, soft, cb_ok, cb_msg)
        r_ok, r_runs, r_sum, r_max, r_tb = ref.pusch_decode(cfg, harq_id, C, llr)
        assert (r_ok, r_runs, r_sum, r_max) == (tb_ok, C - skipped, it_sum, it_max), (tx, (r_ok, r_runs, r_sum, r_max))
        if tb_ok:
            assert np.array_equal(r_tb, got_tb) and np.array_equal(got_tb, tb)
    assert tb_ok


def test_oracle_nzp_csi_rs_vs_reference(oracle, ref):
    """NZP-CSI-RS generator (rows 1-5) against the compiled reference's nzp_csi_rs_generator_impl (AVX2 and generic
    precoders): the signal is mapped into a grid full of other data; every grid word is compared, so both what is written
    (including the zeros a CDM group writes on the ports it does not use) and what is left alone are checked."""
    rng = np.random.default_rng(7415)
    for name, cfg, nof_ports, nof_subc in cases.csi_rs_cases(rng):
        assert oracle.csi_rs_validate(cfg) == 0, name
        grid = (rng.standard_normal((nof_ports, 14, nof_subc, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        got = oracle.csi_rs_map(cfg, grid)
        for simd in (1, 0):
            want = ref.csi_rs_map(cfg, grid, simd=simd)
            assert np.array_equal(got, want), (name, simd, int(np.count_nonzero(got != want)))
        assert np.count_nonzero(got != grid) > 0 or cfg.nof_rb == 1, name   # one PRB at density 0.5 may hold nothing


def test_oracle_vs_ref_random_pdus(oracle, ref):
    """Fuzz: 40 PDUs drawn at random within what the validator accepts (cases.random_pdus), oracle against the compiled
    reference's processors -- the same draw the GPU suite runs against the oracle."""
    rng = np.random.default_rng(20240611)
    for i, (pdu, nof_ports, nof_subc) in enumerate(cases.random_pdus(oracle.tbs, rng, 40)):
        assert oracle.validate(pdu) == ref.validate(pdu), i
        if oracle.validate(pdu) != 0:
            continue
        tb = cases.random_tb(rng, pdu)
        want = oracle.pdsch_process(pdu, tb, nof_ports, nof_subc)
        for impl in (0, 1):
            got = ref.pdsch_process(pdu, tb, nof_ports, nof_subc, impl=impl)
            assert np.array_equal(got, want), (i, impl, int(np.count_nonzero(got != want)), oracle.derive(pdu))


def test_rate_matcher_buffer_ending_inside_the_filler_bits(oracle):
    """Where the reference is undefined -- a limited circular buffer that ends inside the filler range; its select_bits jumps
    beyond the buffer (ldpc_rate_matcher_impl.cpp:115-137) and the compiled reference crashes, so it is not called here -- the
    oracle follows TS 38.212 Section 5.4.2.1 literally (found by profiles/fuzz_sweep.py with NRPHY_FUZZ_SEED=1000000)."""
    rng = np.random.default_rng(3)
    for case in cases.RM_CORNER_CASES:
        bg, rv, qm, nref, tb_bytes, nsym = case
        tb = rng.integers(0, 256, tb_bytes, dtype=np.uint8)
        want = cases.rm_corner_expected(oracle, case, tb)
        got = np.unpackbits(oracle.pdsch_encode_cfg(bg, rv, qm, nref, 1, nsym, tb))[: want.size]
        assert np.array_equal(got, want), case


def test_baseline_config_derived_values(oracle):
    """The derived sizes SURVEY.md section 8d lists for the BASELINE configs."""
    d = oracle.derive(cases.baseline_config(3)[0])
    assert (d["nof_codeblocks"], d["lifting_size"], d["nof_filler_bits"], d["nof_re"]) == (104, 384, 72, 29160)
    assert (d["rm_length_short"], d["rm_length_long"], d["nof_short_segments"], d["n_cb"]) == (8960, 8992, 64, 18432)
    assert cases.baseline_config(3)[0].tb_size_bytes * 8 == 868584
    d = oracle.derive(cases.baseline_config(2)[0])
    assert (d["nof_codeblocks"], d["lifting_size"], d["nof_filler_bits"], d["nof_re"]) == (14, 384, 80, 11448)
    assert cases.baseline_config(2)[0].tb_size_bytes * 8 == 116792
    d = oracle.derive(cases.baseline_config(1)[0])
    assert (d["nof_codeblocks"], d["lifting_size"], d["nof_filler_bits"], d["rm_length_short"]) == (1, 144, 104, 11232)
    assert cases.baseline_config(1)[0].tb_size_bytes * 8 == 1320
    sizes = [(p.tb_size_bytes * 8, oracle.derive(p)["nof_codeblocks"]) for p in cases.mixed_cell(0)[0]]
    assert sizes == [(6920, 2), (75792, 9), (151608, 18), (217128, 26)]


# ---------------------------------------------------------------------------------------------------------------------
# 2. the compiled reference on fresh inputs
# ---------------------------------------------------------------------------------------------------------------------
def test_oracle_vs_ref_scalars_and_crc(oracle, ref):
    rng = np.random.default_rng(1)
    for args in [(12, 36, 0, 8, 948, 4, 270), (12, 36, 0, 2, 120, 1, 52), (12, 36, 0, 6, 873, 2, 106), (14, 12, 0, 4, 378, 3, 17),
                 (2, 6, 6, 2, 30, 1, 1), (13, 24, 12, 6, 666, 2, 273), (7, 12, 18, 8, 711.5, 4, 100)]:
        assert oracle.tbs(*args) == ref.tbs(*args), args
    for n in (1, 3, 100, 1044, 108573):
        data = rng.integers(0, 256, n, dtype=np.uint8)
        for poly in (16, 0x24A, 0x24B):
            assert oracle.crc(poly, data) == ref.crc(poly, data)


def test_oracle_vs_ref_ldpc_encoder_all_graphs(oracle, ref):
    rng = np.random.default_rng(2)
    for bg in (1, 2):
        kb, nshort = (22, 66) if bg == 1 else (10, 50)
        for zc in LIFTING_SIZES:
            msg = np.packbits(rng.integers(0, 2, kb * zc, dtype=np.uint8))
            for out_bits in {nshort * zc, (kb + 2) * zc, nshort * zc - 3 * zc - 1, kb * zc + 2 * zc + 5}:
                assert np.array_equal(oracle.ldpc_encode(bg, zc, msg, out_bits), ref.ldpc_encode(bg, zc, msg, out_bits, 1))


def test_oracle_vs_ref_segmenter_and_rate_matcher(oracle, ref):
    rng = np.random.default_rng(3)
    for bg, nbytes, qm, layers, nre in ((1, 108573, 8, 4, 29160), (2, 165, 2, 1, 5616), (1, 14599, 6, 2, 11448),
                                        (2, 20, 2, 1, 300), (1, 1055, 4, 1, 3000), (2, 479, 4, 3, 999), (1, 3000, 6, 4, 4001),
                                        (2, 3, 2, 1, 100), (1, 4000, 8, 2, 777)):
        tb = rng.integers(0, 256, nbytes, dtype=np.uint8)
        sa, ma, za = oracle.segment(bg, 0, qm, 25344, layers, nre * layers, tb)
        sb, mb, zb = ref.segment(bg, 0, qm, 25344, layers, nre * layers, tb)
        assert za == zb and np.array_equal(ma, mb)
        k = (22 if bg == 1 else 10) * za
        assert np.array_equal(np.unpackbits(sa, axis=1)[:, :k], np.unpackbits(sb, axis=1)[:, :k])
    for bg, zc, rv, qm, nref, nf, e in ((1, 384, 0, 8, 18432, 72, 8992), (2, 144, 0, 2, 0, 104, 11232), (1, 384, 2, 6, 0, 80, 9804),
                                        (2, 352, 3, 4, 8000, 24, 5000), (1, 16, 1, 2, 0, 5, 2000), (1, 384, 1, 8, 365, 72, 8992),
                                        (2, 7, 3, 2, 0, 30, 2304)):
        cb = np.packbits(rng.integers(0, 2, (66 if bg == 1 else 50) * zc, dtype=np.uint8))
        assert np.array_equal(oracle.rate_match(bg, zc, rv, qm, nref, nf, cb, e), ref.rate_match(bg, zc, rv, qm, nref, nf, cb, e))


def test_oracle_vs_ref_prg_and_modulation(oracle, ref):
    rng = np.random.default_rng(4)
    for c_init, off, n in ((1 << 15, 0, 1000), (0x12345678, 777, 933120), (5, 100000, 336), (0x7FFFFFFF, 1, 64)):
        d = rng.integers(0, 256, (n + 7) // 8, dtype=np.uint8)
        assert np.array_equal(oracle.prg_xor(c_init, off, d, n)[: n // 8], ref.prg_xor(c_init, off, d, n)[: n // 8])
        assert np.array_equal(oracle.prg_float(c_init, off, 0.7, 333), ref.prg_float(c_init, off, 0.7, 333))
    for qm in (2, 4, 6, 8):
        bits = rng.integers(0, 256, qm * 100, dtype=np.uint8)
        a, sa = oracle.modulate(qm, bits, 800)
        b, sb = ref.modulate(qm, bits, 800)
        assert np.array_equal(a, b) and sa == sb


def test_oracle_vs_ref_llr_descrambling(oracle, ref):
    """Soft-bit apply_xor: every int8 value (including -128), lengths around the reference's 16-wide and step sizes."""
    rng = np.random.default_rng(44)
    every = np.arange(-128, 128, dtype=np.int8)
    for c_init, off, n in ((1 << 15, 0, 256), (0x12345678, 777, 100003), (5, 100000, 15), (0x7FFFFFFF, 1, 17),
                           ((0x4601 << 15) + 935, 0, 52416), (77, 3, 1), (78, 31, 33), (79, 0, 6 * 32 + 5)):
        llr = rng.integers(-128, 128, n).astype(np.int8)
        llr[: min(n, 256)] = np.tile(every, 2)[: min(n, 256)][rng.permutation(min(n, 256))] if n >= 256 else llr[: min(n, 256)]
        want = ref.prg_apply_xor_llr(c_init, off, llr)
        got = oracle.prg_apply_xor_llr(c_init, off, llr)
        assert np.array_equal(got, want), (c_init, off, n)
        assert np.array_equal(np.abs(got.astype(np.int16)), np.abs(llr.astype(np.int16)))
        assert np.array_equal(oracle.prg_apply_xor_llr(c_init, off, got), llr)  # an involution


def test_oracle_vs_ref_pdsch_processor(oracle, ref):
    """Grid bit-exact against the reference's generic, AVX2 and "lite" processors, codeword against pdsch_encoder."""
    import test_gpu_parity
    rng = np.random.default_rng(5)
    items = [(c, ) + cases.baseline_config(c)[:3] for c in (1, 2, 3)]
    items += [(i, pdu, 4, 26 * 12) for i, pdu in enumerate(cases.unit_test_like_pdus(rng))]
    items += [(n, p, a, b) for n, p, a, b in test_gpu_parity.edge_case_pdus() if p.nof_prg == 1]
    for name, pdu, nof_ports, nof_subc in items:
        assert oracle.validate(pdu) == ref.validate(pdu) == 0
        tb = cases.random_tb(rng, pdu)
        d = oracle.derive(pdu)
        grid, rm, _ = oracle.pdsch_process(pdu, tb, nof_ports, nof_subc, taps=True, codeword_bits=d["codeword_bits"])
        nb = d["codeword_bits"] // 8
        assert np.array_equal(rm[:nb], ref.pdsch_encode(pdu, tb, d)[:nb]), name
        for simd, impl in ((1, 0), (0, 0), (1, 1)):
            assert np.array_equal(grid, ref.pdsch_process(pdu, tb, nof_ports, nof_subc, simd=simd, impl=impl)), (name, simd, impl)


def test_oracle_vs_ref_validator(oracle, ref):
    def variants():
        base = dict(bwp_start_rb=1, bwp_size_rb=25, qm=4, dmrs_symbols=(2, 7), prb_start=3, prb_count=10, start_symbol=2,
                    nof_symbols=10, tb_size_bytes=100, precoding=abi.identity_precoding(2))
        yield dict(base)
        yield dict(base, dmrs_symbols=(1,))                       # DM-RS before the allocation
        yield dict(base, dmrs_symbols=(12,))                      # DM-RS after the allocation
        yield dict(base, dmrs_type=2)
        yield dict(base, start_symbol=6, nof_symbols=10, dmrs_symbols=(7,))  # beyond the slot
        yield dict(base, tbs_lbrm_bytes=0)
        yield dict(base, prb_start=20, prb_count=10)              # outside the BWP
        yield dict(base, nof_cdm_groups_without_data=3)
        yield dict(base, reserved=[(range(0, 26), [1] * 12, [0, 0, 1] + [0] * 11)])   # collides with DM-RS symbol 2
        yield dict(base, reserved=[(range(0, 26), [1] * 12, [0, 0, 0, 1] + [0] * 10)])
    for kw in variants():
        pdu = abi.make_pdu(**kw)
        assert (oracle.validate(pdu) == 0) == (ref.validate(pdu) == 0), kw
        assert (backends.pkg.lib.validate(pdu) == 0) == (ref.validate(pdu) == 0), kw


def test_oracle_vs_ref_dft_and_ofdm(oracle, ref):
    rng = np.random.default_rng(6)
    # every size of dft_processor_generic_impl.cpp:190-208
    for n in (128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 4608, 6144, 9216, 12288, 18432, 24576, 36864, 49152):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        for inv in (0, 1):
            a, b = oracle.dft(x, inv), ref.dft(x, inv)
            assert np.abs(a - b).max() / np.abs(b).max() < 4e-6, n
    # the last five: the extended-cyclic-prefix rows of the reference's ofdm_modulator_test_data.h (numerology 2)
    for mu, bw, n, fc, slot, ext in ((1, 273, 4096, 3.5e9, 0, 0), (1, 273, 4096, 3.5e9, 1, 0), (0, 52, 1024, 2.4e9, 0, 0),
                                     (0, 106, 2048, 0.0, 0, 0), (2, 24, 512, 28e9, 3, 0), (2, 12, 256, 3.5e9, 0, 1),
                                     (2, 24, 512, 3.5e9, 1, 1), (2, 48, 1024, 3.5e9, 2, 1), (2, 96, 2048, 28e9, 3, 1),
                                     (2, 192, 4096, 28e9, 0, 1), (0, 273, 6144, 3.5e9, 0, 0), (0, 216, 4608, 2.4e9, 0, 0)):
        cfg = abi.OfdmConfig(mu, bw, n, ext, 0.37, fc)
        grid = (rng.standard_normal((2, 14, bw * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        a, b = oracle.ofdm_slot(cfg, grid, slot), ref.ofdm_slot(cfg, grid, slot)
        assert a.shape == b.shape
        assert np.abs(a - b).max() / np.abs(b).max() < 2e-6


# ---------------------------------------------------------------------------------------------------------------------
# 2b. downlink control side: polar coding, PDCCH, SS/PBCH block
# ---------------------------------------------------------------------------------------------------------------------
def test_oracle_vs_ref_polar_code(oracle, ref):
    """polar_code::set: code length and information set for every downlink (K, E) of the PDCCH lengths and PBCH, plus a
    sweep over rate-matched lengths that reaches every branch (repetition, shortening, both puncturing bounds)."""
    for K in range(36, 165):
        for E in (108, 216, 432, 864, 1728):
            if K < E:
                a, b = oracle.polar_code(K, E), ref.polar_code(K, E)
                assert a[0] == b[0] and np.array_equal(a[1], b[1]), (K, E)
    rng = np.random.default_rng(5)
    for _ in range(400):
        K = int(rng.integers(36, 165))
        E = int(rng.integers(K + 1, 2000))
        a, b = oracle.polar_code(K, E), ref.polar_code(K, E)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]), (K, E)


def test_oracle_vs_ref_pdcch_encoder(oracle, ref):
    """pdcch_encoder_impl (CRC24C + RNTI mask, polar interleaver / allocator / encoder / rate matcher) on random DCIs."""
    rng = np.random.default_rng(1)
    for _ in range(300):
        E = int(rng.choice([108, 216, 432, 864, 1728]))
        A = int(rng.integers(12, min(129, E - 24)))
        payload, rnti = rng.integers(0, 2, A, dtype=np.uint8), int(rng.integers(0, 65536))
        assert np.array_equal(oracle.pdcch_encode(payload, rnti, E), ref.pdcch_encode(payload, rnti, E)), (A, E)


def test_oracle_vs_ref_pdcch_processor(oracle, ref):
    """pdcch_processor_impl with both precoders of the reference on random PDUs: the three CCE-to-REG mappings, 1-3
    symbols, every aggregation level, 1-4 ports, wideband and per-PRG complex weights, power offsets -- into grids
    full of other data, so that what is left alone is checked too."""
    rng = np.random.default_rng(3)
    for i in range(150):
        pdu = cases.random_pdcch(rng)
        assert oracle.pdcch_validate(pdu) == 0
        grid = (rng.standard_normal((4, 14, 52 * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        want = oracle.pdcch_process(pdu, grid)
        assert np.array_equal(want, ref.pdcch_process(pdu, grid, simd=1)), i
        assert np.array_equal(want, ref.pdcch_process(pdu, grid, simd=0)), i


def test_oracle_vs_ref_ssb_processor(oracle, ref):
    """pbch_encoder_impl and ssb_processor_impl: pattern cases A-C (L_max 4 / 8, both half frames) and case D with
    L_max = 64 (block index bits in the payload), several ports."""
    rng = np.random.default_rng(2)
    for _ in range(120):
        pdu = cases.random_ssb(rng, nof_ports=3)
        assert oracle.ssb_validate(pdu) == 0
        assert np.array_equal(oracle.pbch_encode(pdu), ref.pbch_encode(pdu))
        grid = (rng.standard_normal((3, 14, 52 * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        assert np.array_equal(oracle.ssb_process(pdu, grid), ref.ssb_process(pdu, grid))
    groups = (0, 1, 2, 3, 5, 6, 7, 8, 10, 11, 12, 13, 15, 16, 17, 18)
    for _ in range(30):
        idx = int(rng.integers(0, 64))
        first = (4, 8, 16, 20)[idx % 4] + 28 * groups[idx // 4]
        pdu = abi.make_ssb(pattern_case="D", ssb_idx=idx, L_max=64, phys_cell_id=int(rng.integers(0, 1008)),
                           payload=rng.integers(0, 2, 32, dtype=np.uint8), sfn=int(rng.integers(0, 1024)), numerology=3,
                           slot_index=first // 14, common_scs=3, subcarrier_offset=int(rng.integers(0, 12)),
                           offset_to_pointA=2 * int(rng.integers(0, 8)), ports=(1,))
        assert oracle.ssb_validate(pdu) == 0
        assert np.array_equal(oracle.pbch_encode(pdu), ref.pbch_encode(pdu))
        grid = np.zeros((2, 14, 52 * 12, 2), np.uint16)
        assert np.array_equal(oracle.ssb_process(pdu, grid), ref.ssb_process(pdu, grid))


# ---------------------------------------------------------------------------------------------------------------------
# 2c. lower-PHY tail: amplitude controller, cf32 -> ci16, Open Fronthaul compression
# ---------------------------------------------------------------------------------------------------------------------
def test_oracle_vs_ref_amplitude_controller_and_ci16(oracle, ref):
    """amplitude_controller_{clipping,scaling}_impl::process (samples bit for bit, measurements to float accuracy, clip
    counts exactly) and srsvec::convert to int16 (vector lanes round to nearest even and saturate, the tail of a call
    rounds half away: exact ties included)."""
    rng = np.random.default_rng(4)
    for t in range(60):
        n = int(rng.integers(1, 3000))
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * float(rng.choice([0.1, 0.5, 1.0]))
        if t == 0:
            x[:] = 0   # zero power: the reference reports PAPR 1 and leaves the counters alone
        cfg = abi.AmplitudeCfg(int(rng.integers(0, 2)), int(rng.integers(0, 2)), float(rng.choice([0.0, -3.0, 2.5])),
                               float(rng.choice([1.0, 2.0])), float(rng.choice([-0.1, -6.0, -12.0])))
        ya, ma = oracle.amplitude_control(cfg, x)
        yb, mb = ref.amplitude_control(cfg, x)
        assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))
        for k in ("avg_power_fs", "peak_power_fs", "papr_lin", "gain_dB"):
            assert abs(ma[k] - mb[k]) <= 2e-5 * max(1e-9, abs(mb[k])), (k, ma[k], mb[k])
        assert (ma["nof_clipped"], ma["nof_processed"]) == (mb["nof_clipped"], mb["nof_processed"])
        scale = float(rng.choice([32767.0, 1000.0, 20000.0]))
        assert np.array_equal(oracle.iq_convert_ci16(x, scale), ref.iq_convert_ci16(x, scale))
    x = np.zeros(37, np.complex64)
    x.real, x.imag = np.arange(37) + 0.5, -(np.arange(37) + 0.5)
    for k in (1.0, 2000.0):   # ties; saturation in the vector lanes
        assert np.array_equal(oracle.iq_convert_ci16(x * k, 1.0), ref.iq_convert_ci16(x * k, 1.0))


def test_oracle_vs_ref_ofh_compression(oracle, ref):
    """iq_compression_{none,bfp}: the reference's AVX2 compressors (what its factory picks on this host) for every
    supported width, 1-59 PRBs per call, all-zero PRBs, exact ties; and its generic compressors wherever the two agree
    by construction (BFP: always)."""
    rng = np.random.default_rng(4)
    for t in range(300):
        typ, w, nprb = int(rng.integers(0, 2)), int(rng.integers(8, 17)), int(rng.integers(1, 60))
        x = (rng.standard_normal((nprb, 12, 2)) * float(rng.choice([0.01, 0.2, 0.33]))).astype(np.float32)
        prbs = (x.view(np.uint32) >> 16).astype(np.uint16)
        if t % 7 == 0:
            prbs[0] = 0
        cfg = abi.OfhCompressionCfg(typ, w, float(rng.choice([1.0, 0.5, 0.9])))
        got = oracle.ofh_compress(cfg, prbs)
        assert np.array_equal(got, ref.ofh_compress(cfg, prbs, 1)), (typ, w, nprb)
        if typ == 1:
            assert np.array_equal(got, ref.ofh_compress(cfg, prbs, 0)), (typ, w, nprb)
    for w, typ in ((9, 0), (16, 0), (12, 0), (9, 1), (14, 1)):
        gain = ((1 << (w - 1)) - 1) if typ == 0 else 32767
        vals = (np.arange(24 * 5) % 40 - 20 + 0.5).astype(np.float32)   # half-integers, exact in bf16
        prbs = (vals.view(np.uint32) >> 16).astype(np.uint16).reshape(5, 12, 2)
        cfg = abi.OfhCompressionCfg(typ, w, 1.0 / gain)
        assert np.array_equal(oracle.ofh_compress(cfg, prbs), ref.ofh_compress(cfg, prbs, 1)), (w, typ)


def dl_control_golden(kind, i):
    """(PDU, packed encoder output, expected grid) of entry i of tests/golden/dl_control.npz."""
    g = np.load(os.path.join(cases.GOLDEN, "dl_control.npz"))
    key = "%s%02d_" % (kind, i)
    pdu = cases.pdu_from_arrays(abi.PdcchPdu if kind == "pdcch" else abi.SsbPdu, g, key)
    nof_ports = 4 if kind == "pdcch" else 3
    grid = np.zeros(nof_ports * 14 * 52 * 12, np.uint32)
    grid[g[key + "idx"]] = g[key + "val"]
    return pdu, g[key + "encoded"], grid.view(np.uint16).reshape(nof_ports, 14, 52 * 12, 2)


def test_oracle_dl_control_golden(oracle):
    """The oracle against the reference's outputs stored in tests/golden/dl_control.npz (runs on the GPU box too)."""
    g = np.load(os.path.join(cases.GOLDEN, "dl_control.npz"))
    for i in range(int(g["n_pdcch"])):
        pdu, enc, grid = dl_control_golden("pdcch", i)
        payload = np.array(list(pdu.payload)[: pdu.payload_size], np.uint8)
        assert np.array_equal(np.packbits(oracle.pdcch_encode(payload, pdu.rnti, 108 * pdu.aggregation_level)), enc), i
        assert np.array_equal(oracle.pdcch_process(pdu, np.zeros_like(grid)), grid), i
    for i in range(int(g["n_ssb"])):
        pdu, enc, grid = dl_control_golden("ssb", i)
        assert np.array_equal(np.packbits(oracle.pbch_encode(pdu)), enc), i
        assert np.array_equal(oracle.ssb_process(pdu, np.zeros_like(grid)), grid), i


# ---------------------------------------------------------------------------------------------------------------------
# 3. known-answer restatements
# ---------------------------------------------------------------------------------------------------------------------
def gold_bits(c_init, n):
    x1 = [1] + [0] * 30
    x2 = [(c_init >> i) & 1 for i in range(31)]
    for i in range(1600 + n):
        x1.append(x1[i + 3] ^ x1[i])
        x2.append(x2[i + 3] ^ x2[i + 2] ^ x2[i + 1] ^ x2[i])
    return np.array([x1[i + 1600] ^ x2[i + 1600] for i in range(n)], np.uint8)


def test_oracle_gold_sequence_known_answer(oracle):
    for c_init in (0, 1, 0x5A5A5A5, 0x7FFFFFFF):
        n = 2000
        want = gold_bits(c_init, n + 37)
        got = np.unpackbits(oracle.prg_xor(c_init, 37, np.zeros(n // 8, np.uint8), n))
        assert np.array_equal(got, want[37:])


def test_oracle_crc_long_division(oracle):
    rng = np.random.default_rng(8)
    for poly_id, poly, order in ((16, 0x11021, 16), (0x24A, 0x1864CFB, 24), (0x24B, 0x1800063, 24)):
        data = rng.integers(0, 256, 57, dtype=np.uint8)
        reg = 0
        for b in list(np.unpackbits(data)) + [0] * order:
            reg = (reg << 1) | int(b)
            if reg >> order:
                reg ^= poly
        assert oracle.crc(poly_id, data) == reg


def test_oracle_modulation_formulae(oracle):
    """TS 38.211 Section 5.1.3-5.1.6 closed forms."""
    for qm in (2, 4, 6, 8):
        nsym = 1 << qm
        bits = np.array([[(i >> (qm - 1 - j)) & 1 for j in range(qm)] for i in range(nsym)], np.uint8)
        got, scale = oracle.modulate(qm, np.packbits(bits.reshape(-1)), nsym)
        b = 1 - 2 * bits.astype(np.int32)
        if qm == 2:
            re, im = b[:, 0], b[:, 1]
        elif qm == 4:
            re, im = b[:, 0] * (2 - b[:, 2]), b[:, 1] * (2 - b[:, 3])
        elif qm == 6:
            re, im = b[:, 0] * (4 - b[:, 2] * (2 - b[:, 4])), b[:, 1] * (4 - b[:, 3] * (2 - b[:, 5]))
        else:
            re = b[:, 0] * (8 - b[:, 2] * (4 - b[:, 4] * (2 - b[:, 6])))
            im = b[:, 1] * (8 - b[:, 3] * (4 - b[:, 5] * (2 - b[:, 7])))
        assert np.array_equal(got[:, 0], re) and np.array_equal(got[:, 1], im)
        assert abs(scale - 1 / np.sqrt({2: 2, 4: 10, 6: 42, 8: 170}[qm])) < 1e-7


def test_oracle_dft_vs_numpy(oracle):
    rng = np.random.default_rng(9)
    for n in (128, 384, 1024, 4096):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        assert np.abs(oracle.dft(x, 0) - np.fft.fft(x.astype(np.complex128))).max() < 1e-5 * np.sqrt(n) * 4
        assert np.abs(oracle.dft(x, 1) - np.fft.ifft(x.astype(np.complex128)) * n).max() < 1e-5 * np.sqrt(n) * 4


def test_oracle_ofdm_structure(oracle):
    """ofdm_modulator_unittest.cpp:134-167: bin placement, guard zeros, scale and CP copy, for a single active subcarrier."""
    cfg = abi.OfdmConfig(1, 24, 512, 0, 2.0, 0.0)
    rg = 24 * 12
    for k in (0, 1, rg // 2 - 1, rg // 2, rg - 1):
        grid = np.zeros((1, 14, rg, 2), np.uint16)
        grid[0, 3, k, 0] = 0x3F80  # 1.0 in bf16
        iq = oracle.ofdm_slot(cfg, grid, 0)[0]
        sizes = [oracle._f("ofdm_symbol_size")(cfg, l) for l in range(14)]
        start = sum(sizes[:3])
        cp = sizes[3] - 512
        sym = iq[start + cp: start + sizes[3]]
        bin_ = k - rg // 2                        # subcarrier k sits at frequency bin k - rg/2 (mod N)
        want = 2.0 * np.exp(2j * np.pi * bin_ * np.arange(512) / 512)
        assert np.abs(sym - want).max() < 1e-5
        assert np.array_equal(iq[start: start + cp], sym[-cp:])
        assert not iq[:start].any() and not iq[start + sizes[3]:].any()


# ---- the reference's own unit-test configurations (tests/golden/ref_test_configs.npz) ---------------------------------
# Configurations read from the reference's test-data headers by oracle/ref/ref_testdata.cpp; expected outputs from the
# compiled reference on seeded payloads (the headers' .dat files are not in the reference checkout).

@pytest.fixture(scope="module")
def ref_cfgs():
    return np.load(os.path.join(cases.GOLDEN, "ref_test_configs.npz"))


def check_processor_case(backend, g, key, nof_subc):
    pdu = cases.pdsch_pdu_from_fixture(g, key)
    assert backend.validate(pdu) == 0, key
    d = backend.derive(pdu)
    tb = cases.ref_test_config_tb(g, key, pdu.tb_size_bytes)
    grid, rm, _ = backend.pdsch_process(pdu, tb, pdu.nof_ports, nof_subc, taps=True, codeword_bits=d["codeword_bits"])
    assert sha(rm) == str(g[key + "_cw_sha"]), key
    assert sha(grid) == str(g[key + "_grid_sha"]), key
    return pdu, grid


def test_ref_test_configs_pdsch_processor(oracle, ref_cfgs):
    """pdsch_processor_test_data.h: all 24 PDUs."""
    g = ref_cfgs
    assert int(g["proc_count"]) == 24
    for i in range(24):
        check_processor_case(oracle, g, "proc_%d" % i, int(g["proc_%d_rg" % i][0]) * 12)


def test_ref_test_configs_pdsch_encoder(oracle, ref_cfgs):
    """pdsch_encoder_test_data.h: all 168 segmenter configurations."""
    g = ref_cfgs
    assert g["enc_cfg"].shape == (168, 7)
    for i, (bg, rv, qm, nref, layers, nsym, tb_bytes) in enumerate(g["enc_cfg"].tolist()):
        tb = np.random.default_rng([ord("e"), i]).integers(0, 256, tb_bytes, dtype=np.uint8)
        assert sha(oracle.pdsch_encode_cfg(bg, rv, qm, nref, layers, nsym, tb)) == str(g["enc_cw_sha"][i]), i


def test_ref_test_configs_pdsch_modulator(oracle, ref_cfgs):
    """pdsch_modulator_test_data.h: all 36 configurations, as the PDUs that produce them."""
    g = ref_cfgs
    assert int(g["mod_count"]) == 36
    for i in range(36):
        key = "mod_%d" % i
        pdu = cases.pdsch_pdu_from_fixture(g, key)
        check_processor_case(oracle, g, key, (pdu.bwp_start_rb + pdu.bwp_size_rb) * 12)


def test_ref_test_configs_ldpc_segmenter(oracle, ref_cfgs):
    """ldpc_segmenter_test_data.h: the header's known answers (number of segments, segment length) and the segments."""
    g = ref_cfgs
    assert g["seg_cases"].shape == (11, 4)
    for i, (tbs_bits, bg, nof_segments, segment_length) in enumerate(g["seg_cases"].tolist()):
        tb = np.random.default_rng([ord("s"), i]).integers(0, 256, tbs_bits // 8, dtype=np.uint8)
        segs, meta, zc = oracle.segment(bg, 0, 2, 0, 1, 150, tb)
        assert segs.shape[0] == nof_segments and (22 if bg == 1 else 10) * zc == segment_length
        assert sha(segs) == str(g["seg_sha"][i])
        # the same answers from the PDU-level derivation the product's host side uses
        pdu = backends.abi.make_pdu(base_graph=bg, tb_size_bytes=tbs_bits // 8, prb_count=52, qm=2)
        d = oracle.derive(pdu)
        assert (d["nof_codeblocks"], d["segment_length"]) == (nof_segments, segment_length)


def test_ref_test_configs_ofdm_modulator(oracle, ref_cfgs):
    """ofdm_modulator_test_data.h: all 20 configurations (numerology, bandwidth, DFT size, cyclic prefix, scale, centre
    frequency) at the header's slot indices."""
    g = ref_cfgs
    assert g["ofdm_cases"].shape[0] == 20
    for i, row in enumerate(g["ofdm_cases"]):
        cfg = backends.abi.OfdmConfig(int(row[0]), int(row[1]), int(row[2]), int(row[3]), float(row[4]), float(row[5]))
        iq = oracle.ofdm_slot(cfg, cases.ref_test_config_grid(row, i), int(row[7]))
        assert iq.shape[1] == int(row[8])
        want = g["ofdm_%d_iq" % i]
        got = iq[0, g["ofdm_%d_idx" % i]]
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), i
        energy = float(np.sum(np.abs(iq.astype(np.complex128)) ** 2))
        assert abs(energy - float(g["ofdm_%d_energy" % i])) <= 1e-5 * energy


def check_dmrs_case(backend, g, i):
    key = "dmrs_%d" % i
    pdu = cases.pdsch_pdu_from_fixture(g, key)
    dmrs_type, _numerology, valid = g["dmrs_info"][i].tolist()
    assert pdu.dmrs_type == dmrs_type
    if not valid:
        assert backend.validate(pdu) != 0, key      # type 2: refused like pdsch_processor_validator_impl does
        return
    nof_subc = pdu.bwp_size_rb * 12
    _, grid = check_processor_case(backend, g, key, nof_subc)
    written = np.unpackbits(g[key + "_written"])[: pdu.nof_ports * 14 * nof_subc].astype(bool)
    values = grid.view(np.uint32).reshape(-1)[written]
    assert sha(values) == str(g[key + "_values_sha"]), key


def test_ref_test_configs_dmrs_pdsch(oracle, ref_cfgs):
    """dmrs_pdsch_processor_test_data.h: all 192 configurations.  The 96 type-1 ones run as the PDUs that produce them; the
    DM-RS positions and values must be those dmrs_pdsch_processor_impl::map wrote for the configuration itself."""
    g = ref_cfgs
    assert int(g["dmrs_count"]) == 192 and int(np.sum(g["dmrs_info"][:, 2])) == 96
    for i in range(192):
        check_dmrs_case(oracle, g, i)


# ---- soft demodulator ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("modulation", [0, 1, 2, 4, 6, 8])
def test_oracle_vs_ref_demodulation_mapper(oracle, ref, modulation):
    """oracle/nrphy_oracle_rx.c against the compiled demodulation_mapper_impl: every span length around the vector batch sizes
    (the leading multiple of the batch takes the AVX2 arithmetic, the tail the generic one), values on interval boundaries,
    rounding ties and near-zero components, and variances that are zero, negative or NaN."""
    if ref is None:
        pytest.skip("compiled reference not built")
    rng = np.random.default_rng(1000 + modulation)
    for n in list(range(1, 36)) + [100, 1003, 65536 + 5]:
        for kind in (0, 1, 2):
            sym, noise = cases.demod_inputs(rng, modulation, n, kind)
            got, want = oracle.demodulate_soft(modulation, sym, noise), ref.demodulate_soft(modulation, sym, noise)
            assert np.array_equal(got, want), (modulation, n, kind, np.flatnonzero(got != want)[:4])


def test_demodulation_mapper_golden(oracle):
    g = np.load(os.path.join(cases.GOLDEN, "demod.npz"))
    for modulation in (0, 1, 2, 4, 6, 8):
        for k, n in enumerate(g["lengths"].tolist()):
            for kind in (0, 1, 2):
                sym, noise = cases.demod_inputs(np.random.default_rng([modulation, n, kind]), modulation, n, kind)
                assert sha(oracle.demodulate_soft(modulation, sym, noise)) == str(g["sha_%d_%d" % (modulation, kind)][k])


def test_demodulation_mapper_tables_are_max_log_llrs(oracle):
    """The derived interval tables reproduce a brute-force max-log LLR over the constellation (TS 38.211 Section 5.1 mapping via
    the oracle's modulation mapper), away from rounding: |LLR - (min_{b=1} |v - a|^2 - min_{b=0} |v - a|^2)| small."""
    import ctypes as C
    for qm, norm in ((6, 42.0), (8, 170.0)):
        m = qm // 2
        # constellation of one dimension: map all bit patterns, keep real parts and the even-position bits
        pts = {}
        for word in range(1 << qm):
            bits = np.array([(word >> (qm - 1 - i)) & 1 for i in range(qm)], np.uint8)
            out, sc = oracle.modulate(qm, np.packbits(bits), 1)
            pts[tuple(bits[0::2])] = float(out[0, 0]) * sc
        a = np.array(list(pts.values()))
        labels = np.array(list(pts.keys()))
        assert abs(np.abs(a).max() - ((1 << m) - 1) / np.sqrt(norm)) < 1e-6
        xs = np.linspace(-1.3, 1.3, 2001)
        for pair in range(m):
            width = C.c_float()
            n = C.c_uint()
            slope = (C.c_float * 16)()
            intercept = (C.c_float * 16)()
            oracle.lib.oracle_demod_tables(qm, pair, C.byref(width), C.byref(n), slope, intercept)
            idx = np.clip(np.floor(xs / width.value).astype(int) + n.value // 2, 0, n.value - 1)
            got = np.array(slope[:16])[idx] * xs + np.array(intercept[:16])[idx]
            d = (xs[:, None] - a[None, :]) ** 2
            one = labels[:, pair] == 1
            want = d[:, one].min(axis=1) - d[:, ~one].min(axis=1)
            assert np.abs(got - want).max() < 1e-5, (qm, pair)


# ---- the reference's unit-test configurations of the round-2 components (tests/golden/ref_test_configs2.npz) ----------------
# Same scheme as above for the other downlink grid writers, the receive-side front end and the fronthaul compressor:
# configurations from the headers (oracle/ref/ref_testdata.cpp, sections 7-14), expected outputs from the compiled reference.

@pytest.fixture(scope="module")
def ref_cfgs2():
    return np.load(os.path.join(cases.GOLDEN, "ref_test_configs2.npz"))


class OracleApi2:
    """The calls check_ref_test_configs2 needs, on the oracle (test_gpu_parity.py has the same over the C ABI)."""

    def __init__(self, o):
        self.o = o
        self.pdcch_validate, self.pdcch_process, self.pdcch_encode = o.pdcch_validate, o.pdcch_process, o.pdcch_encode
        self.ssb_validate, self.ssb_process, self.pbch_encode = o.ssb_validate, o.ssb_process, o.pbch_encode
        self.csi_rs_validate, self.csi_rs_map = o.csi_rs_validate, o.csi_rs_map
        self.demodulate_soft, self.ofdm_demod_slot = o.demodulate_soft, o.ofdm_demod_slot

    def ofh_compress(self, cfg, prbs):
        return self.o.ofh_compress(cfg, prbs, simd=1)


def check_ref_test_configs2(api, g, what):
    abi = backends.abi
    if what == "pdcch":
        assert g["pdcch_valid"].shape == (114,) and int(g["pdcch_valid"].sum()) == 114
        for i in range(114):
            pdu = cases.struct_from_fixture(abi.PdcchPdu, g, "pdcch_%d" % i)
            assert api.pdcch_validate(pdu) == 0, i
            grid = cases.seeded_grid([7, i], pdu.nof_ports, 14, (pdu.bwp_start_rb + pdu.bwp_size_rb) * 12)
            assert sha(api.pdcch_process(pdu, grid)) == str(g["pdcch_%d_sha" % i]), i
    elif what == "ssb":
        assert g["ssb_valid"].shape == (240,) and int(g["ssb_valid"].sum()) == 228
        for i in range(240):
            pdu = cases.struct_from_fixture(abi.SsbPdu, g, "ssb_%d" % i)
            if not g["ssb_valid"][i]:
                # pattern case E, block starting at symbol 12: runs past the slot grid (the header's spy grid does not mind)
                assert pdu.pattern_case == 4 and api.ssb_validate(pdu) != 0, i
                continue
            assert api.ssb_validate(pdu) == 0, i
            grid = cases.seeded_grid([8, i], pdu.nof_ports, 14, int(g["ssb_%d_rb" % i]) * 12)
            assert sha(api.ssb_process(pdu, grid)) == str(g["ssb_%d_sha" % i]), i
    elif what == "csi":
        assert g["csi_valid"].shape == (102,) and int(g["csi_valid"].sum()) == 102
        for i in range(102):
            cfg = cases.struct_from_fixture(abi.CsiRsCfg, g, "csi_%d" % i)
            assert api.csi_rs_validate(cfg) == 0, i
            grid = cases.seeded_grid([9, i], cfg.nof_ports, 14, (cfg.start_rb + cfg.nof_rb) * 12)
            assert sha(api.csi_rs_map(cfg, grid)) == str(g["csi_%d_sha" % i]), i
    elif what == "dm":
        assert g["dm_cases"].shape == (12, 2)
        for i, (n, modulation) in enumerate(g["dm_cases"].tolist()):
            sym, noise = cases.demod_inputs(np.random.default_rng([10, i]), modulation, n, 0)
            assert sha(api.demodulate_soft(modulation, sym, noise)) == str(g["dm_sha"][i]), i
    elif what == "od":
        assert g["od_cases"].shape == (20, 10)
        for i, row in enumerate(g["od_cases"]):
            cfg = abi.OfdmConfig(int(row[0]), int(row[1]), int(row[2]), int(row[3]), float(row[4]), float(row[5]))
            slot, window, size = int(row[7]), int(row[8]), int(row[9])
            rng = np.random.default_rng([11, i])
            iq = (rng.standard_normal((1, size)) + 1j * rng.standard_normal((1, size))).astype(np.complex64)
            want = g["od_%d_grid" % i]
            got = api.ofdm_demod_slot(cfg, iq, slot, window)
            assert got.shape == want.shape, i
            assert_bf16_grids_close(got, want)   # cbf16 out of two float FFTs: within a bf16 ulp, 99 % identical
    elif what == "ofh":
        assert g["ofh_cases"].shape == (36, 4)
        ran = 0
        for i, (nof_prb, ctype, width, scaling) in enumerate(g["ofh_cases"].tolist()):
            if str(g["ofh_sha"][i]) == "":
                continue   # a compression type the ABI does not have (the reference builds none/BFP only, too) or width < 8
            cfg = abi.OfhCompressionCfg(int(ctype), int(width), float(scaling))
            prbs = cases.seeded_grid([12, i], int(nof_prb), 12)
            assert sha(api.ofh_compress(cfg, prbs)) == str(g["ofh_sha"][i]), i
            ran += 1
        assert ran >= 12
    elif what == "pe":
        assert g["pe_cases"].shape == (29, 3)
        for i, (E, rnti, k) in enumerate(g["pe_cases"].tolist()):
            assert k == cases.pdcch_encoder_payload_bits(i, E)
            payload = np.random.default_rng([13, i]).integers(0, 2, k, dtype=np.uint8)
            assert sha(api.pdcch_encode(payload, rnti, E)) == str(g["pe_sha"][i]), i
    elif what == "pb":
        assert g["pb_sha"].shape == (232,)
        for i in range(232):
            pdu = cases.pbch_message_pdu(cases.struct_from_fixture(abi.SsbPdu, g, "pb_%d" % i))
            assert sha(api.pbch_encode(pdu)) == str(g["pb_sha"][i]), i
    else:
        raise KeyError(what)


REF_TEST_CONFIGS2 = ("pdcch", "ssb", "csi", "dm", "od", "ofh", "pe", "pb")


@pytest.mark.parametrize("what", REF_TEST_CONFIGS2)
def test_ref_test_configs2(oracle, ref_cfgs2, what):
    """pdcch_processor (114), ssb_processor (240, of which 12 case-E blocks cross the slot boundary and are refused),
    nzp_csi_rs_generator (102), demodulation_mapper (12), ofdm_demodulator (20), ofh_compression (36), pdcch_encoder (29) and
    pbch_encoder (232) test-data headers on the oracle."""
    check_ref_test_configs2(OracleApi2(oracle), ref_cfgs2, what)
