#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the COMPILED REFERENCE (oracle/_ref/libsrsref.so, built by
`make -C oracle ref` from /root/reference).  Run in the build container only; the outputs (small .npz files: inputs +
expected outputs, never reference source) are committed and travel to the GPU box.

    python tests/golden/generate.py [section ...]     (no argument: every section)

Sections: base (the round-1 files: codebooks, ldpc_encoder, pdsch_processor, ofdm_modulator, ofdm_demodulator),
ofdm_sizes (DFT sizes 4608 / 6144: ofdm_sizes.npz), dl_control (PDCCH and SS/PBCH block processors: dl_control.npz),
ref_test_configs (the configurations of the reference's own unit-test vectors, read from its test-data headers by
oracle/ref/ref_testdata.cpp, with the compiled reference's outputs on seeded payloads: ref_test_configs.npz), demod (soft
demodulator: demod.npz), ref_test_configs2 (the same for the test-data headers of the round-2 components: ref_test_configs2.npz).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import backends  # noqa: E402

abi = backends.abi
r = backends.ref()
assert r is not None, "build oracle/_ref first: make -C oracle ref"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


SECTIONS = sys.argv[1:] or ["base", "ofdm_sizes", "dl_control", "ref_test_configs", "demod", "ref_test_configs2"]


def section_ofdm_sizes():
    """OFDM modulator / demodulator at the two DFT sizes added in round 2 (15 kHz SCS: 6144 = 92.16 Msps, the size the
    reference's ofdm_modulator_unittest sweeps; 4608 = 69.12 Msps)."""
    rng = np.random.default_rng(6144)
    og = {}
    for name, (mu, bw, n, fc, slot) in {"n6144": (0, 273, 6144, 3.5e9, 0), "n4608": (0, 270, 4608, 2.4e9, 0)}.items():
        cfg = abi.OfdmConfig(mu, bw, n, 0, 1.0 / np.sqrt(n), fc)
        grid = (rng.standard_normal((1, 14, bw * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        og[name + "_grid"] = grid
        og[name + "_iq"] = r.ofdm_slot(cfg, grid, slot)
        og[name + "_cfg"] = np.array([mu, bw, n, fc, slot], dtype=np.float64)
    for name, (mu, bw, n, fc, slot, wo) in {"d6144": (0, 273, 6144, 3.5e9, 0, 0), "d4608w": (0, 270, 4608, 2.4e9, 0, 11)}.items():
        cfg = abi.OfdmConfig(mu, bw, n, 0, 1.0 / np.sqrt(n), fc)
        size = backends.pkg.lib.slot_size(cfg, slot)
        iq = (rng.standard_normal((1, size)) + 1j * rng.standard_normal((1, size))).astype(np.complex64)
        og[name + "_iq"] = iq
        og[name + "_grid"] = r.ofdm_demod_slot(cfg, iq, slot, wo)
        og[name + "_cfg"] = np.array([mu, bw, n, fc, slot, wo], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "ofdm_sizes.npz"), **og)


def section_dl_control():
    """PDCCH and SS/PBCH block processors of the compiled reference on seeded PDUs (tests/cases.py generators; the PDUs
    are stored field by field, so the fixture does not depend on the generators staying the same): encoder outputs and
    the grid words the processor wrote into an all-zero grid."""
    import cases
    rng = np.random.default_rng(38213)
    g = {}
    n_pdcch, n_ssb = 24, 12
    for i in range(n_pdcch):
        pdu = cases.random_pdcch(rng)
        grid = r.pdcch_process(pdu, np.zeros((4, 14, 52 * 12, 2), np.uint16), simd=1)
        assert np.array_equal(grid, r.pdcch_process(pdu, np.zeros((4, 14, 52 * 12, 2), np.uint16), simd=0))
        g.update(cases.pdu_to_arrays(pdu, "pdcch%02d_" % i))
        payload = np.array(list(pdu.payload)[: pdu.payload_size], np.uint8)
        g["pdcch%02d_encoded" % i] = np.packbits(r.pdcch_encode(payload, pdu.rnti, 108 * pdu.aggregation_level))
        w = grid.view(np.uint32).reshape(-1)
        idx = np.flatnonzero(w)
        g["pdcch%02d_idx" % i] = idx.astype(np.uint32)
        g["pdcch%02d_val" % i] = w[idx]
    for i in range(n_ssb):
        pdu = cases.random_ssb(rng, nof_ports=3)
        grid = r.ssb_process(pdu, np.zeros((3, 14, 52 * 12, 2), np.uint16))
        g.update(cases.pdu_to_arrays(pdu, "ssb%02d_" % i))
        g["ssb%02d_encoded" % i] = np.packbits(r.pbch_encode(pdu))
        w = grid.view(np.uint32).reshape(-1)
        idx = np.flatnonzero(w)   # the SSS writes -0 imaginary parts: non-zero words, kept
        g["ssb%02d_idx" % i] = idx.astype(np.uint32)
        g["ssb%02d_val" % i] = w[idx]
    g["n_pdcch"], g["n_ssb"] = np.array(n_pdcch), np.array(n_ssb)
    np.savez_compressed(os.path.join(HERE, "dl_control.npz"), **g)


def section_ref_test_configs():
    """Every configuration of the six test-data headers SURVEY.md section 8c lists.  The headers' .dat payloads are not in the
    reference checkout, so payloads are seeded here and the expected outputs come from the compiled reference:
      proc_*  pdsch_processor_test_data.h (24): the PDU verbatim -> codeword and grid hashes
      enc_*   pdsch_encoder_test_data.h (168): segmenter_config verbatim -> codeword hash
      mod_*   pdsch_modulator_test_data.h (36): config_t as the PDU that produces it -> codeword and grid hashes
      seg_*   ldpc_segmenter_test_data.h (11): the header's own known answers (segments, segment length) + segment hashes
      ofdm_*  ofdm_modulator_test_data.h (20): configuration, port and slot verbatim -> sampled IQ
      dmrs_*  dmrs_pdsch_processor_test_data.h (192): config_t verbatim through dmrs_pdsch_processor_impl::map -> written
              positions and values; and as the PDU that produces it through the PDSCH processor -> grid hash.
    Transport block sizes (absent with the .dat files) follow one rule: half the codeword capacity, at most 478 bytes for
    base graph 2 (its 3824-bit limit), at least 3 bytes."""
    import ctypes as C
    import cases
    h = r.lib
    o = backends.oracle()
    g = {}
    sizeof_pdu = C.sizeof(abi.PdschPdu)
    g["sizeof_pdu"] = np.array(sizeof_pdu)

    def tb_rule(codeword_bits, bg):
        n = max(3, codeword_bits // 16)
        return min(n, 478) if bg == 2 else n

    def fetch_pdu(fn, i, *extra):
        pdu = abi.PdschPdu()
        w = np.zeros(2 * 4 * 4 * 16, np.float32)
        n = fn(i, C.byref(pdu), w.ctypes.data_as(C.c_void_p), w.size, *extra)
        assert n > 0, n
        return pdu, w[:n].copy()

    def finish_pdu(pdu, w):
        cases.attach_weights(pdu, w)
        d = o.derive(pdu)
        pdu.tb_size_bytes = tb_rule(d["codeword_bits"], pdu.ldpc_base_graph)
        return o.derive(pdu)

    def run_processor(prefix, i, pdu, w, nof_subc):
        d = finish_pdu(pdu, w)
        tb = np.random.default_rng([ord(prefix[0]), i]).integers(0, 256, pdu.tb_size_bytes, dtype=np.uint8)
        assert r.validate(pdu) == 0 and o.validate(pdu) == 0, (prefix, i)
        grid = r.pdsch_process(pdu, tb, pdu.nof_ports, nof_subc, simd=1)
        cw = r.pdsch_encode(pdu, tb, d)[: (d["codeword_bits"] + 7) // 8]
        g["%s%d_pod" % (prefix, i)] = np.frombuffer(bytes(pdu), np.uint8).copy()
        g["%s%d_weights" % (prefix, i)] = w
        g["%s%d_tb_seed" % (prefix, i)] = np.array([ord(prefix[0]), i])
        g["%s%d_cw_sha" % (prefix, i)] = np.array(sha(cw))
        g["%s%d_grid_sha" % (prefix, i)] = np.array(sha(grid))
        return grid

    # 1. pdsch_processor_test_data.h
    fn = h.ref_testdata_pdsch_processor
    n_proc = fn(0, None, None, 0, None)
    g["proc_count"] = np.array(n_proc)
    for i in range(n_proc):
        rg = (C.c_uint * 2)()
        pdu, w = fetch_pdu(fn, i, rg)
        assert rg[1] == 14
        g["proc_%d_rg" % i] = np.array([rg[0], rg[1]])
        run_processor("proc_", i, pdu, w, rg[0] * 12)

    # 2. pdsch_encoder_test_data.h
    fn = h.ref_testdata_pdsch_encoder
    n_enc = fn(0, None)
    cfgs, shas, tbs = [], [], []
    for i in range(n_enc):
        c = (C.c_uint * 6)()
        fn(i, c)
        bg, rv, qm, nref, layers, nsym = list(c)
        tb_bytes = tb_rule(nsym * qm, bg)
        tb = np.random.default_rng([ord("e"), i]).integers(0, 256, tb_bytes, dtype=np.uint8)
        unpacked = np.zeros(nsym * qm, np.uint8)
        assert h.ref_pdsch_encode(bg, rv, qm, nref, layers, nsym, tb.ctypes.data_as(C.c_void_p), tb_bytes,
                                  unpacked.ctypes.data_as(C.c_void_p), 1) == 0
        cfgs.append([bg, rv, qm, nref, layers, nsym, tb_bytes])
        shas.append(sha(np.packbits(unpacked)))
    g["enc_cfg"] = np.array(cfgs, np.uint32)
    g["enc_cw_sha"] = np.array(shas)

    # 3. pdsch_modulator_test_data.h
    fn = h.ref_testdata_pdsch_modulator
    n_mod = fn(0, None, None, 0)
    g["mod_count"] = np.array(n_mod)
    for i in range(n_mod):
        pdu, w = fetch_pdu(fn, i)
        run_processor("mod_", i, pdu, w, (pdu.bwp_start_rb + pdu.bwp_size_rb) * 12)

    # 4. ldpc_segmenter_test_data.h
    fn = h.ref_testdata_ldpc_segmenter
    n_seg = fn(0, None)
    rows, shas = [], []
    for i in range(n_seg):
        c = (C.c_uint * 4)()
        fn(i, c)
        tbs_bits, bg, nof_segments, segment_length = list(c)
        tb = np.random.default_rng([ord("s"), i]).integers(0, 256, tbs_bits // 8, dtype=np.uint8)
        segs, meta, zc = r.segment(bg, 0, 2, 0, 1, 150, tb)      # ldpc_segmenter_test.cpp:112-118
        assert segs.shape[0] == nof_segments and (22 if bg == 1 else 10) * zc == segment_length, (i, segs.shape, zc)
        rows.append([tbs_bits, bg, nof_segments, segment_length])
        shas.append(sha(segs))
    g["seg_cases"] = np.array(rows, np.uint32)
    g["seg_sha"] = np.array(shas)

    # 5. ofdm_modulator_test_data.h
    fn = h.ref_testdata_ofdm_modulator
    n_ofdm = fn(0, None, None)
    rows = []
    for i in range(n_ofdm):
        cfg = abi.OfdmConfig()
        extra = (C.c_uint * 2)()
        fn(i, C.byref(cfg), extra)
        rng = np.random.default_rng([ord("o"), i])
        nsymb = 12 if cfg.cp else 14
        grid = np.zeros((1, 14, cfg.bw_rb * 12, 2), np.uint16)
        grid[:, :nsymb] = (rng.standard_normal((1, nsymb, cfg.bw_rb * 12, 2)).astype(np.float32).view(np.uint32) >> 16)
        iq = r.ofdm_slot(cfg, grid, int(extra[1]))
        n = iq.shape[1]
        idx = np.unique(np.concatenate([np.arange(256), np.arange(n - 256, n), rng.integers(0, n, 1536)]))
        rows.append([cfg.numerology, cfg.bw_rb, cfg.dft_size, cfg.cp, cfg.scale, cfg.center_freq_hz, extra[0], extra[1], n])
        g["ofdm_%d_idx" % i] = idx.astype(np.uint32)
        g["ofdm_%d_iq" % i] = iq[0, idx]
        g["ofdm_%d_energy" % i] = np.array(float(np.sum(np.abs(iq.astype(np.complex128)) ** 2)))
    g["ofdm_cases"] = np.array(rows, np.float64)

    # 6. dmrs_pdsch_processor_test_data.h
    fn = h.ref_testdata_dmrs_pdsch
    n_dmrs = fn(0, None, None, 0, None)
    g["dmrs_count"] = np.array(n_dmrs)
    info_rows = []
    marker = np.uint16(0x7FC1)
    for i in range(n_dmrs):
        info = (C.c_uint * 4)()
        pdu, w = fetch_pdu(fn, i, info)
        dmrs_type, numerology, amp_ok, k_rb = list(info)
        assert amp_ok == 1 and k_rb == 0
        nof_subc = pdu.bwp_size_rb * 12
        # The configuration verbatim through dmrs_pdsch_processor_impl::map.
        dm = np.full((pdu.nof_ports, 14, nof_subc, 2), marker, np.uint16)
        assert h.ref_testdata_dmrs_pdsch_map(i, dm.ctypes.data_as(C.c_void_p), pdu.nof_ports, nof_subc, 1) == 0
        written = (dm.view(np.uint32).reshape(pdu.nof_ports, 14, nof_subc) != 0x7FC17FC1)
        values = dm.view(np.uint32).reshape(pdu.nof_ports, 14, nof_subc)[written]
        g["dmrs_%d_written" % i] = np.packbits(written.reshape(-1))
        g["dmrs_%d_values_sha" % i] = np.array(sha(values))
        cases.attach_weights(pdu, w)
        if dmrs_type == 2:
            # pdsch_processor_validator_impl refuses type 2; so must the C ABI.  Only the DM-RS-only result is kept.
            pdu.tb_size_bytes = 16
            assert r.validate(pdu) != 0 and o.validate(pdu) != 0, i
            g["dmrs_%d_pod" % i] = np.frombuffer(bytes(pdu), np.uint8).copy()
            g["dmrs_%d_weights" % i] = w
            info_rows.append([dmrs_type, numerology, 0])
            continue
        grid = run_processor("dmrs_", i, pdu, w, nof_subc)
        # The PDU reproduces the configuration: the processor's grid holds exactly those values at those positions.
        assert np.array_equal(grid.view(np.uint32).reshape(pdu.nof_ports, 14, nof_subc)[written], values), i
        info_rows.append([dmrs_type, numerology, 1])
    g["dmrs_info"] = np.array(info_rows, np.uint32)
    np.savez_compressed(os.path.join(HERE, "ref_test_configs.npz"), **g)
    print("ref_test_configs:", n_proc, n_enc, n_mod, n_seg, n_ofdm, n_dmrs)


def section_demod():
    """Soft demodulator (demodulation_mapper_impl, AVX2 + generic paths as compiled in oracle/_ref): hashes of the soft bits for
    seeded inputs (tests/cases.py demod_inputs) at span lengths around the vector batch sizes."""
    import cases
    lengths = [1, 3, 4, 7, 8, 15, 16, 17, 31, 33, 100, 1003, 20011]
    g = {"lengths": np.array(lengths)}
    for modulation in (0, 1, 2, 4, 6, 8):
        for kind in (0, 1, 2):
            shas = []
            for n in lengths:
                sym, noise = cases.demod_inputs(np.random.default_rng([modulation, n, kind]), modulation, n, kind)
                shas.append(sha(r.demodulate_soft(modulation, sym, noise)))
            g["sha_%d_%d" % (modulation, kind)] = np.array(shas)
    np.savez_compressed(os.path.join(HERE, "demod.npz"), **g)


def section_ref_test_configs2():
    """Configurations of the reference's unit-test headers for the components built in round 2 (oracle/ref/ref_testdata.cpp, 7-14),
    payloads seeded here, expected outputs from the compiled reference:
      pdcch_*  pdcch_processor_test_data.h (114): PDU verbatim (DCI payload included) -> grid hash
      ssb_*    ssb_processor_test_data.h (240): PDU verbatim -> grid hash
      csi_*    nzp_csi_rs_generator_test_data.h (102): configuration verbatim -> grid hash (rows 1-5; the others are recorded as
               outside this library's rows and must be refused)
      dm_*     demodulation_mapper_test_data.h (12): (symbols, modulation) -> soft-bit hash on seeded inputs
      od_*     ofdm_demodulator_test_data.h (20): configuration, slot, window offset -> grid on seeded IQ
      ofh_*    ofh_compression_test_data.h (36): (PRBs, type, width, scaling) -> bytes hash on seeded PRBs
      pe_*     pdcch_encoder_test_data.h (29): (E, RNTI) -> encoded-bits hash, payload length by cases.pdcch_encoder_payload_bits
      pb_*     pbch_encoder_test_data.h (232): message as the block PDU that produces it -> 864 encoded bits hash."""
    import ctypes as C
    import cases
    h = r.lib
    o = backends.oracle()
    g = {}

    def raw(obj):
        return np.frombuffer(bytes(obj), np.uint8).copy()

    # 7. PDCCH processor
    fn = h.ref_testdata_pdcch_processor
    n = fn(0, None, None, 0)
    flags = []
    for i in range(n):
        pdu = abi.PdcchPdu()
        w = np.zeros(2 * 16 * 4, np.float32)
        nw = fn(i, C.byref(pdu), w.ctypes.data_as(C.c_void_p), w.size)
        assert nw > 0
        g["pdcch_%d_pod" % i], g["pdcch_%d_weights" % i] = raw(pdu), w[:nw].copy()
        pdu = cases.struct_from_fixture(abi.PdcchPdu, g, "pdcch_%d" % i)
        ok = o.pdcch_validate(pdu) == 0
        flags.append(int(ok))
        if ok:
            nof_rb = pdu.bwp_start_rb + pdu.bwp_size_rb
            grid = cases.seeded_grid([7, i], pdu.nof_ports, 14, nof_rb * 12)
            g["pdcch_%d_sha" % i] = np.array(sha(r.pdcch_process(pdu, grid)))
    g["pdcch_valid"] = np.array(flags, np.uint8)

    # 8. SS/PBCH block processor
    fn = h.ref_testdata_ssb_processor
    n = fn(0, None)
    flags = []
    for i in range(n):
        pdu = abi.SsbPdu()
        fn(i, C.byref(pdu))
        # The reference's unit test writes into a spy grid that takes any port index (the header's cases carry indices up to
        # 63); this ABI's grids have at most four ports: the block's ports are renumbered 0, 1, ... (the index only selects
        # the grid plane, the values do not depend on it).  The original indices are kept for the record.
        g["ssb_%d_ports" % i] = np.array(list(pdu.ports)[: pdu.nof_ports], np.uint8)
        for k in range(pdu.nof_ports):
            pdu.ports[k] = k
        g["ssb_%d_pod" % i] = raw(pdu)
        ok = o.ssb_validate(pdu) == 0
        flags.append(int(ok))
        if ok:
            ports = max(list(pdu.ports)[: pdu.nof_ports]) + 1
            nof_rb = cases.ssb_grid_rb(pdu)
            g["ssb_%d_rb" % i] = np.array(nof_rb)
            grid = cases.seeded_grid([8, i], ports, 14, nof_rb * 12)
            g["ssb_%d_sha" % i] = np.array(sha(r.ssb_process(pdu, grid)))
    g["ssb_valid"] = np.array(flags, np.uint8)

    # 9. NZP-CSI-RS generator
    fn = h.ref_testdata_nzp_csi_rs
    n = fn(0, None, None, 0)
    flags = []
    for i in range(n):
        cfg = abi.CsiRsCfg()
        w = np.zeros(2 * 32 * 32, np.float32)
        nw = fn(i, C.byref(cfg), w.ctypes.data_as(C.c_void_p), w.size)
        assert nw > 0
        g["csi_%d_pod" % i], g["csi_%d_weights" % i] = raw(cfg), w[:nw].copy()
        cfg = cases.struct_from_fixture(abi.CsiRsCfg, g, "csi_%d" % i)
        ok = o.csi_rs_validate(cfg) == 0
        flags.append(int(ok))
        if ok:
            grid = cases.seeded_grid([9, i], cfg.nof_ports, 14, (cfg.start_rb + cfg.nof_rb) * 12)
            g["csi_%d_sha" % i] = np.array(sha(r.csi_rs_map(cfg, grid)))
    g["csi_valid"] = np.array(flags, np.uint8)

    # 10. demodulation mapper
    fn = h.ref_testdata_demodulation_mapper
    n = fn(0, None)
    rows, shas = [], []
    for i in range(n):
        c = (C.c_uint * 2)()
        fn(i, c)
        sym, noise = cases.demod_inputs(np.random.default_rng([10, i]), int(c[1]), int(c[0]), 0)
        rows.append([c[0], c[1]])
        shas.append(sha(r.demodulate_soft(int(c[1]), sym, noise)))
    g["dm_cases"], g["dm_sha"] = np.array(rows, np.uint32), np.array(shas)

    # 11. OFDM demodulator
    fn = h.ref_testdata_ofdm_demodulator
    n = fn(0, None, None)
    rows = []
    for i in range(n):
        cfg = abi.OfdmConfig()
        extra = (C.c_uint * 3)()
        fn(i, C.byref(cfg), extra)
        size = backends.pkg.lib.slot_size(cfg, int(extra[1]))
        rng = np.random.default_rng([11, i])
        iq = (rng.standard_normal((1, size)) + 1j * rng.standard_normal((1, size))).astype(np.complex64)
        rows.append([cfg.numerology, cfg.bw_rb, cfg.dft_size, cfg.cp, cfg.scale, cfg.center_freq_hz, extra[0], extra[1], extra[2], size])
        g["od_%d_grid" % i] = r.ofdm_demod_slot(cfg, iq, int(extra[1]), int(extra[2]))
    g["od_cases"] = np.array(rows, np.float64)

    # 12. OFH compression
    fn = h.ref_testdata_ofh_compression
    n = fn(0, None, None)
    rows, shas = [], []
    for i in range(n):
        c = (C.c_uint * 3)()
        sc = C.c_float()
        fn(i, c, C.byref(sc))
        rows.append([c[0], c[1], c[2], sc.value])
        if c[1] > 1 or c[2] < 8:
            shas.append("")   # a compression type / width outside this library's ABI
            continue
        cfg = abi.OfhCompressionCfg(int(c[1]), int(c[2]), float(sc.value))
        prbs = cases.seeded_grid([12, i], int(c[0]), 12)
        a, b = r.ofh_compress(cfg, prbs, simd=1), r.ofh_compress(cfg, prbs, simd=0)
        shas.append(sha(a))
        g["ofh_%d_generic_differs" % i] = np.array(int(not np.array_equal(a, b)))
    g["ofh_cases"], g["ofh_sha"] = np.array(rows, np.float64), np.array(shas)

    # 13. PDCCH encoder
    fn = h.ref_testdata_pdcch_encoder
    n = fn(0, None)
    rows, shas = [], []
    for i in range(n):
        c = (C.c_uint * 2)()
        fn(i, c)
        k = cases.pdcch_encoder_payload_bits(i, int(c[0]))
        payload = np.random.default_rng([13, i]).integers(0, 2, k, dtype=np.uint8)
        rows.append([c[0], c[1], k])
        shas.append(sha(r.pdcch_encode(payload, int(c[1]), int(c[0]))))
    g["pe_cases"], g["pe_sha"] = np.array(rows, np.uint32), np.array(shas)

    # 14. PBCH encoder
    fn = h.ref_testdata_pbch_encoder
    n = fn(0, None)
    shas = []
    for i in range(n):
        pdu = abi.SsbPdu()
        fn(i, C.byref(pdu))
        g["pb_%d_pod" % i] = raw(pdu)
        shas.append(sha(r.pbch_encode(pdu)))
    g["pb_sha"] = np.array(shas)
    np.savez_compressed(os.path.join(HERE, "ref_test_configs2.npz"), **g)
    print("ref_test_configs2:", {k: (int(np.sum(g[k])), len(g[k])) for k in ("pdcch_valid", "ssb_valid", "csi_valid")},
          len(g["dm_sha"]), len(g["od_cases"]), len(g["ofh_sha"]), len(g["pe_sha"]), len(g["pb_sha"]))


if "ref_test_configs2" in SECTIONS:
    section_ref_test_configs2()
if "demod" in SECTIONS:
    section_demod()
if "ref_test_configs" in SECTIONS:
    section_ref_test_configs()
if "ofdm_sizes" in SECTIONS:
    section_ofdm_sizes()
if "dl_control" in SECTIONS:
    section_dl_control()
if "base" not in SECTIONS:
    print("golden vectors written to", HERE, SECTIONS)
    sys.exit(0)


# 1. Precoding codebooks the BASELINE configs use.
np.savez(os.path.join(HERE, "codebooks.npz"),
         single_port=r.codebook(1), two_layer_two_ports_0=r.codebook(3, 0),
         four_layer_four_ports_0_0=r.codebook(7, 0, 0), one_layer_two_ports_1=r.codebook(2, 1),
         three_layer_four_ports_1_0=r.codebook(6, 1, 0), two_layer_four_ports_1_0_1=r.codebook(5, 1, 0, 1))

import cases  # noqa: E402  (needs codebooks.npz)

# 2. LDPC encoder: every (base graph, lifting size), one random message, full-length output
#    (the configurations of ldpc_enc_dec_test.cpp: 51 lifting sizes x 2 base graphs).
rng = np.random.default_rng(38212)
msgs, outs, keys = [], [], []
for bg in (1, 2):
    kb = 22 if bg == 1 else 10
    for zc in [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 44, 48, 52,
               56, 60, 64, 72, 80, 88, 96, 104, 112, 120, 128, 144, 160, 176, 192, 208, 224, 240, 256, 288, 320, 352,
               384]:
        msg = np.packbits(rng.integers(0, 2, kb * zc, dtype=np.uint8))
        n = (66 if bg == 1 else 50) * zc
        out = r.ldpc_encode(bg, zc, msg, n, simd=1)
        assert np.array_equal(out, r.ldpc_encode(bg, zc, msg, n, simd=0))
        keys.append((bg, zc))
        msgs.append(msg)
        outs.append(out)
np.savez_compressed(os.path.join(HERE, "ldpc_encoder.npz"), keys=np.array(keys),
                    **{"msg_%d_%d" % k: m for k, m in zip(keys, msgs)},
                    **{"out_%d_%d" % k: o for k, o in zip(keys, outs)})

# 3. PDSCH processor: BASELINE configs 1-3 + unit-test-like PDUs.  Stored: seed for the TB, SHA-256 of the codeword
#    and of the grid, and the first/last 64 grid words of every port (small files; the TB is regenerated from the seed).
def pdu_fields(pdu):
    return {n: np.array(getattr(pdu, n)) for n, _ in abi.PdschPdu._fields_
            if n not in ("prb_mask", "reserved", "precoding")}


golden = {}
items = [("cfg%d" % c,) + cases.baseline_config(c)[:3] for c in (1, 2, 3)]
for i, pdu in enumerate(cases.unit_test_like_pdus(np.random.default_rng(2024))):
    items.append(("unit%02d" % i, pdu, 4, 26 * 12))
o = backends.oracle()
for name, pdu, nof_ports, nof_subc in items:
    tb = np.random.default_rng(abs(hash(name)) % 2**32 if False else sum(map(ord, name))).integers(
        0, 256, pdu.tb_size_bytes, dtype=np.uint8)
    d = o.derive(pdu)
    grid = r.pdsch_process(pdu, tb, nof_ports, nof_subc, simd=1)
    cw = r.pdsch_encode(pdu, tb, d)[: (d["codeword_bits"] + 7) // 8]
    g32 = grid.view(np.uint32).reshape(nof_ports, -1)
    nz = [np.flatnonzero(g32[p]) for p in range(nof_ports)]
    golden[name + "_tb_seed"] = np.array(sum(map(ord, name)))
    golden[name + "_cw_sha"] = np.array(sha(cw))
    golden[name + "_grid_sha"] = np.array(sha(grid))
    golden[name + "_cw_head"] = cw[:64].copy()
    for p in range(nof_ports):
        idx = nz[p][:64] if len(nz[p]) else np.zeros(0, np.int64)
        golden["%s_p%d_idx" % (name, p)] = idx
        golden["%s_p%d_val" % (name, p)] = g32[p][idx]
np.savez_compressed(os.path.join(HERE, "pdsch_processor.npz"), **golden)

# 4. OFDM modulator: random bf16 grid rows -> reference IQ (generic DFT; FFTW is absent here), sub-sampled.
rng = np.random.default_rng(38211)
og = {}
for name, (mu, bw, n, fc, slot) in {"n4096": (1, 273, 4096, 3.5e9, 1), "n2048": (0, 106, 2048, 2.4e9, 0),
                                    "n1024": (0, 52, 1024, 2.4e9, 0),
                                    # the 3 * 2^k sizes of the 23.04 MHz family of sampling rates
                                    "n1536": (0, 106, 1536, 2.4e9, 0), "n3072": (1, 217, 3072, 3.5e9, 0),
                                    "n768": (0, 52, 768, 2.4e9, 0), "n384": (0, 25, 384, 2.4e9, 0)}.items():
    cfg = abi.OfdmConfig(mu, bw, n, 0, 1.0 / np.sqrt(n), fc)
    grid = (rng.standard_normal((1, 14, bw * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    iq = r.ofdm_slot(cfg, grid, slot)
    og[name + "_grid"] = grid
    og[name + "_iq"] = iq
    og[name + "_cfg"] = np.array([mu, bw, n, fc, slot], dtype=np.float64)
# extended cyclic prefix (12 symbols per slot, 60 kHz SCS: the numerology-2 rows of ofdm_modulator_test_data.h); own
# generator so that the vectors above do not change
rng = np.random.default_rng(38213)
for name, (mu, bw, n, fc, slot) in {"x512": (2, 24, 512, 3.5e9, 3), "x2048": (2, 96, 2048, 28e9, 0)}.items():
    cfg = abi.OfdmConfig(mu, bw, n, 1, 1.0 / np.sqrt(n), fc)
    grid = (rng.standard_normal((1, 14, bw * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    og[name + "_grid"] = grid
    og[name + "_iq"] = r.ofdm_slot(cfg, grid, slot)
    og[name + "_cfg"] = np.array([mu, bw, n, fc, slot, 1], dtype=np.float64)
np.savez_compressed(os.path.join(HERE, "ofdm_modulator.npz"), **og)

# 5. OFDM demodulator ("next" row, PUSCH receive side): random IQ -> reference grid (cbf16), with and without a DFT
#    window offset.
rng = np.random.default_rng(38212)
dg = {}
for name, (mu, bw, n, fc, slot, wo) in {"d4096": (1, 273, 4096, 3.5e9, 1, 0), "d2048w": (0, 106, 2048, 2.4e9, 0, 9),
                                        "d1536": (0, 106, 1536, 2.4e9, 0, 0), "d512w": (0, 25, 512, 2.4e9, 0, 3)}.items():
    cfg = abi.OfdmConfig(mu, bw, n, 0, 1.0 / np.sqrt(n), fc)
    size = backends.pkg.lib.slot_size(cfg, slot)
    iq = (rng.standard_normal((2, size)) + 1j * rng.standard_normal((2, size))).astype(np.complex64)
    dg[name + "_iq"] = iq
    dg[name + "_grid"] = r.ofdm_demod_slot(cfg, iq, slot, wo)
    dg[name + "_cfg"] = np.array([mu, bw, n, fc, slot, wo], dtype=np.float64)
rng = np.random.default_rng(38214)
for name, (mu, bw, n, fc, slot, wo) in {"dx1024w": (2, 48, 1024, 3.5e9, 1, 7)}.items():
    cfg = abi.OfdmConfig(mu, bw, n, 1, 1.0 / np.sqrt(n), fc)
    size = backends.pkg.lib.slot_size(cfg, slot)
    iq = (rng.standard_normal((2, size)) + 1j * rng.standard_normal((2, size))).astype(np.complex64)
    dg[name + "_iq"] = iq
    dg[name + "_grid"] = r.ofdm_demod_slot(cfg, iq, slot, wo)
    dg[name + "_cfg"] = np.array([mu, bw, n, fc, slot, wo, 1], dtype=np.float64)
np.savez_compressed(os.path.join(HERE, "ofdm_demodulator.npz"), **dg)
print("golden vectors written to", HERE)
