"""Runtime evidence for the srsRAN-side adaptors (srsran-edgeric-5g_amd/adaptors/mi355_nrphy_srsran.h).

Build container only (needs /root/reference): oracle/_ref/libadaptor_test.so = the adaptors compiled against the
reference's headers + the compiled reference + a MOCK of the C ABI backed by the CPU oracle
(oracle/ref/adaptor_harness.cpp, `make -C oracle adaptors`).  Every test pushes reference-side objects through an adaptor
and through the reference's own implementation and compares the caller-visible results.  What is under test is the
adaptors' logic (pdu_t -> POD, RE masks, grid access through the mapper, asynchronous completion, HAL protocol, slot
cache); the library behind the real ABI is tested against the same oracle on the GPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import backends
import cases

abi = backends.abi
_vp, _u32, _i = C.c_void_p, C.c_uint32, C.c_int


@pytest.fixture(scope="module")
def harness():
    if not os.path.isdir("/root/reference/srsRAN-5G-ER"):
        pytest.skip("reference sources not available on this machine")
    oracle_dir = os.path.join(backends.ROOT, "oracle")
    subprocess.run(["make", "-C", oracle_dir, "oracle"], check=True, capture_output=True, timeout=600)
    subprocess.run(["make", "-C", oracle_dir, "ref", "-j8"], check=True, capture_output=True, timeout=3000)
    subprocess.run(["make", "-C", oracle_dir, "adaptors"], check=True, capture_output=True, timeout=900)
    return C.CDLL(os.path.join(oracle_dir, "_ref", "libadaptor_test.so"))


def _p(a):
    return a.ctypes.data_as(_vp)


def random_grid(rng, *shape):
    return (rng.standard_normal(shape + (2,)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)


@pytest.mark.parametrize("with_writer_access", [1, 0])
def test_pdsch_processor_adaptor_asynchronous(harness, with_writer_access):
    """pdsch_processor_adaptor: 10 PDUs (layers 1-4, reserved RE patterns, several modulations) submitted back to back
    with 3 in flight.  process() returns before its PDU completes, every notifier fires exactly once from another
    thread, and each caller grid equals what pdsch_processor_impl writes into the same initial grid -- through the
    writer-access grid (bit-exact) and through a plain reference grid (mapper.map with identity precoding)."""
    rng = np.random.default_rng(31)
    pdus = (cases.unit_test_like_pdus(rng) * 2)[:10]
    nof_ports, nof_subc, n = 4, 26 * 12, len(pdus)
    tbs = [cases.random_tb(rng, p) for p in pdus]
    arr = (abi.PdschPdu * n)(*pdus)
    tb_ptrs = (C.c_void_p * n)(*[t.ctypes.data for t in tbs])
    init = random_grid(rng, n, nof_ports, 14, nof_subc)
    got, want = np.zeros_like(init), np.zeros_like(init)
    harness.adaptor_test_pdsch.restype = _i
    rc = harness.adaptor_test_pdsch(_u32(n), arr, tb_ptrs, _u32(nof_ports), _u32(nof_subc), _i(with_writer_access), _u32(3),
                                    _p(init), _p(got), _p(want))
    assert rc == n, rc
    if with_writer_access:
        assert np.array_equal(got, want)
    else:
        # through the precoder a -0.0 may come out as +0.0: compare values, and bits wherever neither is a zero
        a, b = backends.bf16_to_f32(got), backends.bf16_to_f32(want)
        assert np.array_equal(a, b)
        assert np.all(a[got != want] == 0)   # the only bit differences: signed zeros


def test_pdsch_validator_adaptor(harness, ref):
    """pdsch_pdu_validator_adaptor agrees with the reference's validator on valid and broken PDUs."""
    rng = np.random.default_rng(32)
    harness.adaptor_test_pdsch_validator.restype = _i
    pdus = cases.unit_test_like_pdus(rng) + [p for p, _, _ in cases.random_pdus(backends.oracle().tbs, rng, 20)]
    for pdu in pdus:
        for field, value in ((None, 0), ("nof_codewords", 2), ("dmrs_type", 2), ("nof_symbols", 15), ("rv", 1)):
            q = abi.PdschPdu.from_buffer_copy(pdu)
            q._keepalive = pdu._keepalive
            if field:
                setattr(q, field, value)
            assert harness.adaptor_test_pdsch_validator(C.byref(q)) == (1 if ref.validate(q) == 0 else 0), field


def test_pdsch_encoder_hw_adaptor_through_the_references_hal_client(harness):
    """hal::hw_accelerator_pdsch_enc adaptor driven by the reference's pdsch_encoder_hw_impl (configure / enqueue /
    dequeue, transport-block mode) equals pdsch_encoder_impl bit for bit."""
    rng = np.random.default_rng(33)
    harness.adaptor_test_pdsch_encoder_hw.restype = _i
    for bg, rv, qm, layers, nre, tb_bytes in ((1, 0, 8, 4, 2700, 8000), (2, 2, 2, 1, 600, 40), (1, 3, 6, 2, 5000, 3000),
                                              (1, 0, 4, 3, 999, 1200), (2, 1, 2, 1, 288, 100)):
        tb = rng.integers(0, 256, tb_bytes, dtype=np.uint8)
        cw = nre * layers * qm
        a, b = np.zeros(cw, np.uint8), np.zeros(cw, np.uint8)
        n = harness.adaptor_test_pdsch_encoder_hw(_u32(bg), _u32(rv), _u32(qm), _u32(0), _u32(layers), _u32(nre * layers), _p(tb),
                                                  _u32(tb_bytes), _p(a), _p(b))
        assert n == cw and np.array_equal(a, b), (bg, rv, qm, layers)


def test_ofdm_modulator_adaptors(harness):
    """ofdm_symbol_modulator_adaptor in the real-time loop's order (every port of a symbol, then the next symbol: the slot
    cache) and ofdm_slot_modulator_adaptor against ofdm_slot_modulator_impl."""
    rng = np.random.default_rng(34)
    harness.adaptor_test_ofdm.restype = _i
    for mu, bw, n, ext, ports, slot in ((1, 51, 2048, 0, 2, 1), (0, 52, 1024, 0, 1, 0), (2, 24, 512, 1, 2, 3)):
        cfg = abi.OfdmConfig(mu, bw, n, ext, 0.37, 3.5e9)
        grid = random_grid(rng, ports, 14, bw * 12)
        size = backends.pkg.lib.slot_size(cfg, slot)
        by_symbol, by_slot, want = (np.zeros((ports, size), np.complex64) for _ in range(3))
        rc = harness.adaptor_test_ofdm(C.byref(cfg), _p(grid), _u32(ports), _u32(slot), _p(by_symbol), _p(by_slot), _p(want))
        assert rc == size
        scale = np.abs(want).max()
        assert np.abs(by_symbol - want).max() / scale < 1e-5 and np.array_equal(by_symbol, by_slot)


@pytest.mark.parametrize("with_writer_access", [1, 0])
def test_csi_rs_and_pdcch_adaptors(harness, with_writer_access):
    rng = np.random.default_rng(35)
    for fn in ("adaptor_test_csi_rs", "adaptor_test_pdcch"):
        getattr(harness, fn).restype = _i
    for name, cfg, nof_ports, nof_subc in list(cases.csi_rs_cases(rng))[::3]:
        init = random_grid(rng, nof_ports, 14, nof_subc)
        got, want = np.zeros_like(init), np.zeros_like(init)
        assert harness.adaptor_test_csi_rs(C.byref(cfg), _u32(nof_ports), _u32(nof_subc), _i(with_writer_access), _p(init), _p(got), _p(want)) == 0
        assert np.array_equal(backends.bf16_to_f32(got), backends.bf16_to_f32(want)), name
        assert np.array_equal(got, want) if with_writer_access else np.all(backends.bf16_to_f32(got)[got != want] == 0), name
    for i in range(25):
        pdu = cases.random_pdcch(rng)
        init = random_grid(rng, 4, 14, 624)
        got, want = np.zeros_like(init), np.zeros_like(init)
        assert harness.adaptor_test_pdcch(C.byref(pdu), _u32(4), _u32(624), _i(with_writer_access), _p(init), _p(got), _p(want)) == 0
        assert np.array_equal(backends.bf16_to_f32(got), backends.bf16_to_f32(want)), i
        assert np.array_equal(got, want) if with_writer_access else np.all(backends.bf16_to_f32(got)[got != want] == 0), i


def test_ssb_amplitude_and_ofh_adaptors(harness):
    rng = np.random.default_rng(36)
    for fn in ("adaptor_test_ssb", "adaptor_test_amplitude", "adaptor_test_ofh"):
        getattr(harness, fn).restype = _i
    for i in range(15):
        pdu = cases.random_ssb(rng, nof_ports=3)
        init = random_grid(rng, 3, 14, 624)
        got, want = np.zeros_like(init), np.zeros_like(init)
        assert harness.adaptor_test_ssb(C.byref(pdu), _u32(3), _u32(624), _p(init), _p(got), _p(want)) == 0
        assert np.array_equal(got, want), i
    x = ((rng.standard_normal((5, 2000)) + 1j * rng.standard_normal((5, 2000))) * 0.5).astype(np.complex64)
    ya, yr = np.zeros_like(x), np.zeros_like(x)
    ma, mr = np.zeros(8), np.zeros(8)
    assert harness.adaptor_test_amplitude(_i(1), C.c_float(-1.0), C.c_float(1.0), C.c_float(-6.0), _p(x), _u32(2000), _u32(5), _p(ya),
                                          _p(yr), _p(ma), _p(mr)) == 0
    assert np.array_equal(ya.view(np.uint32), yr.view(np.uint32))
    assert np.array_equal(ma[4:], mr[4:]) and np.allclose(ma[:4], mr[:4], rtol=2e-5)   # counters exactly, powers to float accuracy
    for typ, w in ((0, 9), (0, 16), (0, 12), (1, 9), (1, 14), (1, 8)):
        nprb = int(rng.integers(1, 60))
        prbs = ((rng.standard_normal((nprb, 12, 2)) * 0.25).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        a, r = np.zeros(nprb * 49, np.uint8), np.zeros(nprb * 49, np.uint8)
        n = harness.adaptor_test_ofh(_i(typ), _u32(w), C.c_float(0.9), _p(prbs), _u32(nprb), _p(a), _p(r))
        assert n == nprb * (3 * w + typ) and np.array_equal(a[:n], r[:n]), (typ, w)


@pytest.mark.parametrize("modulation", [0, 1, 2, 4, 6, 8])
def test_demodulation_mapper_adaptor(harness, modulation):
    """demodulation_mapper_adaptor against demodulation_mapper_impl on spans around the vector batch sizes."""
    rng = np.random.default_rng(600 + modulation)
    qm = max(modulation, 1)
    for n in (1, 5, 16, 37, 1003):
        for kind in (0, 1, 2):
            sym, noise = cases.demod_inputs(rng, modulation, n, kind)
            got, want = np.zeros(n * qm, np.int8), np.full(n * qm, 99, np.int8)
            harness.adaptor_test_demod.restype = _i
            assert harness.adaptor_test_demod(_u32(modulation), _u32(n), _p(sym), _p(noise), _p(got), _p(want)) == 0
            assert np.array_equal(got, want), (modulation, n, kind)


def test_fapi_pdsch_shim(harness):
    """fapi_to_pod equals convert_pdsch_fapi_to_phy followed by to_pod (every POD byte and weight), for both reference
    points, both resource allocation types, several transmission types and power profiles, with and without a CSI-RS
    rate-matching pattern; and fapi_pdsch_slot_batch (all PDUs of the slot in one call) leaves the slot's grid as the
    reference's processor does PDU by PDU."""
    rng = np.random.default_rng(808)
    nof_ports, nof_subc = 4, 106 * 12
    # rnti, bwp_start, bwp_size, qm, rv, nid, dmrs mask, scr id, nscid, cdm groups, rb_start, rb_size, start symbol, nof symbols,
    # power offset, layers | ref point, type-1 allocation, trans_type, ss profile, csi pattern
    rows = np.array([
        [0x4601, 0, 106, 2, 0, 11, 0b000100000100, 5, 0, 2, 0, 20, 2, 12, 0, 1, 0, 1, 0, 1, 1],
        [0x4602, 0, 106, 4, 1, 12, 0b000100000100, 6, 1, 2, 20, 30, 2, 12, 3, 2, 0, 0, 0, 2, 0],
        [0x4603, 0, 106, 6, 2, 13, 0b100000000100, 7, 0, 1, 50, 25, 1, 13, -2, 3, 0, 1, 0, 0, 1],
        [0x4604, 0, 106, 8, 3, 14, 0b000100000100, 8, 1, 2, 75, 31, 2, 12, 8, 4, 1, 0, 0, 3, 0],
    ], dtype=np.int32)
    n = rows.shape[0]
    tb_sizes = np.array([160, 700, 2100, 9000], dtype=np.uint32)
    tbs = [rng.integers(0, 256, int(s), dtype=np.uint8) for s in tb_sizes]
    tb_ptrs = (C.c_void_p * n)(*[t.ctypes.data for t in tbs])
    init = random_grid(rng, nof_ports, 14, nof_subc)
    got, want = np.zeros_like(init), np.zeros_like(init)
    harness.adaptor_test_fapi.restype = _i
    rc = harness.adaptor_test_fapi(_u32(n), _p(rows), tb_ptrs, _p(tb_sizes), _u32(nof_ports), _u32(nof_subc), _p(init), _p(got), _p(want))
    assert rc == 0, rc
    assert not np.array_equal(want, init)
    assert np.array_equal(got, want)


def _dl_pipeline_inputs(rng, n_slots, nof_ports, nof_rb):
    """Per slot one PDCCH and one PDSCH PDU (other transport block, RNTI, slot index each), an SS/PBCH block and a CSI-RS."""
    w = cases.codebook("two_layer_two_ports_0")
    pdsch, tbs, pdcch = [], [], []
    for i in range(n_slots):
        tb_bits = cases.tbs(12, 12, 4, 490, 2, 30)
        p = abi.make_pdu(bwp_size_rb=nof_rb, qm=4, rnti=17 + i, n_id=5, dmrs_symbols=(2, 11), prb_start=22, prb_count=30,
                         start_symbol=2, nof_symbols=12, precoding=w, tb_size_bytes=tb_bits // 8, slot_index=i)
        pdsch.append(p)
        tbs.append(cases.random_tb(rng, p))
        pdcch.append(abi.make_pdcch(payload=rng.integers(0, 2, 41, dtype=np.uint8), rnti=17 + i, cce_index=0, aggregation_level=4,
                                    duration=2, frequency_resources=tuple(range(8)), mapping="interleaved", reg_bundle_size=6,
                                    interleaver_size=2, shift_index=3, n_id_dmrs=5, n_id_data=5, n_rnti=17 + i, bwp_size_rb=nof_rb,
                                    slot_index=i, precoding=np.array([[1.0, 1.0j]], np.complex64) / np.sqrt(2)))
    ssb = abi.make_ssb(pattern_case="A", ssb_idx=0, L_max=4, phys_cell_id=5, payload=rng.integers(0, 2, 32, dtype=np.uint8),
                       sfn=0, ports=(0,))
    csi = abi.make_csi_rs(row=3, start_rb=0, nof_rb=nof_rb, k0=4, l0=13, density="one", scrambling_id=5,
                          precoding=np.eye(2, dtype=np.complex64)[None])
    return pdsch, tbs, pdcch, ssb, csi


def _run_dl_pipeline(harness, rng, n_slots, mirrored, max_wait_us, settle_ms):
    nof_ports, nof_rb = 2, 52
    nof_subc = 12 * nof_rb
    cfg = abi.OfdmConfig(1, nof_rb, 1024, 0, 1.0, 2.4e9)     # 30 kHz: two slots per subframe, so the slot index matters
    pdsch, tbs, pdcch, ssb, csi = _dl_pipeline_inputs(rng, n_slots, nof_ports, nof_rb)
    arr = (abi.PdschPdu * n_slots)(*pdsch)
    cch = (abi.PdcchPdu * n_slots)(*pdcch)
    tb_ptrs = (C.c_void_p * n_slots)(*[t.ctypes.data for t in tbs])
    stride = backends.pkg.lib.slot_size(cfg, 0)
    grids = [np.zeros((n_slots, nof_ports, 14, nof_subc, 2), np.uint16) for _ in range(2)]
    iqs = [np.zeros((n_slots, nof_ports, stride), np.complex64) for _ in range(2)]
    info = np.zeros(10, np.int32)
    harness.adaptor_test_dl_pipeline.restype = _i
    rc = harness.adaptor_test_dl_pipeline(_u32(n_slots), C.byref(cfg), _u32(nof_ports), arr, tb_ptrs, cch, C.byref(ssb), C.byref(csi),
                                          _i(mirrored), _u32(max_wait_us), _u32(settle_ms), _p(grids[0]), _p(grids[1]), _p(iqs[0]),
                                          _p(iqs[1]), _p(info))
    assert rc == 0
    return grids, iqs, info


@pytest.mark.parametrize("mirrored", [1, 0])
def test_dl_slot_pipeline_adaptors_against_the_references_lower_phy_loop(harness, mirrored):
    """The compiled reference's lower-PHY loop -- handle_request per slot, then process_symbol 14 x per slot with a
    baseband buffer of every transmit port (downlink_processor_baseband_impl.cpp:224-235) -- on pdxch_processor_adaptor next
    to pdxch_processor_impl.  Device-mirrored grids (mirrored = 1): the channel processor adaptors write the slot's device
    grid (the PDSCH adaptor acknowledges at once), the host layer's few resource elements join them at hand-over through ONE
    sparse put, nothing is loaded from or read back to the host until the test itself reads the grids; plain grids
    (mirrored = 0): one grid load per slot.  Either way every process_symbol is served from the slot's finished IQ: the
    same samples as the reference's modulator to 1e-5, no late notification, and no call longer than a copy."""
    rng = np.random.default_rng(41)
    n_slots = 3
    grids, iqs, info = _run_dl_pipeline(harness, rng, n_slots, mirrored, 0, 40)
    late_a, late_r, proc_a, proc_r, longest_ns, synchronous, n_mod, n_load, n_read, n_put = (int(v) for v in info)
    assert (late_a, late_r) == (0, 0) and proc_a == proc_r == 14 * n_slots
    assert np.array_equal(grids[0], grids[1]) if mirrored else np.array_equal(backends.bf16_to_f32(grids[0]), backends.bf16_to_f32(grids[1]))
    scale = np.abs(iqs[1]).max()
    assert scale > 0 and np.abs(iqs[0] - iqs[1]).max() / scale < 1e-5
    assert n_mod == n_slots
    if mirrored:
        assert synchronous == n_slots and n_load == 0 and n_put == n_slots and n_read == n_slots   # (the reads: this test's own)
    else:
        assert synchronous == 0 and n_load == n_slots and n_put == 0 and n_read == 0
    # a symbol of two ports is 2 x 1096 x 8 bytes: microseconds; the bound is loose because the container's clock is noisy
    assert longest_ns < 2_000_000, longest_ns


def test_dl_slot_pipeline_device_grid_filled_on_the_host_goes_down_as_one_copy(harness):
    """A device-mirrored grid that only the REFERENCE's processors wrote (through its mapper / writer: everything is in the host
    layer): at hand-over the dense layer goes to the slot as ONE grid copy (nrphy_dl_slot_load_grid) instead of a sparse put of
    most of the grid; same grid, same IQ as the reference's own lower PHY."""
    rng = np.random.default_rng(43)
    n_slots = 2
    grids, iqs, info = _run_dl_pipeline(harness, rng, n_slots, 2, 0, 40)
    late_a, late_r, proc_a, proc_r, longest_ns, synchronous, n_mod, n_load, n_read, n_put = (int(v) for v in info)
    assert (late_a, late_r) == (0, 0) and proc_a == proc_r == 14 * n_slots
    assert np.array_equal(grids[0], grids[1])
    scale = np.abs(iqs[1]).max()
    assert scale > 0 and np.abs(iqs[0] - iqs[1]).max() / scale < 1e-5
    assert n_mod == n_slots and n_load == n_slots and n_put == 0


def test_dl_slot_pipeline_adaptor_late_slot_and_bounded_wait(harness):
    """A slot whose IQ has not arrived when its first symbol is due: with max_wait_us = 0 the real-time call does not wait --
    on_pdxch_request_late, silence for the slot (pdxch_processor_impl.cpp:65-74 does the same for a request that misses its
    slot); with a bounded wait that covers the (mock's 3 ms) modulation the slot is transmitted."""
    rng = np.random.default_rng(42)
    grids, iqs, info = _run_dl_pipeline(harness, rng, 1, 1, 0, 0)
    assert int(info[0]) == 1 and int(info[2]) == 0 and int(info[3]) == 14 and not iqs[0].any()
    grids, iqs, info = _run_dl_pipeline(harness, rng, 1, 1, 50000, 0)
    assert int(info[0]) == 0 and int(info[2]) == 14
    assert np.abs(iqs[0] - iqs[1]).max() / np.abs(iqs[1]).max() < 1e-5
