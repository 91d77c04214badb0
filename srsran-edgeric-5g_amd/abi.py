"""ctypes mirror of include/mi355_nrphy.h (the C ABI) plus PDU construction helpers.

The structures here are the POD mirror of the reference's ``pdsch_processor::pdu_t``
(srsRAN-5G-ER/include/srsran/phy/upper/channel_processors/pdsch_processor.h:58-155) and
``ofdm_modulator_configuration`` (include/srsran/phy/lower/modulation/ofdm_modulator.h:34-47).
Pure host-side plumbing: no compute happens in this module.
"""
import ctypes as C

import numpy as np

MAX_RB = 275
NRE = 12
NSYMB = 14
MAX_PORTS = 4
MAX_LAYERS = 4
PRB_WORDS = 5
MAX_RESERVED = 4
MAX_CODEBLOCKS = 162

OK = 0
ERR_INVALID_PDU = 1
ERR_ARGUMENT = 2
ERR_DEVICE = 3
ERR_CAPACITY = 4
ERR_NOT_READY = 5

TBS_LBRM_DEFAULT = 159749  # tbs_lbrm_default, include/srsran/ran/sch/sch_constants.h:47


class RePattern(C.Structure):
    _fields_ = [
        ("prb_mask", C.c_uint64 * PRB_WORDS),
        ("re_mask", C.c_uint16),
        ("symbol_mask", C.c_uint16),
        ("reserved_", C.c_uint32),
    ]


class PdschPdu(C.Structure):
    _fields_ = [
        ("slot_index", C.c_uint32),
        ("rnti", C.c_uint32),
        ("bwp_start_rb", C.c_uint32),
        ("bwp_size_rb", C.c_uint32),
        ("cp", C.c_uint32),
        ("qm", C.c_uint32),
        ("rv", C.c_uint32),
        ("nof_codewords", C.c_uint32),
        ("n_id", C.c_uint32),
        ("ref_point", C.c_uint32),
        ("dmrs_symbol_mask", C.c_uint32),
        ("dmrs_type", C.c_uint32),
        ("scrambling_id", C.c_uint32),
        ("n_scid", C.c_uint32),
        ("nof_cdm_groups_without_data", C.c_uint32),
        ("start_symbol_index", C.c_uint32),
        ("nof_symbols", C.c_uint32),
        ("ldpc_base_graph", C.c_uint32),
        ("tbs_lbrm_bytes", C.c_uint32),
        ("vrb_contiguous", C.c_uint32),
        ("prb_mask", C.c_uint64 * PRB_WORDS),
        ("nof_reserved", C.c_uint32),
        ("tb_size_bytes", C.c_uint32),
        ("reserved", RePattern * MAX_RESERVED),
        ("ratio_pdsch_dmrs_to_sss_dB", C.c_float),
        ("ratio_pdsch_data_to_sss_dB", C.c_float),
        ("nof_layers", C.c_uint32),
        ("nof_ports", C.c_uint32),
        ("prg_size_rb", C.c_uint32),
        ("nof_prg", C.c_uint32),
        ("precoding", C.POINTER(C.c_float)),
    ]


class PdschDerived(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "nof_re", "nof_codeblocks", "lifting_size", "segment_length", "cb_info_bits", "nof_filler_bits",
        "nof_tb_crc_bits", "nof_cb_crc_bits", "zero_pad", "full_length", "n_ref", "n_cb", "k0",
        "nof_short_segments", "rm_length_short", "rm_length_long", "codeword_bits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class PdschEncoderCfg(C.Structure):
    """nrphy_pdsch_encoder_cfg_t (pdsch_encoder::configuration + TB size)."""
    _fields_ = [("base_graph", C.c_uint32), ("rv", C.c_uint32), ("qm", C.c_uint32), ("nref", C.c_uint32),
                ("nof_layers", C.c_uint32), ("nof_ch_symbols", C.c_uint32), ("tb_size_bytes", C.c_uint32)]


class LdpcDecoderCfg(C.Structure):
    """nrphy_ldpc_decoder_cfg_t (ldpc_decoder::configuration + number of soft bits)."""
    _fields_ = [("base_graph", C.c_uint32), ("lifting_size", C.c_uint32), ("nof_filler_bits", C.c_uint32),
                ("crc_poly", C.c_uint32), ("nof_llr", C.c_uint32), ("max_iterations", C.c_uint32),
                ("scaling_factor", C.c_float)]


class LdpcRateDematcherCfg(C.Structure):
    """nrphy_ldpc_rate_dematcher_cfg_t (the codeblock_metadata fields the rate dematcher reads + input length)."""
    _fields_ = [("base_graph", C.c_uint32), ("lifting_size", C.c_uint32), ("rv", C.c_uint32), ("qm", C.c_uint32),
                ("nref", C.c_uint32), ("nof_filler_bits", C.c_uint32), ("rm_length", C.c_uint32)]


class PuschDecoderCfg(C.Structure):
    """nrphy_pusch_decoder_cfg_t (pusch_decoder::configuration + TB size + number of channel symbols)."""
    _fields_ = [("base_graph", C.c_uint32), ("qm", C.c_uint32), ("rv", C.c_uint32), ("nof_layers", C.c_uint32),
                ("nref", C.c_uint32), ("tb_size_bytes", C.c_uint32), ("nof_ch_symbols", C.c_uint32),
                ("max_iterations", C.c_uint32), ("use_early_stop", C.c_uint32), ("new_data", C.c_uint32)]


class GridRe(C.Structure):
    """nrphy_grid_re_t: one resource element written from the host into a device grid."""
    _fields_ = [("port", C.c_uint16), ("symbol", C.c_uint16), ("subc", C.c_uint32), ("value", C.c_uint32)]


class CsiRsCfg(C.Structure):
    """nrphy_csi_rs_cfg_t (nzp_csi_rs_generator::config_t)."""
    _fields_ = [("slot_index", C.c_uint32), ("cp", C.c_uint32), ("start_rb", C.c_uint32), ("nof_rb", C.c_uint32),
                ("row", C.c_uint32), ("nof_k_ref", C.c_uint32), ("k_ref", C.c_uint32 * 6), ("symbol_l0", C.c_uint32),
                ("symbol_l1", C.c_uint32), ("cdm", C.c_uint32), ("density", C.c_uint32), ("scrambling_id", C.c_uint32),
                ("amplitude", C.c_float), ("nof_ports", C.c_uint32), ("prg_size_rb", C.c_uint32), ("nof_prg", C.c_uint32),
                ("precoding", C.POINTER(C.c_float))]


PDSCH_DONE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)   # nrphy_pdsch_done_fn

CSI_DENSITY = {"dot5_even": 0, "dot5_odd": 1, "one": 2, "three": 3}
CSI_ROW_PORTS = {1: 1, 2: 1, 3: 2, 4: 4, 5: 4}


def make_csi_rs(*, row, start_rb, nof_rb, k0, l0, density, slot_index=0, cp=0, scrambling_id=0, amplitude=1.0,
                precoding=None, prg_size_rb=MAX_RB):
    """Builds a CsiRsCfg for rows 1-5; ``precoding`` [nof_prg][ports][ports] complex64 (default identity).  The weight
    array is attached to the struct (``_keepalive``)."""
    ports = CSI_ROW_PORTS[row]
    if precoding is None:
        precoding = np.eye(ports, dtype=np.complex64)[None]
    w = np.ascontiguousarray(np.asarray(precoding, dtype=np.complex64)).view(np.float32).reshape(-1)
    c = CsiRsCfg()
    c.slot_index, c.cp, c.start_rb, c.nof_rb, c.row = slot_index, cp, start_rb, nof_rb, row
    c.nof_k_ref, c.symbol_l0, c.symbol_l1 = 1, l0, 0
    c.k_ref[0] = k0
    c.cdm = 0 if row in (1, 2) else 1
    c.density = CSI_DENSITY[density]
    c.scrambling_id, c.amplitude, c.nof_ports = scrambling_id, amplitude, ports
    c.prg_size_rb, c.nof_prg = prg_size_rb, np.asarray(precoding).shape[0]
    c._keepalive = w
    c.precoding = w.ctypes.data_as(C.POINTER(C.c_float))
    return c


PDCCH_MAX_PAYLOAD = 128


class PdcchPdu(C.Structure):
    """nrphy_pdcch_pdu_t (pdcch_processor::pdu_t: coreset_description + dci_description)."""
    _fields_ = [("slot_index", C.c_uint32), ("cp", C.c_uint32), ("bwp_size_rb", C.c_uint32), ("bwp_start_rb", C.c_uint32),
                ("start_symbol_index", C.c_uint32), ("duration", C.c_uint32), ("frequency_resources", C.c_uint64),
                ("cce_to_reg_mapping", C.c_uint32), ("reg_bundle_size", C.c_uint32), ("interleaver_size", C.c_uint32),
                ("shift_index", C.c_uint32), ("rnti", C.c_uint32), ("n_id_pdcch_dmrs", C.c_uint32),
                ("n_id_pdcch_data", C.c_uint32), ("n_rnti", C.c_uint32), ("cce_index", C.c_uint32),
                ("aggregation_level", C.c_uint32), ("dmrs_power_offset_dB", C.c_float), ("data_power_offset_dB", C.c_float),
                ("payload_size", C.c_uint32), ("payload", C.c_uint8 * PDCCH_MAX_PAYLOAD), ("nof_ports", C.c_uint32),
                ("prg_size_rb", C.c_uint32), ("nof_prg", C.c_uint32), ("precoding", C.POINTER(C.c_float))]


CCE_TO_REG = {"coreset0": 0, "non_interleaved": 1, "interleaved": 2}


def make_pdcch(*, payload, rnti, cce_index, aggregation_level, duration, frequency_resources=(), mapping="non_interleaved",
               bwp_start_rb=0, bwp_size_rb=52, start_symbol=0, reg_bundle_size=6, interleaver_size=2, shift_index=0,
               n_id_dmrs=0, n_id_data=0, n_rnti=0, dmrs_dB=0.0, data_dB=0.0, slot_index=0, cp=0, precoding=None,
               prg_size_rb=MAX_RB):
    """Builds a PdcchPdu; ``payload`` = DCI bits, ``frequency_resources`` = indices of the 6-PRB groups of the CORESET,
    ``precoding`` [nof_prg][nof_ports] complex64 (default: one port, weight 1)."""
    p = PdcchPdu()
    p.slot_index, p.cp, p.bwp_size_rb, p.bwp_start_rb = slot_index, cp, bwp_size_rb, bwp_start_rb
    p.start_symbol_index, p.duration = start_symbol, duration
    p.frequency_resources = sum(1 << i for i in frequency_resources)
    p.cce_to_reg_mapping = CCE_TO_REG[mapping]
    p.reg_bundle_size, p.interleaver_size, p.shift_index = reg_bundle_size, interleaver_size, shift_index
    p.rnti, p.n_id_pdcch_dmrs, p.n_id_pdcch_data, p.n_rnti = rnti, n_id_dmrs, n_id_data, n_rnti
    p.cce_index, p.aggregation_level = cce_index, aggregation_level
    p.dmrs_power_offset_dB, p.data_power_offset_dB = dmrs_dB, data_dB
    payload = np.asarray(payload, dtype=np.uint8)
    p.payload_size = payload.size
    for i, b in enumerate(payload[:PDCCH_MAX_PAYLOAD]):
        p.payload[i] = int(b)
    if precoding is None:
        precoding = np.ones((1, 1), np.complex64)
    w = np.ascontiguousarray(np.asarray(precoding, dtype=np.complex64))
    assert w.ndim == 2
    p.nof_prg, p.nof_ports, p.prg_size_rb = w.shape[0], w.shape[1], prg_size_rb
    f = w.view(np.float32).reshape(-1)
    p._keepalive = f
    p.precoding = f.ctypes.data_as(C.POINTER(C.c_float))
    return p


class SsbPdu(C.Structure):
    """nrphy_ssb_pdu_t (ssb_processor::pdu_t)."""
    _fields_ = [("numerology", C.c_uint32), ("sfn", C.c_uint32), ("slot_index", C.c_uint32), ("phys_cell_id", C.c_uint32),
                ("beta_pss_dB", C.c_float), ("ssb_idx", C.c_uint32), ("L_max", C.c_uint32), ("common_scs", C.c_uint32),
                ("subcarrier_offset", C.c_uint32), ("offset_to_pointA", C.c_uint32), ("pattern_case", C.c_uint32),
                ("bch_payload", C.c_uint8 * 32), ("nof_ports", C.c_uint32), ("ports", C.c_uint8 * MAX_PORTS)]


SSB_CASE = {"A": 0, "B": 1, "C": 2, "D": 3, "E": 4}


def make_ssb(*, pattern_case, ssb_idx, L_max, phys_cell_id, payload, sfn=0, numerology=None, slot_index=None, common_scs=None,
             subcarrier_offset=0, offset_to_pointA=0, beta_pss_dB=0.0, ports=(0,)):
    """Builds an SsbPdu; the slot defaults to the one of the first half frame that carries candidate ``ssb_idx``."""
    case = SSB_CASE[pattern_case]
    mu = {0: 0, 1: 1, 2: 1, 3: 3, 4: 4}[case] if numerology is None else numerology
    first = {0: lambda i: (2, 8)[i % 2] + 14 * (i // 2), 1: lambda i: (4, 8, 16, 20)[i % 4] + 28 * (i // 4),
             2: lambda i: (2, 8)[i % 2] + 14 * (i // 2)}.get(case)
    p = SsbPdu()
    p.numerology, p.sfn = mu, sfn
    p.slot_index = (first(ssb_idx) // 14 if first else 0) if slot_index is None else slot_index
    p.phys_cell_id, p.beta_pss_dB, p.ssb_idx, p.L_max = phys_cell_id, beta_pss_dB, ssb_idx, L_max
    p.common_scs = (mu if mu < 4 else 3) if common_scs is None else common_scs
    p.subcarrier_offset, p.offset_to_pointA, p.pattern_case = subcarrier_offset, offset_to_pointA, case
    payload = np.asarray(payload, dtype=np.uint8)
    for i in range(32):
        p.bch_payload[i] = int(payload[i]) if i < payload.size else 0
    p.nof_ports = len(ports)
    for i, q in enumerate(ports):
        p.ports[i] = q
    return p


# NRPHY_MOD_*: bits per symbol, 0 for pi/2-BPSK
MOD_PI2_BPSK, MOD_BPSK, MOD_QPSK, MOD_QAM16, MOD_QAM64, MOD_QAM256 = 0, 1, 2, 4, 6, 8


class AmplitudeCfg(C.Structure):
    """nrphy_amplitude_cfg_t (the constructor arguments of amplitude_controller_{clipping,scaling}_impl)."""
    _fields_ = [("kind", C.c_uint32), ("enable_clipping", C.c_uint32), ("input_gain_dB", C.c_float),
                ("full_scale_lin", C.c_float), ("ceiling_dBFS", C.c_float)]


class AmplitudeStats(C.Structure):
    _fields_ = [("sum_power", C.c_float), ("peak_power", C.c_float), ("nof_clipped", C.c_uint32), ("nof_samples", C.c_uint32)]


class AmplitudeMetrics(C.Structure):
    """nrphy_amplitude_metrics_t (amplitude_controller_metrics)."""
    _fields_ = [("avg_power_fs", C.c_float), ("peak_power_fs", C.c_float), ("papr_lin", C.c_float), ("gain_dB", C.c_float),
                ("nof_processed_samples", C.c_uint64), ("nof_clipped_samples", C.c_uint64),
                ("clipping_probability", C.c_double), ("clipping_enabled", C.c_uint32), ("reserved_", C.c_uint32)]


class IqWireCfg(C.Structure):
    _fields_ = [("amplitude", AmplitudeCfg), ("ci16_scale", C.c_float)]


class DlSlotsCfg(C.Structure):
    """nrphy_dl_slots_cfg_t: the downlink slot pipeline (seams A and C on one device-resident grid)."""
    # (_fields_ set below OfdmConfig)


DL_SLOT_DONE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_uint32)   # nrphy_dl_slot_done_fn


class OfhCompressionCfg(C.Structure):
    """nrphy_ofh_compression_cfg_t (ru_compression_params + the compressor's iq_scaling)."""
    _fields_ = [("type", C.c_uint32), ("data_width", C.c_uint32), ("iq_scaling", C.c_float)]


class OfdmConfig(C.Structure):
    _fields_ = [
        ("numerology", C.c_uint32),
        ("bw_rb", C.c_uint32),
        ("dft_size", C.c_uint32),
        ("cp", C.c_uint32),
        ("scale", C.c_float),
        ("center_freq_hz", C.c_double),
    ]


DlSlotsCfg._fields_ = [("ofdm", OfdmConfig), ("nof_ports", C.c_uint32), ("depth", C.c_uint32), ("max_tb_bytes", C.c_uint32),
                       ("iq_format", C.c_uint32), ("wire", IqWireCfg)]


def prb_mask_words(prbs):
    """Bit mask words (5 x uint64) with the given PRB indices set."""
    words = [0] * PRB_WORDS
    for p in prbs:
        words[p // 64] |= 1 << (p % 64)
    return words


def bits_to_mask(bits):
    m = 0
    for i, b in enumerate(bits):
        if b:
            m |= 1 << i
    return m


def identity_precoding(nof_layers):
    """precoding_configuration::make_wideband(make_identity(n)) -- [1][n][n] complex."""
    w = np.zeros((1, nof_layers, nof_layers, 2), dtype=np.float32)
    for i in range(nof_layers):
        w[0, i, i, 0] = 1.0
    return w


def make_pdu(*, slot_index=0, rnti=1, bwp_start_rb=0, bwp_size_rb=52, qm=2, rv=0, n_id=0, ref_point=0,
             dmrs_symbols=(2,), dmrs_type=1, scrambling_id=0, n_scid=0, nof_cdm_groups_without_data=2,
             prb_start=0, prb_count=52, prbs=None, start_symbol=0, nof_symbols=14, base_graph=1,
             tbs_lbrm_bytes=TBS_LBRM_DEFAULT, reserved=(), ratio_dmrs_dB=0.0, ratio_data_dB=0.0,
             precoding=None, prg_size_rb=MAX_RB, tb_size_bytes=0, nof_codewords=1, cp=0, vrb_contiguous=None):
    """Builds a PdschPdu.  ``precoding`` is an array [nof_prg][nof_ports][nof_layers] complex64 or
    [...][2] float32; ``reserved`` a sequence of (prbs, re_bits[12], symbol_bits[14]).  The numpy weight
    array is attached to the returned struct (``_keepalive``) so the pointer stays valid."""
    pdu = PdschPdu()
    pdu.slot_index = slot_index
    pdu.rnti = rnti
    pdu.bwp_start_rb = bwp_start_rb
    pdu.bwp_size_rb = bwp_size_rb
    pdu.cp = cp
    pdu.qm = qm
    pdu.rv = rv
    pdu.nof_codewords = nof_codewords
    pdu.n_id = n_id
    pdu.ref_point = ref_point
    pdu.dmrs_symbol_mask = sum(1 << l for l in dmrs_symbols)
    pdu.dmrs_type = dmrs_type
    pdu.scrambling_id = scrambling_id
    pdu.n_scid = n_scid
    pdu.nof_cdm_groups_without_data = nof_cdm_groups_without_data
    pdu.start_symbol_index = start_symbol
    pdu.nof_symbols = nof_symbols
    pdu.ldpc_base_graph = base_graph
    pdu.tbs_lbrm_bytes = tbs_lbrm_bytes
    if prbs is None:
        prbs = list(range(prb_start, prb_start + prb_count))
    prbs = sorted(prbs)
    contiguous = len(prbs) > 0 and prbs[-1] - prbs[0] + 1 == len(prbs)
    pdu.vrb_contiguous = int(contiguous if vrb_contiguous is None else vrb_contiguous)
    for i, w in enumerate(prb_mask_words(prbs)):
        pdu.prb_mask[i] = w
    pdu.nof_reserved = len(reserved)
    for i, (rprbs, re_bits, sym_bits) in enumerate(reserved):
        for j, w in enumerate(prb_mask_words(rprbs)):
            pdu.reserved[i].prb_mask[j] = w
        pdu.reserved[i].re_mask = bits_to_mask(re_bits)
        pdu.reserved[i].symbol_mask = bits_to_mask(sym_bits)
    pdu.tb_size_bytes = tb_size_bytes
    pdu.ratio_pdsch_dmrs_to_sss_dB = ratio_dmrs_dB
    pdu.ratio_pdsch_data_to_sss_dB = ratio_data_dB
    if precoding is None:
        precoding = identity_precoding(1)
    w = np.asarray(precoding)
    if np.iscomplexobj(w):
        w = np.stack([w.real, w.imag], axis=-1)
    w = np.ascontiguousarray(w, dtype=np.float32)
    assert w.ndim == 4 and w.shape[3] == 2, "precoding must be [nof_prg][nof_ports][nof_layers] complex"
    pdu.nof_prg, pdu.nof_ports, pdu.nof_layers = w.shape[0], w.shape[1], w.shape[2]
    pdu.prg_size_rb = prg_size_rb
    pdu.precoding = w.ctypes.data_as(C.POINTER(C.c_float))
    pdu._keepalive = w
    return pdu


def declare(lib, prefix="nrphy_"):
    """Attaches argtypes/restypes for the entry points of include/mi355_nrphy.h found in ``lib``.
    The same signatures (with prefix ``oracle_``) are exported by the CPU oracle."""
    P = C.POINTER
    vp, u8p, u32, u64, i32 = C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_int

    def sig(name, restype, *argtypes):
        fn = getattr(lib, prefix + name, None)
        if fn is not None:
            fn.restype = restype
            fn.argtypes = list(argtypes)

    sig("version", C.c_char_p)
    sig("strerror", C.c_char_p, i32)
    sig("create", i32, P(vp), i32)
    sig("destroy", i32, vp)
    sig("synchronize", i32, vp, vp)
    sig("pdsch_validate", i32, P(PdschPdu))
    sig("pdsch_derive", i32, P(PdschPdu), P(PdschDerived))
    sig("tbs_calculate", u32, u32, u32, u32, u32, C.c_float, u32, u32)
    sig("ofdm_symbol_size", u32, P(OfdmConfig), u32)
    sig("ofdm_slot_size", u32, P(OfdmConfig), u32)
    sig("pdsch_plan_create", i32, vp, u32, P(PdschPdu), P(u64), P(u32), u32, u32, u32, P(vp))
    sig("pdsch_plan_destroy", i32, vp)
    sig("pdsch_plan_nof_codeblocks", u32, vp)
    sig("pdsch_plan_codeword_bits", u64, vp)
    sig("pdsch_plan_codeword_offset", u64, vp, u32)
    sig("pdsch_run", i32, vp, u8p, vp, u8p, u8p, i32, vp)
    sig("pdsch_plan_enable_timing", i32, vp, u32)
    sig("pdsch_plan_kernel_times", i32, vp, P(C.c_float), P(u32))
    sig("ofdm_plan_enable_timing", i32, vp, u32)
    sig("ofdm_plan_kernel_time", i32, vp, P(C.c_float), P(u32))
    sig("pdsch_process_host", i32, vp, P(PdschPdu), u8p, vp, u32, u32, u8p, u8p)
    sig("pdsch_async_create", i32, vp, u32, u32, u32, u32, P(vp))
    sig("pdsch_async_submit", i32, vp, P(PdschPdu), u8p, vp, vp)
    sig("pdsch_async_wait", i32, vp)
    sig("pdsch_async_wait_slot", i32, vp)
    sig("pdsch_async_destroy", i32, vp)
    sig("pdsch_encode_host", i32, vp, P(PdschEncoderCfg), u8p, u8p, u8p)
    sig("ldpc_encode", i32, vp, u32, u32, u32, u8p, u32, u32, u8p, u32, vp)
    sig("ofdm_plan_create", i32, vp, P(OfdmConfig), u32, P(vp))
    sig("ofdm_plan_destroy", i32, vp)
    sig("ofdm_plan_slot_stride", u32, vp)
    sig("ofdm_run", i32, vp, u32, vp, vp, vp, vp)
    sig("ofdm_modulate_symbol_host", i32, vp, vp, u32, u32, vp, u32)
    sig("dft_run", i32, vp, u32, i32, u32, vp, vp, vp)
    sig("dft_run_host", i32, vp, u32, i32, vp, vp)
    sig("ofdm_modulate_slot_host", i32, vp, vp, u32, vp)
    sig("ofdm_demod_run", i32, vp, u32, vp, vp, u32, vp, vp)
    sig("ofdm_demodulate_slot_host", i32, vp, vp, u32, u32, vp)
    sig("ofdm_demodulate_symbol_host", i32, vp, vp, u32, u32, u32, vp)
    sig("ldpc_decoder_scratch_bytes", i32, vp, P(LdpcDecoderCfg), u32, P(u64))
    sig("ldpc_decoder_prepare", i32, vp, P(LdpcDecoderCfg))
    sig("ldpc_decode", i32, vp, P(LdpcDecoderCfg), u32, vp, u32, vp, u32, vp, vp, vp)
    sig("ldpc_decode_host", i32, vp, P(LdpcDecoderCfg), vp, u8p, P(u32))
    sig("ldpc_rate_dematch", i32, vp, P(LdpcRateDematcherCfg), u32, vp, u32, vp, u32, i32, vp)
    sig("ldpc_rate_dematch_host", i32, vp, P(LdpcRateDematcherCfg), vp, vp, i32)
    sig("llr_descramble", i32, vp, u32, vp, u32, vp, C.c_size_t, vp, C.c_size_t, vp)
    sig("llr_descramble_host", i32, vp, u32, u32, vp, vp)
    sig("pdsch_process_slot_host", i32, vp, u32, vp, vp, vp, u32, u32)
    sig("pdsch_async_submit_slot", i32, vp, u32, vp, vp, vp, vp)
    sig("demodulate_soft", i32, vp, u32, u32, u32, vp, vp, vp, vp)
    sig("demodulate_soft_host", i32, vp, u32, u32, vp, vp, vp)
    sig("grid_put", i32, vp, vp, u32, u32, u32, P(GridRe), vp)
    sig("csi_rs_validate", i32, P(CsiRsCfg))
    sig("csi_rs_map", i32, vp, u32, P(CsiRsCfg), P(u32), vp, u32, u32, vp)
    sig("csi_rs_map_host", i32, vp, P(CsiRsCfg), vp, u32, u32)
    sig("pdcch_validate", i32, P(PdcchPdu))
    sig("pdcch_process", i32, vp, u32, P(PdcchPdu), P(u32), vp, u32, u32, vp)
    sig("pdcch_process_host", i32, vp, P(PdcchPdu), vp, u32, u32)
    sig("pdcch_encode_host", i32, vp, u8p, u32, u32, u32, u8p)
    sig("ssb_validate", i32, P(SsbPdu))
    sig("ssb_process", i32, vp, u32, P(SsbPdu), P(u32), vp, u32, u32, vp)
    sig("ssb_process_host", i32, vp, P(SsbPdu), vp, u32, u32)
    sig("pbch_encode_host", i32, vp, P(SsbPdu), u8p)
    sig("amplitude_control", i32, vp, P(AmplitudeCfg), u32, u32, vp, C.c_size_t, vp, C.c_size_t, vp, vp)
    sig("amplitude_metrics", i32, P(AmplitudeCfg), P(AmplitudeStats), P(AmplitudeMetrics))
    sig("amplitude_control_host", i32, vp, P(AmplitudeCfg), vp, u32, vp, P(AmplitudeMetrics))
    sig("iq_convert_ci16", i32, vp, u32, u32, vp, C.c_size_t, C.c_float, vp, C.c_size_t, vp)
    sig("iq_convert_ci16_host", i32, vp, vp, u32, C.c_float, vp)
    sig("ofdm_run_ci16", i32, vp, u32, vp, vp, P(IqWireCfg), vp, vp, vp)
    sig("ofh_compressed_prb_bytes", u32, P(OfhCompressionCfg))
    sig("ofh_compress", i32, vp, P(OfhCompressionCfg), u32, u32, vp, C.c_size_t, vp, C.c_size_t, vp)
    sig("ofh_compress_host", i32, vp, P(OfhCompressionCfg), u32, vp, vp)
    sig("pusch_decoder_sizes", i32, vp, P(PuschDecoderCfg), u32, P(u64), P(u64), P(u64), P(u32))
    sig("pusch_decoder_prepare", i32, vp, P(PuschDecoderCfg))
    sig("pusch_decode_batch", i32, vp, P(PuschDecoderCfg), u32, vp, u64, vp, vp, vp, vp, u32, vp, vp)
    sig("dl_slots_create", i32, vp, P(DlSlotsCfg), P(vp))
    sig("dl_slots_destroy", i32, vp)
    sig("dl_slots_wait_free", i32, vp)
    sig("dl_slot_open", i32, vp, P(u32))
    sig("dl_slot_close", i32, vp, u32)
    sig("dl_slot_pdsch", i32, vp, u32, u32, vp, vp)
    sig("dl_slot_pdcch", i32, vp, u32, u32, P(PdcchPdu))
    sig("dl_slot_ssb", i32, vp, u32, u32, P(SsbPdu))
    sig("dl_slot_csi_rs", i32, vp, u32, u32, P(CsiRsCfg))
    sig("dl_slot_put", i32, vp, u32, u32, P(GridRe))
    sig("dl_slot_load_grid", i32, vp, u32, vp)
    sig("dl_slot_modulate", i32, vp, u32, u32, vp, vp)
    sig("dl_slot_poll", i32, vp, u32)
    sig("dl_slot_wait", i32, vp, u32)
    sig("dl_slot_iq", vp, vp, u32, u32, P(u32))
    sig("dl_slot_read_grid", i32, vp, u32, vp)
    sig("dl_slot_amplitude_stats", P(AmplitudeStats), vp, u32, u32)
    sig("dl_slot_device_grid", vp, vp, u32)
    sig("dl_slot_stream", vp, vp, u32)
    sig("pusch_decode_codeblock_host", i32, vp, P(LdpcRateDematcherCfg), u32, u32, C.c_float, vp, vp, i32, u8p, P(u32))
    return lib


# Symbols include/mi355_nrphy.h declares; tests check the shared library exports every one.
ABI_SYMBOLS = [
    "nrphy_version", "nrphy_strerror", "nrphy_trace_enabled", "nrphy_create", "nrphy_destroy", "nrphy_synchronize",
    "nrphy_pdsch_validate", "nrphy_pdsch_derive", "nrphy_tbs_calculate", "nrphy_ofdm_symbol_size",
    "nrphy_ofdm_slot_size", "nrphy_pdsch_plan_create", "nrphy_pdsch_plan_destroy",
    "nrphy_pdsch_plan_nof_codeblocks", "nrphy_pdsch_plan_codeword_bits", "nrphy_pdsch_plan_codeword_offset",
    "nrphy_pdsch_run", "nrphy_pdsch_plan_enable_timing", "nrphy_pdsch_plan_kernel_times", "nrphy_pdsch_plan_timing_stride",
    "nrphy_ofdm_plan_enable_timing", "nrphy_ofdm_plan_kernel_time", "nrphy_ofdm_plan_timing_stride", "nrphy_pdsch_process_host", "nrphy_pdsch_encode_host", "nrphy_ldpc_encode", "nrphy_ofdm_plan_create",
    "nrphy_ofdm_plan_destroy", "nrphy_ofdm_plan_slot_stride", "nrphy_ofdm_run",
    "nrphy_ofdm_modulate_symbol_host", "nrphy_ofdm_modulate_slot_host", "nrphy_dft_run", "nrphy_dft_run_host",
    "nrphy_ofdm_demod_run", "nrphy_ofdm_demodulate_slot_host", "nrphy_ofdm_demodulate_symbol_host",
    "nrphy_ldpc_decode", "nrphy_ldpc_decode_host", "nrphy_ldpc_rate_dematch", "nrphy_ldpc_rate_dematch_host",
    "nrphy_pusch_decode_codeblock_host", "nrphy_pusch_decoder_sizes", "nrphy_pusch_decode_batch",
    "nrphy_ldpc_decoder_scratch_bytes", "nrphy_ldpc_decoder_prepare", "nrphy_pusch_decoder_prepare",
    "nrphy_csi_rs_validate", "nrphy_csi_rs_map", "nrphy_csi_rs_map_host", "nrphy_grid_put",
    "nrphy_llr_descramble", "nrphy_llr_descramble_host", "nrphy_demodulate_soft", "nrphy_demodulate_soft_host", "nrphy_pdsch_process_slot_host", "nrphy_pdsch_async_submit_slot",
    "nrphy_pdcch_validate", "nrphy_pdcch_process", "nrphy_pdcch_process_host", "nrphy_pdcch_encode_host",
    "nrphy_ssb_validate", "nrphy_ssb_process", "nrphy_ssb_process_host", "nrphy_pbch_encode_host",
    "nrphy_pdsch_async_create", "nrphy_pdsch_async_submit", "nrphy_pdsch_async_wait", "nrphy_pdsch_async_wait_slot",
    "nrphy_pdsch_async_destroy",
    "nrphy_pdsch_async_count_done",
    "nrphy_amplitude_control", "nrphy_amplitude_metrics", "nrphy_amplitude_control_host", "nrphy_iq_convert_ci16",
    "nrphy_iq_convert_ci16_host", "nrphy_ofdm_run_ci16", "nrphy_ofh_compressed_prb_bytes", "nrphy_ofh_compress",
    "nrphy_ofh_compress_host",
    "nrphy_dl_slots_create", "nrphy_dl_slots_destroy", "nrphy_dl_slots_wait_free", "nrphy_dl_slot_open", "nrphy_dl_slot_close",
    "nrphy_dl_slot_pdsch", "nrphy_dl_slot_pdcch", "nrphy_dl_slot_ssb", "nrphy_dl_slot_csi_rs", "nrphy_dl_slot_put",
    "nrphy_dl_slot_load_grid", "nrphy_dl_slot_modulate", "nrphy_dl_slot_poll", "nrphy_dl_slot_wait", "nrphy_dl_slot_iq",
    "nrphy_dl_slot_read_grid", "nrphy_dl_slot_device_grid", "nrphy_dl_slot_stream", "nrphy_dl_slot_amplitude_stats",
]
