"""TEST INFRASTRUCTURE: numpy-level wrappers over the CPU checkers.

* ``oracle()``  -> oracle/liboracle.so, the plain-C restatement (travels to the GPU box).
* ``ref()``     -> oracle/_ref/libsrsref.so, the reference's own sources compiled by oracle/Makefile
                   (exists only where /root/reference was available at build time; None otherwise).

Nothing in the product imports this module.
"""
import ctypes as C
import importlib.util
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "srsran-edgeric-5g_amd")


def load_package():
    """Imports the (hyphen-named) package directory as module ``srsran_edgeric_5g_amd``."""
    name = "srsran_edgeric_5g_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


pkg = load_package()
abi = pkg.abi

_P = C.POINTER
_u32, _vp, _f, _i = C.c_uint32, C.c_void_p, C.c_float, C.c_int


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class CpuBackend:
    """Uniform numpy API over the oracle (prefix 'oracle_') or the compiled reference (prefix 'ref_')."""

    def __init__(self, lib, prefix):
        self.lib = lib
        self.prefix = prefix
        self.is_ref = prefix == "ref_"
        g = lambda n: getattr(lib, prefix + n)
        g("tbs_calculate").restype = _u32
        g("tbs_calculate").argtypes = [_u32, _u32, _u32, _u32, _f, _u32, _u32]
        g("crc").restype = _u32
        g("crc").argtypes = [_u32, _vp, _u32]
        g("ldpc_segment").restype = _i
        g("ldpc_segment").argtypes = [_u32] * 6 + [_vp, _u32, _vp, _u32, _vp, _vp]
        g("ldpc_encode").restype = _i
        g("ldpc_rate_match").restype = _i
        g("ldpc_rate_match").argtypes = [_u32] * 6 + [_vp, _u32, _vp, _u32]
        g("prg_apply_xor").restype = None
        g("prg_apply_xor").argtypes = [_u32, _u32, _vp, _u32]
        g("prg_generate_float").restype = None
        g("prg_generate_float").argtypes = [_u32, _u32, _f, _vp, _u32]
        g("modulate_ci8").restype = _f
        g("modulate_ci8").argtypes = [_u32, _vp, _u32, _vp]
        g("pdsch_validate").restype = _i
        g("pdsch_validate").argtypes = [_P(abi.PdschPdu)]
        g("dft").restype = _i
        g("dft").argtypes = [_u32, _i, _vp, _vp]
        g("ofdm_modulate_slot").restype = _i
        g("ofdm_modulate_slot").argtypes = [_P(abi.OfdmConfig), _vp, _u32, _u32, _vp]
        if self.is_ref:
            g("ldpc_encode").argtypes = [_u32, _u32, _vp, _u32, _vp, _i]
            g("pdsch_process").restype = _i
            g("pdsch_process").argtypes = [_P(abi.PdschPdu), _vp, _vp, _u32, _u32, _i, _i]
            g("pdsch_encode").restype = _i
            g("pdsch_encode").argtypes = [_u32] * 6 + [_vp, _u32, _vp, _i]
            g("precoding_codebook").restype = _u32
            g("precoding_codebook").argtypes = [_u32] * 4 + [_vp]
            g("bench_pdsch").restype = C.c_double
            g("bench_pdsch").argtypes = [_P(abi.PdschPdu), _vp, _u32, _u32, _P(abi.OfdmConfig), _u32, _u32, _i]
        else:
            g("ldpc_encode").argtypes = [_u32, _u32, _vp, _u32, _vp]
            g("pdsch_process").restype = _i
            g("pdsch_process").argtypes = [_P(abi.PdschPdu), _vp, _vp, _u32, _u32, _vp, _vp]
            g("pdsch_encode").restype = _i
            g("pdsch_encode").argtypes = [_P(abi.PdschPdu), _vp, _vp]
            g("pdsch_derive").restype = _i
            g("pdsch_derive").argtypes = [_P(abi.PdschPdu), _P(abi.PdschDerived)]
            g("crc_bits").restype = _u32
            g("crc_bits").argtypes = [_u32, _vp, _u32]
            g("ofdm_symbol_size").restype = _u32
            g("ofdm_symbol_size").argtypes = [_P(abi.OfdmConfig), _u32]
            g("ofdm_slot_size").restype = _u32
            g("ofdm_slot_size").argtypes = [_P(abi.OfdmConfig), _u32]
            g("bench").restype = C.c_double
            g("bench").argtypes = [_P(abi.PdschPdu), _vp, _u32, _u32, _P(abi.OfdmConfig), _u32, _u32]

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    # ---- scalar helpers -------------------------------------------------------------------------
    def tbs(self, nof_symb_sh, nof_dmrs_prb, nof_oh_prb, qm, rate_x1024, nof_layers, n_prb):
        return int(self._f("tbs_calculate")(nof_symb_sh, nof_dmrs_prb, nof_oh_prb, qm, float(rate_x1024),
                                            nof_layers, n_prb))

    def crc(self, poly, data):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        return int(self._f("crc")(poly, _ptr(data), data.size))

    def crc_bits(self, poly, bits):
        """CRC of a bit string given one bit per byte (oracle only)."""
        assert not self.is_ref
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        packed = np.packbits(bits)
        f = self._f("crc_bits")
        f.restype = C.c_uint32
        return int(f(C.c_uint32(poly), _ptr(packed), C.c_uint32(bits.size)))

    def validate(self, pdu):
        return int(self._f("pdsch_validate")(C.byref(pdu)))

    # ---- coding ------------------------------------------------------------------------------------
    def segment(self, bg, rv, qm, nref, nof_layers, nof_ch_symbols, tb):
        tb = np.ascontiguousarray(tb, dtype=np.uint8)
        stride = 8448 // 8
        segs = np.zeros((abi.MAX_CODEBLOCKS, stride), np.uint8)
        meta = np.zeros((abi.MAX_CODEBLOCKS, 5), np.uint32)
        zc = C.c_uint32(0)
        n = self._f("ldpc_segment")(bg, rv, qm, nref, nof_layers, nof_ch_symbols, _ptr(tb), tb.size, _ptr(segs),
                                    stride, _ptr(meta), C.addressof(zc))
        k = (22 if bg == 1 else 10) * zc.value
        return segs[:n, :(k + 7) // 8].copy(), meta[:n].copy(), zc.value

    def ldpc_encode(self, bg, zc, msg_packed, out_bits, simd=1):
        msg_packed = np.ascontiguousarray(msg_packed, dtype=np.uint8)
        out = np.zeros((out_bits + 7) // 8, np.uint8)
        args = [bg, zc, _ptr(msg_packed), out_bits, _ptr(out)]
        if self.is_ref:
            args.append(simd)
        rc = self._f("ldpc_encode")(*args)
        assert rc == 0, rc
        return out

    def rate_match(self, bg, zc, rv, qm, nref, nof_filler, codeblock_packed, rm_length):
        cb = np.ascontiguousarray(codeblock_packed, dtype=np.uint8)
        n = (66 if bg == 1 else 50) * zc
        out = np.zeros((rm_length + 7) // 8, np.uint8)
        rc = self._f("ldpc_rate_match")(bg, zc, rv, qm, nref, nof_filler, _ptr(cb), n, _ptr(out), rm_length)
        assert rc == 0, rc
        return out

    def prg_xor(self, c_init, offset, data_packed, nbits):
        d = np.array(data_packed, dtype=np.uint8, copy=True)
        self._f("prg_apply_xor")(c_init, offset, _ptr(d), nbits)
        return d

    def prg_float(self, c_init, offset, value, n):
        out = np.zeros(n, np.float32)
        self._f("prg_generate_float")(c_init, offset, float(value), _ptr(out), n)
        return out

    def modulate(self, qm, bits_packed, nsym):
        b = np.ascontiguousarray(bits_packed, dtype=np.uint8)
        out = np.zeros((nsym, 2), np.int8)
        scale = self._f("modulate_ci8")(qm, _ptr(b), nsym, _ptr(out))
        return out, float(scale)

    # ---- PDSCH ------------------------------------------------------------------------------------
    def pdsch_encode_cfg(self, bg, rv, qm, nref, nof_layers, nof_ch_symbols, tb, simd=1):
        """pdsch_encoder::encode for an explicit configuration, composed from the per-block functions (segment -> encode
        -> rate match -> concatenate): the packed codeword."""
        segs, meta, zc = self.segment(bg, rv, qm, nref, nof_layers, nof_ch_symbols, tb)
        out = np.zeros(nof_ch_symbols * qm, np.uint8)
        for seg, (rm_length, cw_offset, nof_filler, _full, _crc) in zip(segs, meta.tolist()):
            cb = self.ldpc_encode(bg, zc, seg, (66 if bg == 1 else 50) * zc, simd)
            e = self.rate_match(bg, zc, rv, qm, nref, nof_filler, cb, rm_length)
            out[cw_offset:cw_offset + rm_length] = np.unpackbits(e)[:rm_length]
        return np.packbits(out)

    def derive(self, pdu):
        d = abi.PdschDerived()
        assert not self.is_ref
        self._f("pdsch_derive")(C.byref(pdu), C.byref(d))
        return d.as_dict()

    def pdsch_encode(self, pdu, tb, derived=None, simd=1):
        """Packed rate-matched codeword (before scrambling)."""
        tb = np.ascontiguousarray(tb, dtype=np.uint8)
        if self.is_ref:
            d = derived
            unpacked = np.zeros(d["codeword_bits"], np.uint8)
            rc = self._f("pdsch_encode")(pdu.ldpc_base_graph, pdu.rv, pdu.qm, d["n_ref"], pdu.nof_layers,
                                         d["nof_re"] * pdu.nof_layers, _ptr(tb), tb.size, _ptr(unpacked), simd)
            assert rc == 0
            return np.packbits(unpacked)
        d = self.derive(pdu)
        out = np.zeros((d["codeword_bits"] + 7) // 8, np.uint8)
        rc = self._f("pdsch_encode")(C.byref(pdu), _ptr(tb), _ptr(out))
        assert rc == 0
        return out

    def pdsch_process(self, pdu, tb, nof_ports, nof_subc, simd=1, impl=0, taps=False, codeword_bits=None):
        """Returns grid [nof_ports][14][nof_subc][2] uint16 (raw bf16) (+ (cw_rm, cw_scr) for the oracle)."""
        tb = np.ascontiguousarray(tb, dtype=np.uint8)
        grid = np.zeros((nof_ports, 14, nof_subc, 2), np.uint16)
        if self.is_ref:
            rc = self._f("pdsch_process")(C.byref(pdu), _ptr(tb), _ptr(grid), nof_ports, nof_subc, impl, simd)
            assert rc == 0, rc
            return grid
        if taps:
            nb = (codeword_bits + 7) // 8
            rm = np.zeros(nb, np.uint8)
            scr = np.zeros(nb, np.uint8)
            rc = self._f("pdsch_process")(C.byref(pdu), _ptr(tb), _ptr(grid), nof_ports, nof_subc, _ptr(rm), _ptr(scr))
            assert rc == 0, rc
            return grid, rm, scr
        rc = self._f("pdsch_process")(C.byref(pdu), _ptr(tb), _ptr(grid), nof_ports, nof_subc, None, None)
        assert rc == 0, rc
        return grid

    # ---- OFDM ---------------------------------------------------------------------------------------
    def dft(self, x, inverse):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        out = np.zeros_like(x)
        rc = self._f("dft")(x.size, int(inverse), _ptr(x), _ptr(out))
        assert rc == 0, rc
        return out

    def ofdm_slot(self, cfg, grid, slot_index=0):
        """grid: [nof_ports][14][12*bw_rb][2] uint16 -> iq [nof_ports][slot_size] complex64."""
        grid = np.ascontiguousarray(grid, dtype=np.uint16)
        nof_ports = grid.shape[0]
        iq = np.zeros((nof_ports, 2 * (cfg.dft_size + cfg.dft_size // 8) * 14), np.complex64)  # generous
        n = self._f("ofdm_modulate_slot")(C.byref(cfg), _ptr(grid), nof_ports, slot_index, _ptr(iq))
        assert n > 0, n
        return iq.reshape(-1)[: nof_ports * n].reshape(nof_ports, n).copy()

    def ldpc_decode(self, bg, zc, nof_filler, crc_poly_id, max_iterations, scaling, llr, simd=0):
        """ldpc_decoder::decode: llr int8 (codeblock without its first 2*Zc bits) -> (iterations or 0, Kb*Zc bits)."""
        llr = np.ascontiguousarray(llr, dtype=np.int8)
        bits = np.zeros((22 if bg == 1 else 10) * zc, np.uint8)
        f = self._f("ldpc_decode")
        f.restype = C.c_int
        args = [C.c_uint32(bg), C.c_uint32(zc), C.c_uint32(nof_filler), C.c_uint32(crc_poly_id),
                C.c_uint32(max_iterations), C.c_float(scaling), _ptr(llr), C.c_uint32(llr.size), _ptr(bits)]
        if self.is_ref:
            args.append(C.c_int(simd))
        return int(f(*args)), bits

    def ldpc_rate_dematch(self, bg, zc, rv, qm, nref, nof_filler, new_data, llr_in, soft_buffer, simd=0):
        """ldpc_rate_dematcher::rate_dematch: returns the updated soft buffer ((66 or 50) * Zc int8, copy)."""
        llr_in = np.ascontiguousarray(llr_in, dtype=np.int8)
        out = np.array(soft_buffer, dtype=np.int8, copy=True)
        assert out.size == (66 if bg == 1 else 50) * zc
        args = [C.c_uint32(bg), C.c_uint32(zc), C.c_uint32(rv), C.c_uint32(qm), C.c_uint32(nref),
                C.c_uint32(nof_filler), C.c_int(int(new_data)), _ptr(llr_in), C.c_uint32(llr_in.size), _ptr(out)]
        if self.is_ref:
            args.append(C.c_int(simd))
        f = self._f("ldpc_rate_dematch")
        f.restype = C.c_int
        assert f(*args) == 0
        return out

    def prg_apply_xor_llr(self, c_init, offset, llr):
        """pseudo_random_generator::apply_xor on soft bits (sign flips); returns a new int8 array."""
        llr = np.ascontiguousarray(llr, dtype=np.int8)
        out = np.empty_like(llr)
        f = self._f("prg_apply_xor_llr")
        f.restype = None
        f(C.c_uint32(c_init), C.c_uint32(offset), _ptr(llr), _ptr(out), C.c_uint32(llr.size))
        return out

    def demodulate_soft(self, modulation, symbols, noise_vars):
        """demodulation_mapper::demodulate_soft of one span: complex64 [n], float32 [n] -> int8 [n * bits per symbol]."""
        symbols = np.ascontiguousarray(symbols, dtype=np.complex64)
        noise_vars = np.ascontiguousarray(noise_vars, dtype=np.float32)
        out = np.zeros(symbols.size * max(modulation, 1), np.int8)
        f = self._f("demodulate_soft")
        f.argtypes = [C.c_uint32, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = f(modulation, symbols.size, symbols.ctypes.data, noise_vars.ctypes.data, out.ctypes.data)
        assert rc == 0, rc
        return out

    def csi_rs_map(self, cfg, grid, simd=1):
        """nzp_csi_rs_generator::map into a copy of grid [nof_ports][14][nof_subc][2] uint16 (raw cbf16)."""
        out = np.array(grid, dtype=np.uint16, copy=True)
        nof_ports, _, nof_subc, _ = out.shape
        args = [C.byref(cfg), _ptr(out), C.c_uint32(nof_ports), C.c_uint32(nof_subc)]
        if self.is_ref:
            args.append(C.c_int(simd))
        rc = self._f("csi_rs_map")(*args)
        assert rc == 0, rc
        return out

    def csi_rs_validate(self, cfg):
        assert not self.is_ref
        return int(self._f("csi_rs_validate")(C.byref(cfg)))

    def pusch_decode(self, cfg, harq_id, nof_codeblocks, llr):
        """pusch_decoder_impl of the compiled reference on one transport block (HARQ state kept in the reference's
        rx_buffer_pool under harq_id): returns (tb_crc_ok, decoder runs, iteration sum, iteration max, tb bytes)."""
        assert self.is_ref
        llr = np.ascontiguousarray(llr, dtype=np.int8)
        tb = np.zeros(cfg.tb_size_bytes, np.uint8)
        res = np.zeros(4, np.uint32)
        f = self._f("pusch_decode")
        f.restype = C.c_int
        rc = f(C.c_uint32(cfg.base_graph), C.c_uint32(cfg.qm), C.c_uint32(cfg.rv), C.c_uint32(cfg.nof_layers),
               C.c_uint32(cfg.nref), C.c_uint32(cfg.tb_size_bytes), C.c_uint32(cfg.max_iterations),
               C.c_int(cfg.use_early_stop), C.c_int(cfg.new_data), C.c_uint32(harq_id), C.c_uint32(nof_codeblocks),
               _ptr(llr), C.c_uint32(llr.size), _ptr(tb), _ptr(res))
        assert rc == 0, rc
        return bool(res[0]), int(res[1]), int(res[2]), int(res[3]), tb

    def ofdm_demod_slot(self, cfg, iq, slot_index=0, window_offset=0):
        """iq: [nof_ports][slot_size] complex64 -> grid [nof_ports][14][12*bw_rb][2] uint16 (raw bf16)."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        nof_ports = iq.shape[0]
        grid = np.zeros((nof_ports, 14, 12 * cfg.bw_rb, 2), np.uint16)
        n = self._f("ofdm_demodulate_slot")(C.byref(cfg), _ptr(iq), nof_ports, slot_index, window_offset, _ptr(grid))
        assert n == iq.shape[1], (n, iq.shape)
        return grid

    # ---- downlink control side (section 8f-2) ---------------------------------------------------------------
    def polar_code(self, K, E, n_max=9):
        """polar_code::set -> (N, mask of the K information positions as N bytes)."""
        mask = np.zeros(1024, np.uint8)
        f = self._f("polar_code")
        f.restype = C.c_int
        n = f(C.c_uint32(K), C.c_uint32(E), C.c_uint32(n_max), _ptr(mask))
        assert n > 0, n
        return n, mask[:n].copy()

    def pdcch_encode(self, payload_bits, rnti, rm_length):
        """pdcch_encoder::encode: payload bits (one per byte) -> rm_length bits (one per byte)."""
        payload = np.ascontiguousarray(payload_bits, dtype=np.uint8)
        out = np.zeros(rm_length, np.uint8)
        f = self._f("pdcch_encode")
        f.restype = C.c_int
        rc = f(_ptr(payload), C.c_uint32(payload.size), C.c_uint32(rnti), C.c_uint32(rm_length), _ptr(out))
        assert rc == 0, rc
        return out

    def pdcch_validate(self, pdu):
        assert not self.is_ref
        f = self._f("pdcch_validate")
        f.restype = C.c_int
        return int(f(C.byref(pdu)))

    def pdcch_process(self, pdu, grid, simd=1):
        """pdcch_processor::process into a copy of grid [nof_ports][14][nof_subc][2] uint16 (raw cbf16)."""
        out = np.array(grid, dtype=np.uint16, copy=True)
        nof_ports, _, nof_subc, _ = out.shape
        args = [C.byref(pdu), _ptr(out), C.c_uint32(nof_ports), C.c_uint32(nof_subc)]
        if self.is_ref:
            args.append(C.c_int(simd))
        f = self._f("pdcch_process")
        f.restype = C.c_int
        rc = f(*args)
        assert rc == 0, rc
        return out

    def ssb_validate(self, pdu):
        assert not self.is_ref
        f = self._f("ssb_validate")
        f.restype = C.c_int
        return int(f(C.byref(pdu)))

    def pbch_encode(self, pdu):
        """pbch_encoder::encode -> 864 bits (one per byte)."""
        out = np.zeros(864, np.uint8)
        f = self._f("pbch_encode")
        f.restype = C.c_int
        rc = f(C.byref(pdu), _ptr(out))
        assert rc == 0, rc
        return out

    def ssb_process(self, pdu, grid):
        """ssb_processor::process into a copy of grid [nof_ports][14][nof_subc][2] uint16 (raw cbf16)."""
        out = np.array(grid, dtype=np.uint16, copy=True)
        nof_ports, _, nof_subc, _ = out.shape
        f = self._f("ssb_process")
        f.restype = C.c_int
        rc = f(C.byref(pdu), _ptr(out), C.c_uint32(nof_ports), C.c_uint32(nof_subc))
        assert rc == 0, rc
        return out

    # ---- lower-PHY tail (section 8f-3) ------------------------------------------------------------------------
    def amplitude_control(self, cfg, x):
        """amplitude_controller::process on one buffer (complex64) -> (output, dict of measurements).  For the reference:
        avg_power_fs, peak_power_fs, papr_lin, gain_dB, nof_processed, nof_clipped; for the oracle the raw statistics."""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        out = np.zeros_like(x)
        if self.is_ref:
            m, cnt = np.zeros(4, np.float32), np.zeros(2, np.uint64)
            f = self._f("amplitude_control")
            f.restype = C.c_int
            rc = f(C.c_int(cfg.kind), C.c_int(cfg.enable_clipping), C.c_float(cfg.input_gain_dB), C.c_float(cfg.full_scale_lin),
                   C.c_float(cfg.ceiling_dBFS), _ptr(x), C.c_uint32(x.size), _ptr(out), _ptr(m), _ptr(cnt))
            assert rc == 0
            return out, dict(avg_power_fs=float(m[0]), peak_power_fs=float(m[1]), papr_lin=float(m[2]), gain_dB=float(m[3]),
                             nof_processed=int(cnt[0]), nof_clipped=int(cnt[1]))
        st = abi.AmplitudeStats()
        f = self._f("amplitude_control")
        f.restype = C.c_int
        assert f(C.byref(cfg), _ptr(x), C.c_uint32(x.size), _ptr(out), C.byref(st)) == 0
        m = abi.AmplitudeMetrics()
        g = self._f("amplitude_metrics")
        g.restype = C.c_int
        assert g(C.byref(cfg), C.byref(st), C.byref(m)) == 0
        return out, dict(avg_power_fs=m.avg_power_fs, peak_power_fs=m.peak_power_fs, papr_lin=m.papr_lin, gain_dB=m.gain_dB,
                         nof_processed=int(m.nof_processed_samples), nof_clipped=int(m.nof_clipped_samples), stats=st)

    def iq_convert_ci16(self, x, scale):
        """srsvec::convert(cf -> int16 with scale): complex64 [n] -> int16 [2n]."""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        out = np.zeros(2 * x.size, np.int16)
        f = self._f("convert_cf_to_ci16" if self.is_ref else "iq_convert_ci16")
        f.restype = C.c_int
        assert f(_ptr(x), C.c_uint32(x.size), C.c_float(scale), _ptr(out)) == 0
        return out

    def ofh_compress(self, cfg, prbs, simd=0):
        """iq_compressor::compress + serialisation of one call's PRBs: prbs [nof_prb][12][2] uint16 (raw cbf16) -> bytes."""
        prbs = np.ascontiguousarray(prbs, dtype=np.uint16)
        nof_prb = prbs.size // 24
        out = np.zeros(nof_prb * 49, np.uint8)
        f = self._f("ofh_compress")
        f.restype = C.c_int
        if self.is_ref:
            n = f(C.c_int(cfg.type), C.c_int(simd), C.c_uint32(cfg.data_width), C.c_float(cfg.iq_scaling), _ptr(prbs),
                  C.c_uint32(nof_prb), _ptr(out))
        else:
            n = f(C.byref(cfg), _ptr(prbs), C.c_uint32(nof_prb), _ptr(out))
        assert n > 0, n
        return out[:n].copy()

    def codebook(self, kind, a=0, b=0, c=0):
        assert self.is_ref
        w = np.zeros((4, 4, 2), np.float32)
        r = self._f("precoding_codebook")(kind, a, b, c, _ptr(w))
        ports, layers = r >> 8, r & 0xFF
        return w.reshape(-1)[: ports * layers * 2].reshape(1, ports, layers, 2).copy()


_ORACLE = None
_REF = None


def build_oracle():
    """Compiles oracle/liboracle.so when missing or stale (gcc, a second or two)."""
    if os.environ.get("NRPHY_ORACLE_SO"):      # e.g. the sanitizer build (oracle/Makefile: make sanitize)
        return os.path.abspath(os.environ["NRPHY_ORACLE_SO"])
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", n) for n in ("nrphy_oracle.c", "nrphy_oracle_dl.c", "nrphy_oracle_lower.c", "nrphy_oracle_rx.c",
                                                       "nrphy_oracle.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True, capture_output=True,
                       timeout=300)
    return so


def oracle():
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = CpuBackend(C.CDLL(build_oracle()), "oracle_")
    return _ORACLE


def ref():
    """The compiled reference, or None when oracle/_ref was not built (e.g. on the GPU box)."""
    global _REF
    so = os.path.join(ROOT, "oracle", "_ref", "libsrsref.so")
    if _REF is None and os.path.exists(so):
        _REF = CpuBackend(C.CDLL(so), "ref_")
    return _REF


def bf16_to_f32(u16):
    return (np.asarray(u16, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def grid_to_complex(grid_u16):
    f = bf16_to_f32(grid_u16)
    return f[..., 0] + 1j * f[..., 1]
