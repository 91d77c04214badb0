"""Loader and thin object wrappers for csrc/libmi355nrphy.so (the C ABI of include/mi355_nrphy.h).

There is deliberately no CPU implementation behind these classes: if the HIP library is missing or no GPU is
present the constructors raise.  PyTorch is used only as the device-memory allocator / stream provider of the
callers (tests, bench.py): tensors are passed down as raw device pointers.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
# NRPHY_LIB_SO: another build of the same library (A/B variants, the host-sanitizer build of profiles/sanitize_cpu.sh)
LIB_PATH = os.environ.get("NRPHY_LIB_SO") or os.path.join(HERE, "csrc", "libmi355nrphy.so")

_LIB = None


class NrphyError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        super().__init__("%s failed: status %d (%s)" % (what, status, strerror(status)))


def load():
    """Returns the ctypes handle of libmi355nrphy.so; raises if it has not been built."""
    global _LIB
    if _LIB is None:
        try:
            # PyTorch bundles its own HIP runtime; let it load first so that this library binds to the same
            # libamdhip64 instead of bringing a second runtime into the process (device memory comes from torch).
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("HIP library %s is missing: run `python srsran-edgeric-5g_amd/build.py` "
                               "(there is no CPU fallback)" % LIB_PATH)
        _LIB = abi.declare(C.CDLL(LIB_PATH))
    return _LIB


def strerror(status):
    return load().nrphy_strerror(status).decode()


def _check(status, what):
    if status != abi.OK:
        raise NrphyError(status, what)


# Callers that order their PyTorch work and this library's streams themselves (bench.py: explicit synchronisation around the timed
# region) switch the wait below off.
ORDER_AFTER_TORCH = True


def _stream(stream):
    """The stream argument of a device entry point.  None selects the context's own stream, which is non-blocking and therefore
    NOT ordered after PyTorch's streams: work PyTorch has queued on its current stream (the fill of a fresh torch.zeros, a
    host-to-device copy) is waited for first, so that a tensor handed over is what the caller sees.  An explicit stream is the
    caller's to order."""
    if stream is None and ORDER_AFTER_TORCH:
        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()
    return stream


def _dptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


class Context:
    """nrphy_ctx: device tables + default stream.  Replaces the reference's factory chain."""

    def __init__(self, device_id=0):
        self.lib = load()
        h = C.c_void_p()
        _check(self.lib.nrphy_create(C.byref(h), device_id), "nrphy_create")
        self.handle = h
        self.device_id = device_id

    def close(self):
        if self.handle:
            self.lib.nrphy_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self, stream=None):
        _check(self.lib.nrphy_synchronize(self.handle, stream), "nrphy_synchronize")

    # ---- single-PDU host-span entry points (reference semantics) --------------------------------------
    def pdsch_process_host(self, pdu, tb, nof_ports, nof_subc, grid=None, taps=False):
        """pdsch_processor::process with host spans.  Returns grid [ports][14][subc][2] uint16 (raw bf16)
        and, with taps, the packed rate-matched and scrambled codewords."""
        tb = np.ascontiguousarray(tb, dtype=np.uint8)
        if grid is None:
            grid = np.zeros((nof_ports, 14, nof_subc, 2), np.uint16)
        d = derive(pdu)
        nb = (d["codeword_bits"] + 7) // 8
        rm = np.zeros(nb, np.uint8) if taps else None
        scr = np.zeros(nb, np.uint8) if taps else None
        _check(self.lib.nrphy_pdsch_process_host(
            self.handle, C.byref(pdu), tb.ctypes.data, grid.ctypes.data, nof_ports, nof_subc,
            rm.ctypes.data if taps else None, scr.ctypes.data if taps else None), "nrphy_pdsch_process_host")
        return (grid, rm, scr) if taps else grid

    def pdsch_encode_host(self, base_graph, rv, qm, nref, nof_layers, nof_ch_symbols, tb):
        """pdsch_encoder::encode: returns (codeword bits one per byte, the same packed MSB-first)."""
        tb = np.ascontiguousarray(tb, dtype=np.uint8)
        cfg = abi.PdschEncoderCfg(base_graph, rv, qm, nref, nof_layers, nof_ch_symbols, tb.size)
        bits = np.zeros(nof_ch_symbols * qm, np.uint8)
        packed = np.zeros((bits.size + 7) // 8, np.uint8)
        _check(self.lib.nrphy_pdsch_encode_host(self.handle, C.byref(cfg), tb.ctypes.data, bits.ctypes.data,
                                                packed.ctypes.data), "nrphy_pdsch_encode_host")
        return bits, packed

    def ldpc_encode(self, base_graph, lifting_size, d_msg, msg_stride, out_bits, d_out, out_stride, n_cb, stream=None):
        _check(self.lib.nrphy_ldpc_encode(self.handle, base_graph, lifting_size, n_cb, _dptr(d_msg), msg_stride,
                                          out_bits, _dptr(d_out), out_stride, _stream(stream)), "nrphy_ldpc_encode")

    def ldpc_rate_dematch(self, cfg, n_cb, d_in, in_stride, d_soft, soft_stride, new_data, stream=None):
        """ldpc_rate_dematcher::rate_dematch for n_cb codeblocks resident in HBM (cfg: abi.LdpcRateDematcherCfg)."""
        _check(self.lib.nrphy_ldpc_rate_dematch(self.handle, C.byref(cfg), n_cb, _dptr(d_in), in_stride, _dptr(d_soft),
                                                soft_stride, int(new_data), _stream(stream)), "nrphy_ldpc_rate_dematch")

    def ldpc_rate_dematch_host(self, base_graph, lifting_size, rv, qm, nref, nof_filler, new_data, llr_in, soft_buffer):
        """ldpc_rate_dematcher::rate_dematch on host spans: returns the updated soft buffer (copy)."""
        llr_in = np.ascontiguousarray(llr_in, dtype=np.int8)
        out = np.array(soft_buffer, dtype=np.int8, copy=True)
        cfg = abi.LdpcRateDematcherCfg(base_graph, lifting_size, rv, qm, nref, nof_filler, llr_in.size)
        _check(self.lib.nrphy_ldpc_rate_dematch_host(self.handle, C.byref(cfg), llr_in.ctypes.data, out.ctypes.data,
                                                     int(new_data)), "nrphy_ldpc_rate_dematch_host")
        return out

    def pusch_decode_codeblock_host(self, base_graph, lifting_size, rv, qm, nref, nof_filler, crc_poly, max_iterations,
                                    scaling, new_data, llr_in, soft_buffer):
        """Rate dematcher + decoder of one codeblock on host spans (the hw_accelerator_pusch_dec operation):
        returns (iterations or 0, Kb*Zc hard bits, updated soft buffer)."""
        llr_in = np.ascontiguousarray(llr_in, dtype=np.int8)
        soft = np.array(soft_buffer, dtype=np.int8, copy=True)
        cfg = abi.LdpcRateDematcherCfg(base_graph, lifting_size, rv, qm, nref, nof_filler, llr_in.size)
        k = (22 if base_graph == 1 else 10) * lifting_size
        packed = np.zeros((k + 7) // 8, np.uint8)
        it = C.c_uint32(0)
        _check(self.lib.nrphy_pusch_decode_codeblock_host(self.handle, C.byref(cfg), crc_poly, max_iterations, scaling,
                                                          llr_in.ctypes.data, soft.ctypes.data, int(new_data),
                                                          packed.ctypes.data, C.byref(it)),
               "nrphy_pusch_decode_codeblock_host")
        return int(it.value), np.unpackbits(packed)[:k], soft

    def llr_descramble(self, d_c_init, n_cw, length, d_in, in_stride, d_out, out_stride, stream=None):
        """pseudo_random_generator::apply_xor on soft bits for n_cw codewords in device memory (d_c_init: device uint32)."""
        _check(self.lib.nrphy_llr_descramble(self.handle, n_cw, _dptr(d_c_init), length, _dptr(d_in), in_stride, _dptr(d_out),
                                             out_stride, _stream(stream)), "nrphy_llr_descramble")

    def llr_descramble_host(self, c_init, llr):
        """One codeword of int8 soft bits from host memory; returns the descrambled copy."""
        llr = np.ascontiguousarray(llr, dtype=np.int8)
        out = np.empty_like(llr)
        _check(self.lib.nrphy_llr_descramble_host(self.handle, c_init, llr.size, llr.ctypes.data, out.ctypes.data),
               "nrphy_llr_descramble_host")
        return out

    def pdsch_process_slot_host(self, pdus, tbs, grid):
        """All PDSCH PDUs of one slot into one host grid [nof_ports][14][nof_subc][2] uint16 (read first, written back)."""
        n = len(pdus)
        arr = (abi.PdschPdu * n)(*pdus)
        keep = [np.ascontiguousarray(t, dtype=np.uint8) for t in tbs]
        ptrs = (C.c_void_p * n)(*[t.ctypes.data for t in keep])
        grid = np.ascontiguousarray(grid, dtype=np.uint16)
        _check(self.lib.nrphy_pdsch_process_slot_host(self.handle, n, arr, ptrs, grid.ctypes.data, grid.shape[0], grid.shape[2]),
               "nrphy_pdsch_process_slot_host")
        return grid

    def demodulate_soft(self, modulation, nof_spans, span_len, d_symbols, d_noise_vars, d_llr, stream=None):
        """demodulation_mapper::demodulate_soft for nof_spans spans of span_len symbols in device memory."""
        _check(self.lib.nrphy_demodulate_soft(self.handle, modulation, nof_spans, span_len, _dptr(d_symbols), _dptr(d_noise_vars),
                                              _dptr(d_llr), _stream(stream)), "nrphy_demodulate_soft")

    def demodulate_soft_host(self, modulation, symbols, noise_vars):
        """One span from host memory: symbols complex64 [n], noise_vars float32 [n] -> int8 [n * bits per symbol]."""
        symbols = np.ascontiguousarray(symbols, dtype=np.complex64)
        noise_vars = np.ascontiguousarray(noise_vars, dtype=np.float32)
        assert symbols.size == noise_vars.size
        out = np.zeros(symbols.size * max(modulation, 1), np.int8)
        _check(self.lib.nrphy_demodulate_soft_host(self.handle, modulation, symbols.size, symbols.ctypes.data,
                                                   noise_vars.ctypes.data, out.ctypes.data), "nrphy_demodulate_soft_host")
        return out

    def grid_put(self, d_grid, nof_ports, nof_subc, entries, stream=None):
        """Sparse host writes into ONE device grid: entries = [(port, symbol, subc, cbf16 word)], later ones win."""
        n = len(entries)
        arr = (abi.GridRe * n)(*[abi.GridRe(*e) for e in entries])
        _check(self.lib.nrphy_grid_put(self.handle, _dptr(d_grid), nof_ports, nof_subc, n, arr, _stream(stream)), "nrphy_grid_put")

    def csi_rs_map(self, cfgs, grid_indices, d_grid, nof_ports, nof_subc, stream=None):
        """nzp_csi_rs_generator::map for a batch of signals into device grids [grid][port][14][subc]."""
        n = len(cfgs)
        arr = (abi.CsiRsCfg * n)(*cfgs)
        idx = (C.c_uint32 * n)(*grid_indices)
        _check(self.lib.nrphy_csi_rs_map(self.handle, n, arr, idx, _dptr(d_grid), nof_ports, nof_subc, _stream(stream)),
               "nrphy_csi_rs_map")

    def csi_rs_map_host(self, cfg, grid):
        """nzp_csi_rs_generator::map into a copy of a host grid [nof_ports][14][nof_subc][2] uint16 (raw cbf16)."""
        out = np.array(grid, dtype=np.uint16, copy=True)
        _check(self.lib.nrphy_csi_rs_map_host(self.handle, C.byref(cfg), out.ctypes.data, out.shape[0], out.shape[2]),
               "nrphy_csi_rs_map_host")
        return out

    # ---- downlink control channels (pdcch_processor, ssb_processor) --------------------------------------------
    def pdcch_process(self, pdus, grid_indices, d_grid, nof_ports, nof_subc, stream=None):
        """pdcch_processor::process for a batch of DCIs into device grids [grid][port][14][subc]."""
        n = len(pdus)
        arr = (abi.PdcchPdu * n)(*pdus)
        idx = (C.c_uint32 * n)(*grid_indices)
        _check(self.lib.nrphy_pdcch_process(self.handle, n, arr, idx, _dptr(d_grid), nof_ports, nof_subc, _stream(stream)),
               "nrphy_pdcch_process")

    def pdcch_process_host(self, pdu, grid):
        """pdcch_processor::process into a copy of a host grid [nof_ports][14][nof_subc][2] uint16 (raw cbf16)."""
        out = np.array(grid, dtype=np.uint16, copy=True)
        _check(self.lib.nrphy_pdcch_process_host(self.handle, C.byref(pdu), out.ctypes.data, out.shape[0], out.shape[2]),
               "nrphy_pdcch_process_host")
        return out

    def pdcch_encode_host(self, payload_bits, rnti, rm_length):
        """pdcch_encoder::encode: payload bits (one per byte) -> rm_length bits (one per byte)."""
        payload = np.ascontiguousarray(payload_bits, dtype=np.uint8)
        out = np.zeros(rm_length, np.uint8)
        _check(self.lib.nrphy_pdcch_encode_host(self.handle, payload.ctypes.data, payload.size, rnti, rm_length,
                                                out.ctypes.data), "nrphy_pdcch_encode_host")
        return out

    def ssb_process(self, pdus, grid_indices, d_grid, nof_ports, nof_subc, stream=None):
        """ssb_processor::process for a batch of SS/PBCH blocks into device grids [grid][port][14][subc]."""
        n = len(pdus)
        arr = (abi.SsbPdu * n)(*pdus)
        idx = (C.c_uint32 * n)(*grid_indices)
        _check(self.lib.nrphy_ssb_process(self.handle, n, arr, idx, _dptr(d_grid), nof_ports, nof_subc, _stream(stream)),
               "nrphy_ssb_process")

    def ssb_process_host(self, pdu, grid):
        """ssb_processor::process into a copy of a host grid [nof_ports][14][nof_subc][2] uint16 (raw cbf16)."""
        out = np.array(grid, dtype=np.uint16, copy=True)
        _check(self.lib.nrphy_ssb_process_host(self.handle, C.byref(pdu), out.ctypes.data, out.shape[0], out.shape[2]),
               "nrphy_ssb_process_host")
        return out

    def pbch_encode_host(self, pdu):
        """pbch_encoder::encode: the 864 rate-matched PBCH bits (one per byte)."""
        out = np.zeros(864, np.uint8)
        _check(self.lib.nrphy_pbch_encode_host(self.handle, C.byref(pdu), out.ctypes.data), "nrphy_pbch_encode_host")
        return out

    # ---- lower-PHY tail (amplitude controller, radio sample format, fronthaul compression) ------------------------
    def amplitude_control(self, cfg, n_buffers, nof_samples, d_in, d_out, d_stats=None, in_stride=None, out_stride=None,
                          stream=None):
        """amplitude_controller::process for n_buffers device buffers of nof_samples complex floats."""
        _check(self.lib.nrphy_amplitude_control(self.handle, C.byref(cfg), n_buffers, nof_samples, _dptr(d_in),
                                                in_stride or nof_samples, _dptr(d_out), out_stride or nof_samples,
                                                _dptr(d_stats), _stream(stream)), "nrphy_amplitude_control")

    def amplitude_control_host(self, cfg, x, metrics=None):
        """One host buffer (complex64) -> (output, abi.AmplitudeMetrics updated in place when given)."""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        out = np.zeros_like(x)
        m = metrics if metrics is not None else abi.AmplitudeMetrics()
        _check(self.lib.nrphy_amplitude_control_host(self.handle, C.byref(cfg), x.ctypes.data, x.size, out.ctypes.data,
                                                     C.byref(m)), "nrphy_amplitude_control_host")
        return out, m

    def iq_convert_ci16_host(self, x, scale):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        out = np.zeros(2 * x.size, np.int16)
        _check(self.lib.nrphy_iq_convert_ci16_host(self.handle, x.ctypes.data, x.size, scale, out.ctypes.data),
               "nrphy_iq_convert_ci16_host")
        return out

    def iq_convert_ci16(self, n_buffers, nof_samples, d_in, scale, d_out, in_stride=None, out_stride=None, stream=None):
        _check(self.lib.nrphy_iq_convert_ci16(self.handle, n_buffers, nof_samples, _dptr(d_in), in_stride or nof_samples,
                                              scale, _dptr(d_out), out_stride or nof_samples, _stream(stream)), "nrphy_iq_convert_ci16")

    def ofh_compress_host(self, cfg, prbs):
        """iq_compressor::compress + serialisation: prbs [nof_prb][12][2] uint16 (raw cbf16) -> bytes."""
        prbs = np.ascontiguousarray(prbs, dtype=np.uint16)
        nof_prb = prbs.size // 24
        out = np.zeros(nof_prb * self.lib.nrphy_ofh_compressed_prb_bytes(C.byref(cfg)), np.uint8)
        _check(self.lib.nrphy_ofh_compress_host(self.handle, C.byref(cfg), nof_prb, prbs.ctypes.data, out.ctypes.data),
               "nrphy_ofh_compress_host")
        return out

    def ofh_compress(self, cfg, n_rows, nof_prb, d_prbs, d_out, row_stride=None, out_row_stride=None, stream=None):
        rec = self.lib.nrphy_ofh_compressed_prb_bytes(C.byref(cfg))
        _check(self.lib.nrphy_ofh_compress(self.handle, C.byref(cfg), n_rows, nof_prb, _dptr(d_prbs), row_stride or 12 * nof_prb,
                                           _dptr(d_out), out_row_stride or rec * nof_prb, _stream(stream)), "nrphy_ofh_compress")

    def pusch_decoder_sizes(self, cfg, n_tb):
        """(soft-buffer bytes per transport block, state bytes of the batch, codeblocks per transport block)."""
        soft, state, scratch, ncb = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        _check(self.lib.nrphy_pusch_decoder_sizes(self.handle, C.byref(cfg), n_tb, C.byref(soft), C.byref(state),
                                                  C.byref(scratch), C.byref(ncb)), "nrphy_pusch_decoder_sizes")
        self._pusch_scratch_bytes = int(scratch.value)
        return int(soft.value), int(state.value), int(ncb.value)

    def _scratch(self, nbytes, d_scratch):
        """The decoder scratch the caller owns: given, or a grow-only device buffer kept by this wrapper (one call in
        flight at a time, as the tests and benchmarks use it)."""
        if d_scratch is not None:
            return d_scratch
        import torch
        cur = getattr(self, "_dec_scratch", None)
        if cur is None or cur.numel() < nbytes:
            torch.cuda.synchronize()
            self._dec_scratch = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device="cuda")
        return self._dec_scratch

    def pusch_decode_batch(self, cfg, n_tb, d_llr, llr_stride, d_soft, d_state, d_tb, tb_stride, d_result, stream=None,
                           d_scratch=None):
        """pusch_decoder for n_tb transport blocks of one configuration, everything resident in HBM."""
        scratch = C.c_uint64(0)
        _check(self.lib.nrphy_pusch_decoder_sizes(self.handle, C.byref(cfg), n_tb, None, None, C.byref(scratch), None),
               "nrphy_pusch_decoder_sizes")
        d_scratch = self._scratch(scratch.value, d_scratch)
        _check(self.lib.nrphy_pusch_decode_batch(self.handle, C.byref(cfg), n_tb, _dptr(d_llr), llr_stride, _dptr(d_soft),
                                                 _dptr(d_state), _dptr(d_scratch), _dptr(d_tb), tb_stride, _dptr(d_result),
                                                 _stream(stream)), "nrphy_pusch_decode_batch")

    def ldpc_decoder_scratch_bytes(self, cfg, n_cb):
        b = C.c_uint64(0)
        _check(self.lib.nrphy_ldpc_decoder_scratch_bytes(self.handle, C.byref(cfg), n_cb, C.byref(b)),
               "nrphy_ldpc_decoder_scratch_bytes")
        return int(b.value)

    def ldpc_decode(self, cfg, n_cb, d_llr, llr_stride, d_out, out_stride, d_iterations=None, stream=None, d_scratch=None):
        """ldpc_decoder::decode for n_cb codeblocks resident in HBM (cfg: abi.LdpcDecoderCfg)."""
        d_scratch = self._scratch(self.ldpc_decoder_scratch_bytes(cfg, max(1, n_cb)), d_scratch)
        _check(self.lib.nrphy_ldpc_decode(self.handle, C.byref(cfg), n_cb, _dptr(d_llr), llr_stride, _dptr(d_out),
                                          out_stride, _dptr(d_iterations) if d_iterations is not None else None,
                                          _dptr(d_scratch), _stream(stream)), "nrphy_ldpc_decode")

    def ldpc_decode_host(self, base_graph, lifting_size, nof_filler, crc_poly, max_iterations, scaling, llr):
        """ldpc_decoder::decode on host spans: returns (iterations or 0, Kb*Zc hard bits one per byte)."""
        llr = np.ascontiguousarray(llr, dtype=np.int8)
        cfg = abi.LdpcDecoderCfg(base_graph, lifting_size, nof_filler, crc_poly, llr.size, max_iterations, scaling)
        k = (22 if base_graph == 1 else 10) * lifting_size
        packed = np.zeros((k + 7) // 8, np.uint8)
        it = C.c_uint32(0)
        _check(self.lib.nrphy_ldpc_decode_host(self.handle, C.byref(cfg), llr.ctypes.data, packed.ctypes.data,
                                               C.byref(it)), "nrphy_ldpc_decode_host")
        return int(it.value), np.unpackbits(packed)[:k]

    def dft(self, size, inverse, batch, d_in, d_out, stream=None):
        _check(self.lib.nrphy_dft_run(self.handle, size, int(inverse), batch, _dptr(d_in), _dptr(d_out), _stream(stream)),
               "nrphy_dft_run")


class PdschPlan:
    """nrphy_pdsch_plan: a batch of PDUs with their grids; run() launches the whole PDSCH path."""

    def __init__(self, ctx, pdus, tb_offsets, grid_indices, nof_grids, nof_ports, nof_subc):
        self.ctx = ctx
        n = len(pdus)
        arr = (abi.PdschPdu * n)(*pdus)
        self._keep = [getattr(p, "_keepalive", None) for p in pdus]
        offs = (C.c_uint64 * n)(*tb_offsets)
        gidx = (C.c_uint32 * n)(*grid_indices)
        h = C.c_void_p()
        _check(ctx.lib.nrphy_pdsch_plan_create(ctx.handle, n, arr, offs, gidx, nof_grids, nof_ports, nof_subc,
                                               C.byref(h)), "nrphy_pdsch_plan_create")
        self.handle = h
        self.nof_grids, self.nof_ports, self.nof_subc = nof_grids, nof_ports, nof_subc

    @property
    def nof_codeblocks(self):
        return int(self.ctx.lib.nrphy_pdsch_plan_nof_codeblocks(self.handle))

    @property
    def codeword_bits(self):
        return int(self.ctx.lib.nrphy_pdsch_plan_codeword_bits(self.handle))

    def codeword_offset(self, pdu):
        return int(self.ctx.lib.nrphy_pdsch_plan_codeword_offset(self.handle, pdu))

    def run(self, d_tb, d_grid, d_cw_rm=None, d_cw_scr=None, zero_grids=True, stream=None):
        _check(self.ctx.lib.nrphy_pdsch_run(self.handle, _dptr(d_tb), _dptr(d_grid), _dptr(d_cw_rm), _dptr(d_cw_scr),
                                            int(zero_grids), _stream(stream)), "nrphy_pdsch_run")

    def enable_timing(self, max_runs, stride=1):
        """Records HIP events around the kernels of the next runs: every `stride`-th one, up to max_runs of them."""
        _check(self.ctx.lib.nrphy_pdsch_plan_enable_timing(self.handle, max_runs), "nrphy_pdsch_plan_enable_timing")
        _check(self.ctx.lib.nrphy_pdsch_plan_timing_stride(self.handle, stride), "nrphy_pdsch_plan_timing_stride")

    def kernel_times(self):
        """Average ms of (tb_crc, codeblock, dmrs, whole run) over the recorded runs, and the run count."""
        ms = (C.c_float * 4)()
        n = C.c_uint32(0)
        _check(self.ctx.lib.nrphy_pdsch_plan_kernel_times(self.handle, ms, C.byref(n)), "nrphy_pdsch_plan_kernel_times")
        return [float(x) for x in ms], int(n.value)

    def close(self):
        if self.handle:
            self.ctx.lib.nrphy_pdsch_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PdschAsyncQueue:
    """nrphy_pdsch_async: up to `depth` PDUs in flight through the host-span seam, completion on a runtime thread."""

    def __init__(self, ctx, depth, nof_ports, nof_subc, max_tb_bytes):
        self.ctx, self.nof_ports, self.nof_subc = ctx, nof_ports, nof_subc
        h = C.c_void_p()
        _check(ctx.lib.nrphy_pdsch_async_create(ctx.handle, depth, nof_ports, nof_subc, max_tb_bytes, C.byref(h)),
               "nrphy_pdsch_async_create")
        self.handle = h
        self._keep = []

    def submit(self, pdu, tb, on_done):
        """on_done(status, grid) runs on a HIP runtime thread; grid is a copy [nof_ports][14][nof_subc][2] uint16.
        Returns False when `depth` PDUs are in flight (retry after a completion)."""
        shape = (self.nof_ports, 14, self.nof_subc, 2)

        def trampoline(user, status, grid_ptr):
            grid = np.ctypeslib.as_array(C.cast(grid_ptr, C.POINTER(C.c_uint16)), shape=shape).copy() if status == 0 else None
            on_done(status, grid)

        cb = abi.PDSCH_DONE_FN(trampoline)
        self._keep.append(cb)
        tb = np.ascontiguousarray(tb, dtype=np.uint8)
        rc = self.ctx.lib.nrphy_pdsch_async_submit(self.handle, C.byref(pdu), tb.ctypes.data, cb, None)
        if rc == abi.ERR_CAPACITY:
            self._keep.pop()
            return False
        _check(rc, "nrphy_pdsch_async_submit")
        return True

    def submit_slot(self, pdus, tbs, on_done):
        """All PDUs of one slot as one operation (nrphy_pdsch_async_submit_slot); otherwise as submit()."""
        shape = (self.nof_ports, 14, self.nof_subc, 2)

        def trampoline(user, status, grid_ptr):
            grid = np.ctypeslib.as_array(C.cast(grid_ptr, C.POINTER(C.c_uint16)), shape=shape).copy() if status == 0 else None
            on_done(status, grid)

        cb = abi.PDSCH_DONE_FN(trampoline)
        self._keep.append(cb)
        n = len(pdus)
        arr = (abi.PdschPdu * n)(*pdus)
        keep = [np.ascontiguousarray(t, dtype=np.uint8) for t in tbs]
        ptrs = (C.c_void_p * n)(*[t.ctypes.data for t in keep])
        rc = self.ctx.lib.nrphy_pdsch_async_submit_slot(self.handle, n, arr, ptrs, cb, None)
        if rc == abi.ERR_CAPACITY:
            self._keep.pop()
            return False
        _check(rc, "nrphy_pdsch_async_submit_slot")
        return True

    def wait(self):
        _check(self.ctx.lib.nrphy_pdsch_async_wait(self.handle), "nrphy_pdsch_async_wait")
        self._keep.clear()

    def wait_slot(self):
        """Blocks while all `depth` operations are in flight."""
        _check(self.ctx.lib.nrphy_pdsch_async_wait_slot(self.handle), "nrphy_pdsch_async_wait_slot")

    def close(self):
        if self.handle:
            self.ctx.lib.nrphy_pdsch_async_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DlSlotPool:
    """nrphy_dl_slots: the downlink slot pipeline -- every channel of a slot written into ONE device-resident grid, the
    whole slot modulated at grid hand-over, IQ served from pinned host memory (pdxch_processor_impl.cpp:47-112)."""

    def __init__(self, ctx, ofdm_cfg, nof_ports, depth, max_tb_bytes, wire_cfg=None):
        self.ctx, self.cfg, self.nof_ports = ctx, ofdm_cfg, nof_ports
        self.nof_subc = 12 * ofdm_cfg.bw_rb
        c = abi.DlSlotsCfg()
        c.ofdm, c.nof_ports, c.depth, c.max_tb_bytes = ofdm_cfg, nof_ports, depth, max_tb_bytes
        c.iq_format = 0 if wire_cfg is None else 1
        if wire_cfg is not None:
            c.wire = wire_cfg
        self.wire = wire_cfg is not None
        h = C.c_void_p()
        _check(ctx.lib.nrphy_dl_slots_create(ctx.handle, C.byref(c), C.byref(h)), "nrphy_dl_slots_create")
        self.handle = h
        self._keep = {}

    def open(self):
        """A free slot with an all-zero grid, or None when all `depth` slots are open."""
        sid = C.c_uint32()
        rc = self.ctx.lib.nrphy_dl_slot_open(self.handle, C.byref(sid))
        if rc == abi.ERR_CAPACITY:
            return None
        _check(rc, "nrphy_dl_slot_open")
        return sid.value

    def close(self, sid):
        _check(self.ctx.lib.nrphy_dl_slot_close(self.handle, sid), "nrphy_dl_slot_close")
        self._keep.pop(sid, None)

    def wait_free(self):
        _check(self.ctx.lib.nrphy_dl_slots_wait_free(self.handle), "nrphy_dl_slots_wait_free")

    def pdsch(self, sid, pdus, tbs):
        n = len(pdus)
        arr = (abi.PdschPdu * n)(*pdus)
        keep = [np.ascontiguousarray(t, dtype=np.uint8) for t in tbs]
        ptrs = (C.c_void_p * n)(*[t.ctypes.data for t in keep])
        return self.ctx.lib.nrphy_dl_slot_pdsch(self.handle, sid, n, arr, ptrs)

    def pdcch(self, sid, pdus):
        arr = (abi.PdcchPdu * len(pdus))(*pdus)
        _check(self.ctx.lib.nrphy_dl_slot_pdcch(self.handle, sid, len(pdus), arr), "nrphy_dl_slot_pdcch")

    def ssb(self, sid, pdus):
        arr = (abi.SsbPdu * len(pdus))(*pdus)
        _check(self.ctx.lib.nrphy_dl_slot_ssb(self.handle, sid, len(pdus), arr), "nrphy_dl_slot_ssb")

    def csi_rs(self, sid, cfgs):
        arr = (abi.CsiRsCfg * len(cfgs))(*cfgs)
        _check(self.ctx.lib.nrphy_dl_slot_csi_rs(self.handle, sid, len(cfgs), arr), "nrphy_dl_slot_csi_rs")

    def put(self, sid, entries):
        arr = (abi.GridRe * len(entries))(*entries)
        _check(self.ctx.lib.nrphy_dl_slot_put(self.handle, sid, len(entries), arr), "nrphy_dl_slot_put")

    def load_grid(self, sid, grid):
        grid = np.ascontiguousarray(grid, dtype=np.uint16)
        assert grid.size == self.nof_ports * 14 * self.nof_subc * 2
        _check(self.ctx.lib.nrphy_dl_slot_load_grid(self.handle, sid, grid.ctypes.data), "nrphy_dl_slot_load_grid")

    def modulate(self, sid, subframe_slot_index, on_done=None):
        """on_done(status, slot_id) runs on a HIP runtime thread when the slot's IQ is in pinned host memory."""
        cb = None
        if on_done is not None:
            cb = abi.DL_SLOT_DONE_FN(lambda user, status, slot_id: on_done(status, slot_id))
            self._keep[sid] = cb
        return self.ctx.lib.nrphy_dl_slot_modulate(self.handle, sid, subframe_slot_index, cb, None)

    def poll(self, sid):
        return self.ctx.lib.nrphy_dl_slot_poll(self.handle, sid)

    def wait(self, sid):
        return self.ctx.lib.nrphy_dl_slot_wait(self.handle, sid)

    def iq(self, sid, port):
        """A copy of the slot's samples of `port`: complex64, or int16 pairs [n][2] for a wire-format pool."""
        n = C.c_uint32()
        ptr = self.ctx.lib.nrphy_dl_slot_iq(self.handle, sid, port, C.byref(n))
        assert ptr
        if self.wire:
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int16)), shape=(n.value, 2)).copy()
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n.value, 2)).copy().view(np.complex64).reshape(-1)

    def amplitude_stats(self, sid, port):
        """Wire-format pools: a copy of the amplitude controller's raw measurements of the slot's buffer of `port`."""
        p = self.ctx.lib.nrphy_dl_slot_amplitude_stats(self.handle, sid, port)
        assert p
        st = abi.AmplitudeStats()
        C.memmove(C.byref(st), p, C.sizeof(abi.AmplitudeStats))
        return st

    def read_grid(self, sid):
        grid = np.zeros((self.nof_ports, 14, self.nof_subc, 2), dtype=np.uint16)
        _check(self.ctx.lib.nrphy_dl_slot_read_grid(self.handle, sid, grid.ctypes.data), "nrphy_dl_slot_read_grid")
        return grid

    def destroy(self):
        if self.handle:
            self.ctx.lib.nrphy_dl_slots_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class OfdmPlan:
    """nrphy_ofdm_plan: ofdm_modulator_configuration + port count."""

    def __init__(self, ctx, cfg, nof_ports):
        self.ctx = ctx
        self.cfg = cfg
        self.nof_ports = nof_ports
        h = C.c_void_p()
        _check(ctx.lib.nrphy_ofdm_plan_create(ctx.handle, C.byref(cfg), nof_ports, C.byref(h)),
               "nrphy_ofdm_plan_create")
        self.handle = h
        self.slot_stride = int(ctx.lib.nrphy_ofdm_plan_slot_stride(h))

    def run(self, nof_grids, d_grid, d_iq, d_slot_index=None, stream=None):
        _check(self.ctx.lib.nrphy_ofdm_run(self.handle, nof_grids, _dptr(d_grid), _dptr(d_slot_index), _dptr(d_iq),
                                           _stream(stream)), "nrphy_ofdm_run")

    def run_ci16(self, nof_grids, d_grid, wire_cfg, d_iq16, d_slot_index=None, d_stats=None, stream=None):
        """nrphy_ofdm_run with the amplitude controller and the complex int16 conversion fused into the store."""
        _check(self.ctx.lib.nrphy_ofdm_run_ci16(self.handle, nof_grids, _dptr(d_grid), _dptr(d_slot_index), C.byref(wire_cfg),
                                                _dptr(d_iq16), _dptr(d_stats), _stream(stream)), "nrphy_ofdm_run_ci16")

    def enable_timing(self, max_runs, stride=1):
        _check(self.ctx.lib.nrphy_ofdm_plan_enable_timing(self.handle, max_runs), "nrphy_ofdm_plan_enable_timing")
        _check(self.ctx.lib.nrphy_ofdm_plan_timing_stride(self.handle, stride), "nrphy_ofdm_plan_timing_stride")

    def kernel_time(self):
        ms = C.c_float(0)
        n = C.c_uint32(0)
        _check(self.ctx.lib.nrphy_ofdm_plan_kernel_time(self.handle, C.byref(ms), C.byref(n)),
               "nrphy_ofdm_plan_kernel_time")
        return float(ms.value), int(n.value)

    def modulate_symbol_host(self, grid, port, symbol_index):
        grid = np.ascontiguousarray(grid, dtype=np.uint16)
        n = symbol_size(self.cfg, symbol_index)
        out = np.zeros(n, np.complex64)
        _check(self.ctx.lib.nrphy_ofdm_modulate_symbol_host(self.handle, grid.ctypes.data, port, symbol_index,
                                                            out.ctypes.data, n), "nrphy_ofdm_modulate_symbol_host")
        return out

    def demod_run(self, nof_grids, d_iq, d_grid, d_slot_index=None, window_offset=0, stream=None):
        """ofdm_slot_demodulator::demodulate for every port of nof_grids slots (device buffers)."""
        _check(self.ctx.lib.nrphy_ofdm_demod_run(self.handle, nof_grids, _dptr(d_iq), _dptr(d_slot_index),
                                                 window_offset, _dptr(d_grid), _stream(stream)), "nrphy_ofdm_demod_run")

    def demodulate_slot_host(self, iq, slot_index, window_offset=0):
        """Host-span form: iq [nof_ports][slot samples] complex64 -> grid [nof_ports][14][12*bw_rb][2] uint16."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        grid = np.zeros((self.nof_ports, 14, 12 * self.cfg.bw_rb, 2), np.uint16)
        _check(self.ctx.lib.nrphy_ofdm_demodulate_slot_host(self.handle, iq.ctypes.data, slot_index, window_offset,
                                                            grid.ctypes.data), "nrphy_ofdm_demodulate_slot_host")
        return grid

    def demodulate_symbol_host(self, samples, symbol_index, window_offset=0):
        """ofdm_symbol_demodulator::demodulate for one symbol of one port -> [12*bw_rb][2] uint16."""
        samples = np.ascontiguousarray(samples, dtype=np.complex64)
        row = np.zeros((12 * self.cfg.bw_rb, 2), np.uint16)
        _check(self.ctx.lib.nrphy_ofdm_demodulate_symbol_host(self.handle, samples.ctypes.data, samples.size,
                                                              symbol_index, window_offset, row.ctypes.data),
               "nrphy_ofdm_demodulate_symbol_host")
        return row

    def modulate_slot_host(self, grid, slot_index, out=None):
        """ofdm_slot_modulator::modulate for every port of one host grid: [nof_ports][slot samples] complex64."""
        grid = np.ascontiguousarray(grid, dtype=np.uint16)
        n = slot_size(self.cfg, slot_index)
        if out is None:
            out = np.zeros((self.nof_ports, n), np.complex64)
        _check(self.ctx.lib.nrphy_ofdm_modulate_slot_host(self.handle, grid.ctypes.data, slot_index, out.ctypes.data),
               "nrphy_ofdm_modulate_slot_host")
        return out

    def close(self):
        if self.handle:
            self.ctx.lib.nrphy_ofdm_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- host-only helpers (no GPU needed) ---------------------------------------------------------------------------
def validate(pdu):
    return int(load().nrphy_pdsch_validate(C.byref(pdu)))


def derive(pdu):
    d = abi.PdschDerived()
    _check(load().nrphy_pdsch_derive(C.byref(pdu), C.byref(d)), "nrphy_pdsch_derive")
    return d.as_dict()


def tbs_calculate(nof_symb_sh, nof_dmrs_prb, nof_oh_prb, qm, rate_x1024, nof_layers, n_prb):
    return int(load().nrphy_tbs_calculate(nof_symb_sh, nof_dmrs_prb, nof_oh_prb, qm, float(rate_x1024), nof_layers,
                                          n_prb))


def symbol_size(cfg, symbol_index):
    return int(load().nrphy_ofdm_symbol_size(C.byref(cfg), symbol_index))


def slot_size(cfg, slot_index):
    return int(load().nrphy_ofdm_slot_size(C.byref(cfg), slot_index))
