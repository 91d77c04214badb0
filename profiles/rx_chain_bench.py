#!/usr/bin/env python3
"""Secondary measurement, BASELINE config 5 ("PUSCH receive path add-on: OFDM demod + LDPC min-sum decode, 8 iterations,
100 MHz, 1 MI355X"): the receive-side kernels built so far on one batch of slots, everything resident in HBM.
A step = OFDM demodulation of `slots` 100 MHz slots (4 receive ports) + the soft demodulator on the slot's 256-QAM symbols
(nrphy_demodulate_soft) + soft-bit descrambling (nrphy_llr_descramble) + the transport-block decoder on the config-3
transport block of each slot (nrphy_pusch_decode_batch: rate dematching of its 104 codeblocks, LDPC decoding with CRC24B
early stop, concatenation, TB CRC24A).  The transmitter is this library's PDSCH path (its scrambled codeword tap); channel
estimation and equalisation, which sit between the OFDM demodulator and the soft demodulator in a receiver, are not built:
the equalised symbols are the scrambled codeword through the 256-QAM map of TS 38.211 Section 5.1.6 plus white Gaussian
noise, with the true noise variance handed to the soft demodulator.  Every decoded transport block is compared with what
was sent.  Prints one JSON line in bench.py's schema.  Usage (GPU box, repository root):
python3 profiles/rx_chain_bench.py [--slots 64] [--iterations 8] [--steps 10] [--snr-db 32]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=256)
    ap.add_argument("--iterations", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--snr-db", type=float, default=32.0)
    print(json.dumps(run(ap.parse_args())))


def run(args):
    """One measurement of the receive chain; `args` carries slots, iterations, steps, warmup, snr_db.  Returns the bench line
    as a dict (bench.py embeds it as its config-5 entry)."""
    snr_db = getattr(args, "snr_db", 32.0)
    import torch
    import backends
    import cases
    abi, lib = backends.abi, backends.pkg.lib
    ctx = lib.Context(0)
    pdu, ports, subc, ocfg = cases.baseline_config(3)
    slots = args.slots
    d = lib.derive(pdu)
    G, C, tb_size = d["codeword_bits"], d["nof_codeblocks"], pdu.tb_size_bytes
    # transmit side: the PDSCH plan with its scrambled codeword tap (what the air interface carries)
    pdus = [cases.baseline_config(3, slot_index=i % 20)[0] for i in range(slots)]
    tb_stride = (tb_size + 3) & ~3
    plan = lib.PdschPlan(ctx, pdus, [i * tb_stride for i in range(slots)], list(range(slots)), slots, ports, subc)
    d_tb = torch.randint(0, 256, (slots, tb_stride), dtype=torch.uint8, device="cuda")
    d_cw = torch.zeros((plan.codeword_bits + 7) // 8 + 64, dtype=torch.uint8, device="cuda")
    plan.run(d_tb.reshape(-1), None, d_cw_scr=d_cw)
    ctx.synchronize()
    torch.cuda.synchronize()
    offs = [plan.codeword_offset(i) for i in range(slots)]
    assert all(o % 8 == 0 for o in offs)
    cw = d_cw.cpu().numpy()
    bits = np.stack([np.unpackbits(cw[o // 8: o // 8 + (G + 7) // 8])[:G] for o in offs])
    # equalised symbols: 256-QAM map of the scrambled codeword (TS 38.211 Section 5.1.6) + AWGN at the given SNR
    assert pdu.qm == 8
    nsym = G // 8
    sgn = 1.0 - 2.0 * torch.from_numpy(bits).cuda().reshape(slots, nsym, 8).float()
    re = sgn[..., 0] * (8 - sgn[..., 2] * (4 - sgn[..., 4] * (2 - sgn[..., 6])))
    im = sgn[..., 1] * (8 - sgn[..., 3] * (4 - sgn[..., 5] * (2 - sgn[..., 7])))
    noise_var = float(10.0 ** (-snr_db / 10.0))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    d_sym = torch.stack((re, im), dim=-1) / float(np.sqrt(170.0)) + torch.randn(
        (slots, nsym, 2), device="cuda", generator=gen) * float(np.sqrt(noise_var / 2))
    d_sym = d_sym.contiguous()
    d_nv = torch.full((slots, nsym), noise_var, dtype=torch.float32, device="cuda")
    d_llr_scr = torch.zeros((slots, G), dtype=torch.int8, device="cuda")   # soft bits, still scrambled
    del sgn, re, im
    d_llr = torch.empty_like(d_llr_scr)
    d_c_init = torch.tensor([(p.rnti << 15) + p.n_id for p in pdus], dtype=torch.int32, device="cuda")  # TS 38.211 7.3.1.1, q = 0
    cfg = abi.PuschDecoderCfg(pdu.ldpc_base_graph, pdu.qm, 0, pdu.nof_layers, d["n_ref"], tb_size, G // pdu.qm,
                              args.iterations, 1, 1)
    soft_bytes, state_bytes, ncb = ctx.pusch_decoder_sizes(cfg, slots)
    d_soft = torch.zeros((slots, soft_bytes), dtype=torch.int8, device="cuda")
    d_state = torch.zeros((state_bytes,), dtype=torch.uint8, device="cuda")
    d_out = torch.zeros((slots, tb_stride), dtype=torch.uint8, device="cuda")
    d_res = torch.zeros((slots, 4), dtype=torch.int32, device="cuda")
    oplan = lib.OfdmPlan(ctx, ocfg, ports)
    d_iq = torch.randn((slots, ports, oplan.slot_stride, 2), dtype=torch.float32, device="cuda")
    d_grid = torch.zeros((slots, ports, 14, subc), dtype=torch.int32, device="cuda")
    d_slot = torch.tensor([i % 2 for i in range(slots)], dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]

    def step(timed):
        if timed:
            ev[0].record(s)
        oplan.demod_run(slots, d_iq, d_grid, d_slot_index=d_slot, stream=s.cuda_stream)
        if timed:
            ev[1].record(s)
        ctx.demodulate_soft(8, slots, nsym, d_sym, d_nv, d_llr_scr, s.cuda_stream)
        if timed:
            ev[2].record(s)
        ctx.llr_descramble(d_c_init, slots, G, d_llr_scr, G, d_llr, G, s.cuda_stream)
        if timed:
            ev[3].record(s)
        ctx.pusch_decode_batch(cfg, slots, d_llr, G, d_soft, d_state, d_out, tb_stride, d_res, s.cuda_stream)
        if timed:
            ev[4].record(s)

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    res = d_res.cpu().numpy()
    assert res[:, 0].all(), "every transport block must decode at this SNR (%d of %d did)" % (int(res[:, 0].sum()), slots)
    assert torch.equal(d_out[:, :tb_size], d_tb[:, :tb_size]), "decoded transport blocks differ from what was sent"
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(args.steps):
        step(False)
    b.record(s)
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / args.steps
    step(True)
    torch.cuda.synchronize()
    kernel_ms = {"ofdm_demod": ev[0].elapsed_time(ev[1]), "demodulate_soft": ev[1].elapsed_time(ev[2]),
                 "llr_descramble": ev[2].elapsed_time(ev[3]), "pusch_decode_batch": ev[3].elapsed_time(ev[4])}
    demod_bytes = slots * nsym * (12 + 8)   # symbol + noise variance in, 8 soft bits out
    n_cb = slots * C
    alg = n_cb * d["full_length"] + slots * tb_size  # decoder: soft buffers in, transport blocks out
    return {
        "metric": "pusch_rx_slots_per_second", "value": slots / ms * 1e3, "unit": "slots/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int8 LLR / f32 IQ", "data": "synthetic",
        "config": {"workload": "BASELINE config 5: OFDM demod (4096, 273 PRB, %d ports) + soft demodulator (256-QAM) + descrambling + "
                               "UL-SCH decoder (rate dematch, LDPC "
                               "max %d iterations with CRC24B early stop, concatenation, TB CRC) on config-3 transport "
                               "blocks (868584 bit, 104 CB, BG1 Zc384)" % (ports, args.iterations),
                   "slots_per_step": slots, "codeblocks_per_step": n_cb, "snr_dB": snr_db},
        "kernel_ms": kernel_ms, "mean_iterations": float(res[:, 2].sum()) / n_cb, "tb_crc_ok": int(res[:, 0].sum()),
        "info_gbps": slots * tb_size * 8 / ms * 1e-6,
        "demodulate_soft_roofline": {"bound": "hbm", "kernel": "demodulate_soft_kernel", "unit": "GB/s", "peak": 8000.0,
                                     "achieved": demod_bytes / kernel_ms["demodulate_soft"] * 1e-6,
                                     "frac": demod_bytes / kernel_ms["demodulate_soft"] * 1e-6 / 8000.0,
                                     "algorithmic_bytes_per_launch": demod_bytes},
        "roofline": {"bound": "hbm", "kernel": "ldpc_decode_kernel", "achieved": alg / kernel_ms["pusch_decode_batch"] * 1e-6,
                     "peak": 8000.0, "unit": "GB/s", "frac": alg / kernel_ms["pusch_decode_batch"] * 1e-6 / 8000.0,
                     "traffic": None, "note": "VALU-issue bound, see DESIGN.md section 5"},
        "verified": "every transport block decoded (CRC24A) and equal to what was sent",
    }


if __name__ == "__main__":
    main()
