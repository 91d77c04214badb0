#!/usr/bin/env python3
"""Secondary measurement, BASELINE config 5 ("PUSCH receive path add-on: OFDM demod + LDPC min-sum decode, 8 iterations,
100 MHz, 1 MI355X"): the three receive-side kernels built so far on one batch of slots, inputs resident in HBM.
A step = OFDM demodulation of `slots` 100 MHz slots (4 receive ports) + rate dematching and decoding of the 104
codeblocks (BG1, Zc 384, 256-QAM, E = 8960) each slot's transport block has.  Equalisation and soft demodulation sit
between the two in a real receiver and are not built; the LLRs are synthetic noisy codewords of the GPU encoder.
Prints one JSON line in bench.py's schema.  Usage (GPU box, repository root):
python3 profiles/rx_chain_bench.py [--slots 64] [--iterations 8] [--steps 10]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=64)
    ap.add_argument("--iterations", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    import torch
    import backends
    import cases
    abi, lib = backends.abi, backends.pkg.lib
    ctx = lib.Context(0)
    _, ports, subc, ocfg = cases.baseline_config(3)
    plan = lib.OfdmPlan(ctx, ocfg, ports)
    slots = args.slots
    bg, zc, e, nf = 1, 384, 8960, 72
    k, n, n_cb = 22 * zc, 66 * zc, 104 * slots
    rng = np.random.default_rng(5)
    # valid codeblocks (CRC24B) so that early stop fires as it does on a live link
    oracle = backends.oracle()
    base = []
    for _ in range(8):
        payload = rng.integers(0, 2, k - nf - 24, dtype=np.uint8)
        crc = oracle.crc_bits(0x24B, payload)
        base.append(np.packbits(np.concatenate([payload, [(crc >> (23 - b)) & 1 for b in range(24)],
                                                np.zeros(nf, np.uint8)]).astype(np.uint8)))
    msgs = np.stack([base[i % 8] for i in range(n_cb)])
    d_msg = torch.from_numpy(msgs).cuda()
    enc_bits = e + nf + 8  # rate matching skips the filler bits
    d_cb = torch.zeros((n_cb, (enc_bits + 7) // 8), dtype=torch.uint8, device="cuda")
    ctx.ldpc_encode(bg, zc, d_msg, k // 8, enc_bits, d_cb, d_cb.shape[1], n_cb)
    torch.cuda.synchronize()
    cb = np.unpackbits(d_cb.cpu().numpy(), axis=1)
    nof_sys = 20 * zc
    bits = np.concatenate([cb[:, : nof_sys - nf], cb[:, nof_sys:]], axis=1)[:, :e].astype(np.float32)
    # the rate matcher's bit interleaver is undone by the dematcher: interleave here the way the transmitter does
    cols = e // 8
    tx = bits.reshape(n_cb, 8, cols).transpose(0, 2, 1).reshape(n_cb, e)
    llr = np.clip(np.rint((1 - 2 * tx) * 20 + rng.normal(0, 7.0, tx.shape)), -120, 120).astype(np.int8)
    d_llr = torch.from_numpy(llr).cuda()
    d_soft = torch.zeros((n_cb, n), dtype=torch.int8, device="cuda")
    d_out = torch.zeros((n_cb, k // 8), dtype=torch.uint8, device="cuda")
    d_its = torch.zeros((n_cb,), dtype=torch.int32, device="cuda")
    d_iq = torch.randn((slots, ports, plan.slot_stride, 2), dtype=torch.float32, device="cuda")
    d_grid = torch.zeros((slots, ports, 14, subc), dtype=torch.int32, device="cuda")
    d_slot = torch.tensor([i % 2 for i in range(slots)], dtype=torch.int32, device="cuda")
    dm = abi.LdpcRateDematcherCfg(bg, zc, 0, 8, 0, nf, e)
    dec = abi.LdpcDecoderCfg(bg, zc, nf, 0x24B, n, args.iterations, 0.8)
    s = torch.cuda.Stream()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    kernel_ms = {"ofdm_demod": 0.0, "rate_dematch": 0.0, "ldpc_decode": 0.0}

    def step(timed):
        if timed:
            ev[0].record(s)
        plan.demod_run(slots, d_iq, d_grid, d_slot_index=d_slot, stream=s.cuda_stream)
        if timed:
            ev[1].record(s)
        ctx.ldpc_rate_dematch(dm, n_cb, d_llr, e, d_soft, n, True, s.cuda_stream)
        if timed:
            ev[2].record(s)
        ctx.ldpc_decode(dec, n_cb, d_soft, n, d_out, k // 8, d_its, s.cuda_stream)
        if timed:
            ev[3].record(s)

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    assert int(d_its.min()) >= 1, "every codeblock must decode at this SNR"
    assert np.array_equal(d_out.cpu().numpy()[:16], msgs[:16]), "decoded messages differ from what was sent"
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(args.steps):
        step(False)
    b.record(s)
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / args.steps
    step(True)
    torch.cuda.synchronize()
    for i, name in enumerate(kernel_ms):
        kernel_ms[name] = ev[i].elapsed_time(ev[i + 1])
    alg = n_cb * (n + k // 8)  # decoder: soft buffer in, packed message out
    print(json.dumps({
        "metric": "pusch_rx_slots_per_second", "value": slots / ms * 1e3, "unit": "slots/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int8 LLR / f32 IQ", "data": "synthetic",
        "config": {"workload": "BASELINE config 5: OFDM demod (4096, 273 PRB, %d ports) + LDPC rate dematch + decode, "
                               "104 CB/slot BG1 Zc384 E8960, max %d iterations, CRC24B early stop" % (ports, args.iterations),
                   "slots_per_step": slots, "codeblocks_per_step": n_cb},
        "kernel_ms": kernel_ms, "mean_iterations": float(d_its.float().mean()),
        "info_gbps": n_cb * (k - nf - 24) / ms * 1e-6,
        "roofline": {"bound": "hbm", "kernel": "ldpc_decode_kernel", "achieved": alg / kernel_ms["ldpc_decode"] * 1e-6,
                     "peak": 8000.0, "unit": "GB/s", "frac": alg / kernel_ms["ldpc_decode"] * 1e-6 / 8000.0,
                     "traffic": None, "note": "VALU-issue bound, see DESIGN.md section 5"},
    }))


if __name__ == "__main__":
    main()
