#!/usr/bin/env python3
"""Secondary measurement, BASELINE config 5 ("PUSCH receive path add-on: OFDM demod + LDPC min-sum decode (BG1/BG2, 8 iters),
100 MHz, 1 MI355X"): the receive-side kernels on one batch of slots, everything resident in HBM.
A step = OFDM demodulation of `slots` 100 MHz slots (4 receive ports) + the soft demodulator on the slot's symbols
(nrphy_demodulate_soft) + soft-bit descrambling (nrphy_llr_descramble) + the transport-block decoder on the transport block
of each slot (nrphy_pusch_decode_batch: rate dematching of its codeblocks, LDPC decoding, concatenation, TB CRC24A).  The
transmitter is this library's PDSCH path (its scrambled codeword tap); channel estimation and equalisation, which sit
between the OFDM demodulator and the soft demodulator in a receiver, are not built: the equalised symbols are the scrambled
codeword through the QPSK / 256-QAM map of TS 38.211 Section 5.1 plus white Gaussian noise, with the true noise variance
handed to the soft demodulator.  Every transport block whose CRC holds is compared with what was sent.

Legs (`run_all`, what bench.py embeds as its config-5 entry):
  bg1_fixed8      BG1 (config-3 transport blocks: 868,584 bit, 104 codeblocks, Zc 384), 8 iterations, NO early stop -- the
                  workload as BASELINE.json words it (ldpc_decoder_impl.cpp:60-143 with use_early_stop = false)
  bg2_fixed8      BG2 (full-band QPSK R = 120/1024, 4 layers: 27,144 bit, 8 codeblocks, Zc 352), 8 iterations, no early stop
  bg1_threshold   BG1 with CRC24B early stop at an SNR near the decoding threshold (several iterations per codeblock)
  bg1_early_stop  BG1 with early stop at 32 dB (the easy case: 1.6 iterations; the round-2 entry)
Prints one JSON line in bench.py's schema.  Usage (GPU box, repository root):
python3 profiles/rx_chain_bench.py [--slots 1024] [--iterations 8] [--steps 10] [--snr-db 32] [--leg bg1|bg2] [--no-early-stop]
python3 profiles/rx_chain_bench.py --sweep bg1 28 32 0.5     # mean iterations / failures against SNR"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

# SNR of the near-threshold leg and of the BG2 leg (chosen with --sweep, profiles/r03_rx_snr_sweeps.txt: every transport block
# still decodes within 8 iterations -- a dB lower none does -- and the mean iteration count is 3.8 (BG1) / 4.7 (BG2)).
SNR_THRESHOLD_BG1 = 26.5
SNR_FIXED_BG2 = -4.0
# Slots per step: the batch of the transmit-side headline (bench.py).  One box, BG1 at 8 fixed iterations: 256 slots per step
# 59.0 k slots/s, 512 -> 60.2 k, 1024 -> 61.1 k (the launch tails of the small kernels around the decoder amortise).
RX_SLOTS = 1024


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=RX_SLOTS)
    ap.add_argument("--iterations", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--snr-db", type=float, default=32.0)
    ap.add_argument("--leg", default="bg1", choices=["bg1", "bg2"])
    ap.add_argument("--no-early-stop", action="store_true")
    ap.add_argument("--allow-failures", action="store_true", help="do not insist that every transport block decodes")
    ap.add_argument("--all", action="store_true", help="the four legs bench.py reports")
    ap.add_argument("--cpu-baseline", action="store_true", help="time the reference's CPU decoding chain on the same soft bits")
    ap.add_argument("--sweep", nargs=4, metavar=("LEG", "FROM", "TO", "STEP"), help="mean iterations against SNR")
    args = ap.parse_args()
    if args.sweep:
        leg, lo, hi, st = args.sweep[0], float(args.sweep[1]), float(args.sweep[2]), float(args.sweep[3])
        snr = lo
        while snr <= hi + 1e-9:
            r = run(argparse.Namespace(slots=64, iterations=args.iterations, steps=2, warmup=1, snr_db=snr, leg=leg,
                                       early_stop=True, require_all=False))
            print("%s snr %6.2f dB: mean iterations %.2f, tb_crc_ok %d / %d, %.0f slots/s" % (
                leg, snr, r["mean_iterations"], r["tb_crc_ok"], 64, r["value"]), flush=True)
            snr += st
        return
    if args.all:
        print(json.dumps(run_all(args.steps, args.warmup)))
        return
    args.early_stop = not args.no_early_stop
    args.require_all = not args.allow_failures
    print(json.dumps(run(args)))


def leg_pdu(leg, slot_index=0):
    """(pdu, ports, subcarriers, OFDM configuration) of a leg: "bg1" = BASELINE config 3; "bg2" = the same 100 MHz grid
    and allocation with QPSK R = 120/1024 (the rate of BASELINE config 1) on 4 layers, which selects base graph 2."""
    import cases
    pdu, ports, subc, ocfg = cases.baseline_config(3, slot_index=slot_index)
    if leg == "bg2":
        abi = cases.abi
        tb_bits = cases.tbs(12, 36, 2, 120, 4, 270)
        pdu = abi.make_pdu(slot_index=slot_index, rnti=1, n_id=0, bwp_start_rb=0, bwp_size_rb=273, qm=2,
                           dmrs_symbols=(2, 7, 11), nof_cdm_groups_without_data=2, prb_start=0, prb_count=270,
                           start_symbol=0, nof_symbols=12, base_graph=2, precoding=cases.codebook("four_layer_four_ports_0_0"),
                           tb_size_bytes=tb_bits // 8)
    return pdu, ports, subc, ocfg


def run_all(steps, warmup):
    """The config-5 entry of bench.py: the four legs of the module docstring; the entry's headline `value` is the workload as
    BASELINE.json states it (BG1, 8 iterations, no early stop)."""
    def leg(name, **kw):
        a = dict(slots=RX_SLOTS, iterations=8, steps=steps, warmup=warmup, snr_db=32.0, leg="bg1", early_stop=True, require_all=True,
                 cpu_baseline=True)
        a.update(kw)
        return run(argparse.Namespace(**a))
    out = leg("bg1_fixed8", early_stop=False)
    out["legs"] = {
        "bg2_fixed8": leg("bg2_fixed8", leg="bg2", early_stop=False, snr_db=SNR_FIXED_BG2),
        "bg1_threshold": leg("bg1_threshold", snr_db=SNR_THRESHOLD_BG1, require_all=False),
        "bg1_early_stop": leg("bg1_early_stop"),
    }
    return out


def cpu_decode_baseline(pdu, d, iterations, early_stop, llr, budget_s=5.0):
    """The reference's CPU receive-side coding chain on the same soft bits (one transport block's, as the device chain hands them
    to its decoder): pusch_decoder_impl with ldpc_rate_dematcher_avx2_impl + ldpc_decoder_avx2 (oracle/_ref, compiled from the
    reference's sources: ref_bench_pusch_decode), T worker threads each with its own decoder instance, batches of transport
    blocks per thread -- the scheme of the reference's benchmarks (ldpc_decoder_benchmark.cpp:36-182,
    pdsch_processor_benchmark.cpp:684-737).  Bounded: about `budget_s` seconds.  None when the compiled reference is absent."""
    import ctypes as C
    import backends
    r = backends.ref()
    if r is None or not hasattr(r.lib, "ref_bench_pusch_decode"):
        return None
    fn = r.lib.ref_bench_pusch_decode
    fn.restype = C.c_double
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    nok = C.c_uint32(0)

    def go(threads, reps):
        return fn(C.c_uint32(pdu.ldpc_base_graph), C.c_uint32(pdu.qm), C.c_uint32(0), C.c_uint32(pdu.nof_layers), C.c_uint32(d["n_ref"]),
                  C.c_uint32(pdu.tb_size_bytes), C.c_uint32(iterations), C.c_int(1 if early_stop else 0), C.c_uint32(d["nof_codeblocks"]),
                  llr.ctypes.data_as(C.c_void_p), C.c_uint32(llr.size), C.c_uint(threads), C.c_uint(reps), C.c_int(1), C.byref(nok))

    def cpu_quota():
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            return None if quota == "max" else float(quota) / float(period)
        except Exception:
            return None

    quota, affinity = cpu_quota(), len(os.sched_getaffinity(0))
    many = max(2, min(affinity, int(quota) if quota else affinity, 64))
    t1 = go(1, 1)                                   # calibration (and page faults): seconds per transport block, one thread
    out = {}
    for threads, share in ((many, 0.6), (1, 0.4)):
        reps = int(max(1, min(2000, budget_s * share / max(t1, 1e-4))))
        dt = go(threads, reps)
        out[threads] = {"threads": threads, "slots_per_sec": round(threads * reps / dt, 2), "tb_per_thread": reps,
                        "tb_crc_ok": int(nok.value), "of": threads * reps, "seconds": round(dt, 2)}
    best = max(out.values(), key=lambda e: e["slots_per_sec"])
    return {"value": best["slots_per_sec"], "unit": "slots/s", "cores": best["threads"], "kind": "reference",
            "all_threads": out[many], "one_thread": out[1], "affinity_cpus": affinity, "cgroup_cpu_quota": quota,
            "sample": "one transport block's soft bits (slot 0 of the device leg, after descrambling) through the reference's "
                      "pusch_decoder_impl (AVX2 rate dematcher + AVX2 LDPC decoder, %d iterations%s, CRCs), %d threads x %d and 1 x %d "
                      "transport blocks; the coding chain only -- the device leg's value also contains OFDM demodulation, soft "
                      "demodulation and descrambling" % (iterations, " with early stop" if early_stop else ", no early stop", many,
                                                         out[many]["tb_per_thread"], out[1]["tb_per_thread"])}


def run(args):
    """One measurement of the receive chain; `args` carries slots, iterations, steps, warmup, snr_db and optionally leg
    ("bg1" / "bg2"), early_stop, require_all.  Returns the bench line as a dict."""
    snr_db = getattr(args, "snr_db", 32.0)
    leg = getattr(args, "leg", "bg1")
    early_stop = bool(getattr(args, "early_stop", True))
    require_all = bool(getattr(args, "require_all", True))
    import torch
    import backends
    abi, lib = backends.abi, backends.pkg.lib
    ctx = lib.Context(0)
    pdu, ports, subc, ocfg = leg_pdu(leg)
    slots = args.slots
    d = lib.derive(pdu)
    G, C, tb_size, qm = d["codeword_bits"], d["nof_codeblocks"], pdu.tb_size_bytes, pdu.qm
    # transmit side: the PDSCH plan with its scrambled codeword tap (what the air interface carries)
    pdus = [leg_pdu(leg, slot_index=i % 20)[0] for i in range(slots)]
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
    plan.close()
    # equalised symbols: QPSK / 256-QAM map of the scrambled codeword (TS 38.211 Sections 5.1.3, 5.1.6) + AWGN at the given SNR
    nsym = G // qm
    sgn = 1.0 - 2.0 * torch.from_numpy(bits).cuda().reshape(slots, nsym, qm).float()
    if qm == 8:
        re = sgn[..., 0] * (8 - sgn[..., 2] * (4 - sgn[..., 4] * (2 - sgn[..., 6]))) / float(np.sqrt(170.0))
        im = sgn[..., 1] * (8 - sgn[..., 3] * (4 - sgn[..., 5] * (2 - sgn[..., 7]))) / float(np.sqrt(170.0))
    else:
        assert qm == 2
        re, im = sgn[..., 0] / float(np.sqrt(2.0)), sgn[..., 1] / float(np.sqrt(2.0))
    noise_var = float(10.0 ** (-snr_db / 10.0))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    d_sym = torch.stack((re, im), dim=-1) + torch.randn((slots, nsym, 2), device="cuda", generator=gen) * float(
        np.sqrt(noise_var / 2))
    d_sym = d_sym.contiguous()
    d_nv = torch.full((slots, nsym), noise_var, dtype=torch.float32, device="cuda")
    d_llr_scr = torch.zeros((slots, G), dtype=torch.int8, device="cuda")   # soft bits, still scrambled
    del sgn, re, im
    d_llr = torch.empty_like(d_llr_scr)
    d_c_init = torch.tensor([(p.rnti << 15) + p.n_id for p in pdus], dtype=torch.int32, device="cuda")  # TS 38.211 7.3.1.1, q = 0
    cfg = abi.PuschDecoderCfg(pdu.ldpc_base_graph, pdu.qm, 0, pdu.nof_layers, d["n_ref"], tb_size, G // pdu.qm,
                              args.iterations, 1 if early_stop else 0, 1)
    soft_bytes, state_bytes, ncb = ctx.pusch_decoder_sizes(cfg, slots)
    d_soft = torch.zeros((slots, soft_bytes), dtype=torch.int8, device="cuda")
    d_state = torch.zeros((state_bytes,), dtype=torch.uint8, device="cuda")
    d_out = torch.zeros((slots, tb_stride), dtype=torch.uint8, device="cuda")
    d_res = torch.zeros((slots, 4), dtype=torch.int32, device="cuda")
    oplan = lib.OfdmPlan(ctx, ocfg, ports)
    # The OFDM demodulator's input is noise of the right shape: without the channel estimator and equaliser its output
    # does not feed the soft demodulator (module docstring); its time is that of any IQ of this size.
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
        ctx.demodulate_soft(qm, slots, nsym, d_sym, d_nv, d_llr_scr, s.cuda_stream)
        if timed:
            ev[2].record(s)
        ctx.llr_descramble(d_c_init, slots, G, d_llr_scr, G, d_llr, G, s.cuda_stream)
        if timed:
            ev[3].record(s)
        ctx.pusch_decode_batch(cfg, slots, d_llr, G, d_soft, d_state, d_out, tb_stride, d_res, s.cuda_stream)
        if timed:
            ev[4].record(s)

    for _ in range(max(1, args.warmup)):
        step(False)
    torch.cuda.synchronize()
    res = d_res.cpu().numpy()
    ok = res[:, 0] != 0
    if require_all:
        assert ok.all(), "every transport block must decode at this SNR (%d of %d did)" % (int(ok.sum()), slots)
    okt = torch.from_numpy(ok).cuda()
    assert torch.equal(d_out[okt][:, :tb_size], d_tb[okt][:, :tb_size]), "a decoded transport block differs from what was sent"
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
    demod_bytes = slots * nsym * (12 + qm)   # symbol + noise variance in, Qm soft bits out
    n_cb = slots * C
    alg = n_cb * d["full_length"] + slots * tb_size  # decoder: soft buffers in, transport blocks out
    # The decoder is bound by vector instruction issue: vector instructions per codeblock from the PMC
    # profile of this script (profiles/rx_leg_profile.sh -> profiles/traffic.json, used only when measured on these kernel
    # sources) x the average issue cost of the kernel's own instruction mix (profiles/valu_issue_model.py).
    hbm = {"achieved": alg / kernel_ms["pusch_decode_batch"] * 1e-6, "peak": 8000.0, "unit": "GB/s",
           "frac": alg / kernel_ms["pusch_decode_batch"] * 1e-6 / 8000.0}
    kernel_name = "ldpc_decode_msg_bg1_kernel" if leg == "bg1" else "ldpc_decode_msg_bg2_slot_kernel"
    roofline = dict(hbm, bound="hbm", kernel=kernel_name, traffic=None)
    mean_it = float(res[:, 2].sum()) / n_cb
    iterations_run = float(args.iterations) if not early_stop else mean_it
    try:
        import roofline_util
        tj = roofline_util.profile_table()
        if tj.get("stale"):
            roofline["note"] = tj["stale"]
        c4 = tj.get("rx_valu_insts_per_codeblock_at_4_iterations", {}).get(leg)
        c8 = tj.get("rx_valu_insts_per_codeblock_at_8_iterations", {}).get(leg)
        cost = tj.get("valu_issue_model", {}).get(kernel_name, {}).get("avg_issue_cycles_per_instruction")
        if c8 and cost:
            # instructions at this run's iteration count: on the line through the two profiled points (below four iterations:
            # pro rata of the first point -- the first iteration is the cheaper one, so that is an upper estimate); with one
            # point only, pro rata of it
            it = iterations_run
            if c4:
                per_cb = c4 + (c8 - c4) * (it - 4.0) / 4.0 if it >= 4.0 else c4 * it / 4.0
            else:
                per_cb = c8 * it / 8.0
            roofline = roofline_util.valu_roof(n_cb * per_cb, kernel_ms["pusch_decode_batch"], cost, tj.get("rx_clock_ghz", {}).get(leg))
            roofline.update({"kernel": kernel_name, "hbm": hbm, "valu_insts_per_codeblock": per_cb,
                             "traffic": (int(tj["rx_hbm_bytes_per_codeblock"][leg] * n_cb) if leg in tj.get("rx_hbm_bytes_per_codeblock", {}) else None),
                             "note": "vector instructions per codeblock from the PMC profile named in profiles/traffic.json (rx_source), "
                                     "launch time of this run (the whole nrphy_pusch_decode_batch call: dematcher, decoder, assembly)"})
    except Exception:
        pass
    cpu = None
    if getattr(args, "cpu_baseline", False):
        try:
            cpu = cpu_decode_baseline(pdu, d, args.iterations, early_stop, d_llr[0].cpu().numpy())
        except Exception as e:  # the checker is optional at bench time
            cpu = {"unavailable": str(e)}
    extra = {}
    if cpu is not None:
        extra["cpu_baseline"] = cpu
        extra["decode_chain_only_slots_per_sec"] = slots / kernel_ms["pusch_decode_batch"] * 1e3  # what cpu_baseline is the counterpart of
    return {
        **extra,
        "metric": "pusch_rx_slots_per_second", "value": slots / ms * 1e3, "unit": "slots/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int8 LLR / f32 IQ", "data": "synthetic",
        "config": {"workload": "BASELINE config 5: OFDM demod (4096, 273 PRB, %d ports) + soft demodulator (%s) + descrambling + "
                               "UL-SCH decoder (rate dematch, LDPC BG%d %s, concatenation, TB CRC) on 100 MHz 4-layer transport "
                               "blocks (%d bit, %d CB, Zc %d)" % (
                                   ports, "256-QAM" if qm == 8 else "QPSK", pdu.ldpc_base_graph,
                                   ("max %d iterations with CRC early stop" if early_stop else "%d iterations, no early stop") %
                                   args.iterations, 8 * tb_size, C, d["lifting_size"]),
                   "slots_per_step": slots, "codeblocks_per_step": n_cb, "snr_dB": snr_db, "early_stop": early_stop,
                   "base_graph": int(pdu.ldpc_base_graph)},
        "kernel_ms": kernel_ms, "mean_iterations": mean_it, "iterations_run_per_codeblock": iterations_run,
        "tb_crc_ok": int(ok.sum()), "codeblocks_per_sec": n_cb / ms * 1e3,
        "info_gbps": slots * tb_size * 8 / ms * 1e-6,
        "demodulate_soft_roofline": {"bound": "hbm", "kernel": "demodulate_soft_kernel", "unit": "GB/s", "peak": 8000.0,
                                     "achieved": demod_bytes / kernel_ms["demodulate_soft"] * 1e-6,
                                     "frac": demod_bytes / kernel_ms["demodulate_soft"] * 1e-6 / 8000.0,
                                     "algorithmic_bytes_per_launch": demod_bytes},
        "roofline": roofline,
        "verified": "%d of %d transport blocks decoded (CRC24A), each equal to what was sent" % (int(ok.sum()), slots),
    }


if __name__ == "__main__":
    main()
