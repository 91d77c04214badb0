#!/usr/bin/env python3
"""Headline benchmark: PDSCH slots/s (+ IQ Gsamples/s) at 100 MHz / 4 layers / 256-QAM on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...          # N > 1 without a launcher: starts the N ranks itself (a child
                                          # `python -m torch.distributed.run`, before this process touches a GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 2 --dry-run    # the launch / sharding / aggregation path without device work (gloo; CPU tests)

A "step" is one pass of the whole hot path (TB CRC -> segmentation -> LDPC -> rate matching -> scrambling -> QAM ->
layer mapping/precoding/RE mapping -> DM-RS -> OFDM) over one batch of --slots BASELINE-config-3 slots per GPU with
distinct transport blocks already resident in HBM.  Slots are independent, so ranks own disjoint slot batches
(sharding.shard_slots; config 4: sharding.cell_affine_rank), weak scaling, no data-path collective; RCCL is used for the
barrier and the max-over-ranks time only.  Rank 0 prints ONE JSON line.  Timing: W warm-up steps, then exactly K steps between
barriers + synchronisation; when W is smaller than --settle (default 30 steps, about 25 ms of load: what the engine clocks need
to settle) the difference runs as untimed settling steps in front of the warm-up and is reported as "settle_steps".

At N=1 the same line also carries, under "secondary", measured-and-verified entries for the other BASELINE configs
(2: 20 MHz 2-layer 64-QAM 1000-slot batch; 4: four cells x four UEs mixed MCS, 1024 cell-slots; 5: receive chain with
8 LDPC iterations) and for config 3 with the wire-format (complex int16) OFDM output, each with its own roofline dict,
and "cpu_baseline": the reference's CPU path timed on this host's cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota():
    """CPUs the container may use at a time (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return float(quota) / float(period)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except Exception:
        pass
    return None


def cpu_baseline(pdu, tb, nof_ports, nof_subc, ofdm, budget_s=30.0):
    """Times the CPU path on the host cores for a bounded sample of the same workload (rank 0, N=1 only), with the reference
    benchmark's scheme (pdsch_processor_benchmark.cpp:684-737: T worker threads, each with its own processor instance):
    PDSCH + OFDM and PDSCH alone, at T = 1 and at several T up to every CPU this process may run on -- the affinity mask of
    a GPU box covers the whole host (256 logical CPUs), of which other tenants use a share that varies, so the line reports
    every T measured and takes the best as `value`.  Uses the compiled reference (oracle/_ref, kind "reference": it
    travels to the GPU box as a built library like the product's own .so) when present, the C oracle (kind "port")
    otherwise."""
    import ctypes as C
    import backends
    affinity = len(os.sched_getaffinity(0))
    quota = cpu_quota()
    cap = int(os.environ.get("NRPHY_CPU_THREADS_MAX", "0"))  # 0 = no cap
    limit = min(affinity, cap) if cap > 0 else affinity
    r = backends.ref()
    if r is not None:
        kind = "reference"

        def run(threads, reps, with_ofdm):
            return r.lib.ref_bench_pdsch(C.byref(pdu), tb.ctypes.data, nof_ports, nof_subc,
                                         C.byref(ofdm) if with_ofdm else None, threads, reps, 1)
    else:
        kind = "port"
        o = backends.oracle()

        def run(threads, reps, with_ofdm):
            return o.lib.oracle_bench(C.byref(pdu), tb.ctypes.data, nof_ports, nof_subc,
                                      C.byref(ofdm) if with_ofdm else None, threads, reps)

    def measure(threads, share_s, with_ofdm, batches=4):
        run(threads, 1, with_ofdm)               # warms caches / page-faults the buffers
        t1 = run(threads, 2, with_ofdm) / 2.0    # calibration: seconds per slot per thread
        reps = int(max(1, min(5000, share_s / batches / max(t1, 1e-4))))
        rates, total = [], 0.0
        for _ in range(batches):
            dt = run(threads, reps, with_ofdm)
            rates.append(threads * reps / dt)
            total += dt
        rates = np.array(rates)
        return {"threads": threads, "slots_per_sec": round(float(np.median(rates)), 2),
                "p5": round(float(np.percentile(rates, 5)), 2), "p95": round(float(np.percentile(rates, 95)), 2),
                "batches": batches, "slots_per_thread_per_batch": reps, "seconds": round(total, 1)}

    counts = sorted({t for t in (16, 32, 64, 128, int(quota) if quota else 0, limit) if 1 < t <= limit})
    share = budget_s * 0.6 / max(1, 2 * len(counts))
    sweep = [(measure(t, share, True), measure(t, share, False)) for t in counts]
    one = measure(1, budget_s * 0.22, True)
    one_pdsch = measure(1, budget_s * 0.18, False)
    full, full_pdsch = max(sweep, key=lambda e: e[0]["slots_per_sec"]) if sweep else (one, one_pdsch)
    seconds = sum(a["seconds"] + b["seconds"] for a, b in sweep) + one["seconds"] + one_pdsch["seconds"]
    return {
        "value": full["slots_per_sec"],
        "unit": "slots/s",
        "cores": full["threads"],
        "kind": kind,
        "cpu_model": cpu_model(),
        "host_logical_cpus": os.cpu_count(),
        "affinity_cpus": affinity,
        "cgroup_cpu_quota": quota,
        "thread_cap": cap if cap > 0 else None,
        "all_threads": full,
        "all_threads_pdsch_only": full_pdsch,
        "thread_sweep": [{"threads": a["threads"], "pdsch_ofdm_slots_per_sec": a["slots_per_sec"],
                          "pdsch_only_slots_per_sec": b["slots_per_sec"]} for a, b in sweep],
        "one_thread": one,
        "one_thread_pdsch_only": one_pdsch,
        "sample": "config-3 slots (PDSCH %s; OFDM generic radix-2 DFT), the reference benchmark's threads x batch scheme at "
                  "T = %s and 1 threads (affinity mask: %d of the host's %s logical CPUs, cgroup quota %s), PDSCH + OFDM and PDSCH "
                  "alone, %d batches each (%.0f s in all); value = the best T's median PDSCH + OFDM rate" % (
                      "AVX2 LDPC/precoder, the reference's own objects" if kind == "reference" else "scalar C oracle",
                      "/".join(str(t) for t in counts), affinity, os.cpu_count(), quota if quota else "none", full["batches"],
                      seconds),
    }


sys.path.insert(0, os.path.join(ROOT, "profiles"))
from roofline_util import profile_table, valu_roof  # noqa: E402  (vector-issue cost model, profiles/traffic.json with its source check)


def roof(name, ms, nbytes, slots=None, config=None, wire=False):
    """Roofline dict of one kernel.  HBM figures always; for a kernel that profiles/traffic.json lists as bound by vector
    instruction issue ("valu_bound"), `bound` says so and achieved / peak / frac are its vector-issue figures (valu_roof), with
    the HBM figures alongside under "hbm"."""
    gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    out = {"kernel": name, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(gbs / HBM_PEAK_GBS, 4), "ms_per_launch": round(ms, 4), "algorithmic_bytes_per_launch": int(nbytes),
           "traffic": None}
    tj = profile_table()
    if tj.get("stale"):
        out["note"] = tj["stale"]
    key = name if not wire else name   # (the wire-format kernel has its own key: "ofdm_kernel<4096, ci16>")
    if config == 3 and tj.get("slots") == slots:
        if not wire or "ci16" in name:
            out["traffic"] = tj.get("hbm_bytes_per_launch", {}).get(key)
        valu = tj.get("valu_insts_per_launch", {}).get(key)
        cost = tj.get("valu_issue_model", {}).get(key, {}).get("avg_issue_cycles_per_instruction")
        clock = tj.get("clock_ghz", {}).get(key)
        if valu and cost and ms > 0:
            v = valu_roof(valu, ms, cost, clock)
            if name in tj.get("valu_bound", []):
                hbm = {k: out[k] for k in ("achieved", "peak", "unit", "frac")}
                out.update(v)
                out["hbm"] = hbm
            else:
                out["valu"] = {k: v[k] for k in v if k != "bound"}
                if name == "prologue_kernel":
                    # neither roof binds this launch: both figures, and what the workgroup timeline shows
                    out["note"] = ("bound by its two dependent chains, whose waves take turns at the SIMDs and the LDS: a TB-CRC workgroup's "
                                   "regions (byte-table look-ups) and a sequence wave's blocks of 31 rows (profiles/r03_prologue_trace.txt)")
    return out


EVENT_STRIDE = 4  # every fourth step of the timed region carries the HIP events of the kernel durations


def run_downlink(env, config, slots, steps, warmup, wire=False):
    """Builds the workload of BASELINE config 2, 3 or 4 for this rank, times `steps` steps and returns the measurement (on
    rank 0: the JSON fields; elsewhere None)."""
    torch, lib, abi, cases, sharding = env["torch"], env["lib"], env["abi"], env["cases"], env["sharding"]
    ctx, dist, rank, world, device = env["ctx"], env["dist"], env["rank"], env["world"], env["device"]
    if config == 4:
        # Four cells x four UEs (68 PRB each, QPSK / 16 / 64 / 256-QAM): one grid per cell-slot, four PDUs per grid.  The
        # job's stream of cell-slots (cell = i % 4, slot = i // 4) is placed by cell affinity; every rank gets `slots` of them.
        # (1, 2, 4, 8 ranks: `slots` each; other world sizes leave the shares uneven -- every rank then runs its real share and
        # the totals add up what was processed.)
        mine = [i for i in range(slots * world) if sharding.cell_affine_rank(i % 4, i // 4, world, 4) == rank]
        slots = len(mine)
        pdus, grid_of = [], []
        for g, i in enumerate(mine):
            cell, nof_ports, nof_subc = cases.mixed_cell(i % 4, slot_index=(i // 4) % 20)
            pdus += cell
            grid_of += [g] * len(cell)
        ofdm = cases.baseline_config(3)[3]
        workload = ("BASELINE config 4: 100 MHz cell-slots, 4 cells x 4 UEs (68 PRB each; QPSK 120, 16-QAM 658, "
                    "64-QAM 873, 256-QAM 948; 4 layers) + 4-port OFDM")
    else:
        nof_ports, nof_subc, ofdm = cases.baseline_config(config)[1:]
        first, count = sharding.shard_slots(slots * world, rank, world)
        assert count == slots
        period = 20 if config == 3 else 10
        pdus = [cases.baseline_config(config, slot_index=(first + i) % period)[0] for i in range(slots)]
        grid_of = list(range(slots))
        d0 = lib.derive(pdus[0])
        workload = {
            3: "BASELINE config 3: 100 MHz (FFT 4096, 30 kHz SCS, 273 PRB grid) 4-layer 256-QAM R=948/1024 full-TBS "
               "PDSCH (TBS %d bit, %d CB, BG1 Zc %d) + 4-port OFDM",
            2: "BASELINE config 2: 20 MHz (FFT 2048, 15 kHz SCS, 106 PRB) 2-layer 64-QAM R=873/1024 PDSCH "
               "(TBS %d bit, %d CB, BG1 Zc %d) + 2-port OFDM",
        }[config] % (8 * pdus[0].tb_size_bytes, d0["nof_codeblocks"], d0["lifting_size"])
    if wire:
        workload += ", OFDM output as complex int16 (amplitude controller + cf32->ci16 fused into the modulator's store)"
    if slots == 0:  # a rank without a share (config 4 on 5 ... 7 ranks): takes part in the barriers and the totals only
        if dist is not None:
            dist.barrier()
            dist.barrier()
        sharding.aggregate(dist, device, 0, 0, 0.0)
        sharding.gather(dist, device, [0.0, 0.0, 0.0, 0.0])
        return None
    tb_strides = [(q.tb_size_bytes + 255) & ~255 for q in pdus]
    tb_offsets = [0]
    for st in tb_strides[:-1]:
        tb_offsets.append(tb_offsets[-1] + st)
    tb_total = tb_offsets[-1] + tb_strides[-1]
    tb_bytes_per_step = sum(q.tb_size_bytes for q in pdus)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234 + rank)
    # Four sets of transport blocks taken in turn (0.45 GB at the default size): a step never finds its input in the
    # memory-side cache because an earlier step read the same bytes.
    tb_sets = [torch.randint(0, 256, (tb_total,), dtype=torch.uint8, device="cuda", generator=gen) for _ in range(4)]
    plan = lib.PdschPlan(ctx, pdus, tb_offsets, grid_of, slots, nof_ports, nof_subc)
    oplan = lib.OfdmPlan(ctx, ofdm, nof_ports)
    d_grid = torch.zeros((slots, nof_ports, 14, nof_subc), dtype=torch.int32, device="cuda")
    if wire:
        # The amplitude controller as the reference's radio unit configures it: the modulator's scale is 1 (pdxch_processor_factories.cpp:59),
        # and the input gain takes out the DFT's power gain plus a back-off for the signal's peaks -- input_gain_dB =
        # -10 log10(subcarriers) - tx_gain_backoff (12 dB), ceiling -0.1 dBFS (apps/units/flexible_du/split_8/ru_sdr_config_translator.cpp:
        # 94-100, ru_sdr_config.h:75-81) -- with clipping switched ON (the reference's default is off), the costlier setting.
        # (Until late round 4 this leg ran with a gain of -14 dB: at scale 1 that puts the signal 14 dB ABOVE full scale -- every sample
        # clipped, the conversion's exact path for all of them: a measurement of the wrong thing.  profiles/r04_ofdm_sinks.txt.)
        wire_cfg = abi.IqWireCfg(abi.AmplitudeCfg(0, 1, -(10.0 * float(np.log10(nof_subc)) + 12.0), 1.0, -0.1), 32767.0)
        d_iq = torch.zeros((slots, nof_ports, oplan.slot_stride, 2), dtype=torch.int16, device="cuda")
        d_stats = torch.zeros((slots * nof_ports, 4), dtype=torch.int32, device="cuda")
    else:
        d_iq = torch.zeros((slots, nof_ports, oplan.slot_stride, 2), dtype=torch.float32, device="cuda")
    sps = 2 if config != 2 else 1  # slots per subframe
    d_slot = torch.tensor([i % sps for i in range(slots)], dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    # Everything PyTorch queued (allocations, fills, copies) has completed; from here on only the library's own stream runs, and
    # both sides are synchronised explicitly around the timed region: no per-call wait for PyTorch's stream inside it.
    lib.ORDER_AFTER_TORCH = False

    step_no = [0]

    def step():
        plan.run(tb_sets[step_no[0] % len(tb_sets)], d_grid, zero_grids=True)
        if wire:
            oplan.run_ci16(slots, d_grid, wire_cfg, d_iq, d_slot_index=d_slot, d_stats=d_stats)
        else:
            oplan.run(slots, d_grid, d_iq, d_slot_index=d_slot)
        step_no[0] += 1

    def barrier():
        if dist is not None:
            dist.barrier()

    # The engine clocks take some tens of milliseconds of continuous load to settle (with 5 warm-up steps
    # -- 4 ms -- the vector-bound codeblock launch reads 0.355 ms where it runs at 0.306 ms): when the caller asks for fewer
    # warm-up steps than that takes, untimed settling steps run in front of them.  They are reported ("settle_steps"), the
    # warm-up count stays what was asked for, and --settle 0 switches them off.
    settle_steps = max(0, env.get("settle", 0) - warmup)
    for _ in range(settle_steps + warmup):
        step()
    ctx.synchronize()
    # Kernel durations by HIP events on every fourth step of the timed region: an event between two launches costs the stream
    # 4-5 us, six per step 0.02 ms of a 0.88 ms step (profiles/r03_codeblock_experiments.txt, "events").
    # NRPHY_BENCH_NO_KERNEL_EVENTS=1: none (the probe that measured it).
    if os.environ.get("NRPHY_BENCH_NO_KERNEL_EVENTS") != "1":
        plan.enable_timing(steps, stride=EVENT_STRIDE)
        oplan.enable_timing(steps, stride=EVENT_STRIDE)
    barrier()
    torch.cuda.synchronize()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    local_dt = time.perf_counter() - t0   # this rank's own timed region: before the barrier, so that the ranks can differ
    barrier()
    dt = time.perf_counter() - t0
    lib.ORDER_AFTER_TORCH = True
    # Whole-job totals: slots and IQ samples summed over ranks, time = max over ranks (RCCL all-reduce of 3 numbers).
    samples_per_slot = nof_ports * oplan.slot_stride
    total_slots, total_samples, dt = sharding.aggregate(dist, device, slots * steps, slots * steps * samples_per_slot, dt)

    (ms_crc, ms_cb, ms_dmrs, ms_run), _ = plan.kernel_times()
    ms_ofdm, _ = oplan.kernel_time()
    # Per rank: slots per step, seconds of the timed region, codeblock and OFDM launch times (rank 0 reports them).
    per_rank = sharding.gather(dist, device, [float(slots), local_dt, ms_cb, ms_ofdm])
    out = None
    if rank == 0:
        # Algorithmic bytes per slot: TB read + grid written once (incl. zeros) + grid read by the OFDM
        # modulator + IQ write (8 bytes per sample as complex float, 4 as complex int16).
        grid_bytes = nof_ports * 14 * nof_subc * 4
        alg_pdsch = tb_bytes_per_step / slots + grid_bytes
        alg_ofdm = grid_bytes + samples_per_slot * (4 if wire else 8)
        ofdm_name = "ofdm_kernel<%d%s>" % (ofdm.dft_size, ", ci16" if wire else "")
        kernels = {
            # name: (avg ms per launch, algorithmic bytes per launch)
            ofdm_name: (ms_ofdm, slots * alg_ofdm),
            # The codeblock launch also carries the DM-RS and zero-fill waves: it writes every grid word exactly once.
            "codeblock_kernel": (ms_cb, tb_bytes_per_step + slots * grid_bytes),
        }
        if ms_crc > 0:
            kernels["prologue_kernel"] = (ms_crc, tb_bytes_per_step)
        dom = max(kernels, key=lambda k: kernels[k][0])
        alg_slot = alg_pdsch + alg_ofdm
        rank_rates = [r[0] * steps / r[1] for r in per_rank if r[1] > 0]
        out = {
            "metric": "pdsch_slots_per_sec",
            "value": round(total_slots / dt, 1),
            "unit": "slots/s",
            "steps": steps,
            # every untimed step before the timed region: the W asked for plus the settling steps in front of them
            "warmup": settle_steps + warmup,
            "warmup_requested": warmup,
            "settle_steps": settle_steps,
            "ms_per_step": round(1e3 * dt / steps, 4),
            "config": {"workload": workload, "slots_per_gpu_per_step": slots,
                       "parallelism": "slot-sharded x%d, no data-path collective" % world,
                       "clock": ("settled: %d untimed steps (%d requested + %d settling) precede the timed region -- the engine clocks need "
                                 "about 30 ms of load; --settle 0 times straight after the requested warm-up" % (
                                     settle_steps + warmup, warmup, settle_steps)) if settle_steps else
                                "as requested: %d untimed warm-up steps" % warmup},
            "iq_gsamples_per_sec": round(total_samples / dt / 1e9, 3),
            "whole_path_hbm_frac": round(total_slots * (alg_pdsch + alg_ofdm) / dt / 1e9 / (HBM_PEAK_GBS * world), 4),
            "kernel_ms": {"prologue_tbcrc_scrambling_seq": round(ms_crc, 4), "codeblock_dmrs_zerofill": round(ms_cb, 4),
                          "separate_dmrs": round(ms_dmrs, 4), "pdsch_run": round(ms_run, 4), "ofdm": round(ms_ofdm, 4)},
            "roofline": roof(dom, *kernels[dom], slots=slots, config=config, wire=wire),
            "roofline_other": [roof(k, *kernels[k], slots=slots, config=config, wire=wire) for k in kernels if k != dom],
            "per_rank": {
                "slots_per_step": [int(r[0]) for r in per_rank],
                "slots_per_sec": {"min": round(min(rank_rates), 1), "max": round(max(rank_rates), 1)},
                # algorithmic HBM rate of each GPU over its own timed region, and its fraction of one GPU's 8 TB/s
                "hbm_gbs": [round(r[0] * steps * alg_slot / r[1] / 1e9, 1) if r[1] > 0 else 0.0 for r in per_rank],
                "hbm_frac": [round(r[0] * steps * alg_slot / r[1] / 1e9 / HBM_PEAK_GBS, 4) if r[1] > 0 else 0.0
                             for r in per_rank],
                "codeblock_ms": [round(r[2], 4) for r in per_rank], "ofdm_ms": [round(r[3], 4) for r in per_rank],
            },
        }
        if config == 4:
            out["config"]["sharding"] = "sharding.cell_affine_rank (cell c keeps to its ranks)"
        # Sanity: grid 0 of the last step against the CPU oracle (checker only, outside the timed region).
        last = tb_sets[(step_no[0] - 1) % len(tb_sets)]
        try:
            import backends
            o = backends.oracle()
            want = None
            for k, q in enumerate(pdus):  # the PDUs of grid 0 map disjoint RE: their grids OR together
                if grid_of[k] != 0:
                    break
                tbk = last[tb_offsets[k]: tb_offsets[k] + q.tb_size_bytes].cpu().numpy()
                part = o.pdsch_process(q, tbk, nof_ports, nof_subc)
                want = part if want is None else np.bitwise_or(want, part)
            got = d_grid[0].cpu().numpy().view(np.uint16).reshape(want.shape)
            ok = bool(np.array_equal(got, want))
            if ok and wire:
                # wire format: slot 0, port 0 against the oracle's modulator -> amplitude controller -> int16 chain (one LSB:
                # the two FFTs differ by 1e-7 relative)
                ref_iq = o.ofdm_slot(ofdm, want, 0)[0]
                y, _ = o.amplitude_control(wire_cfg.amplitude, ref_iq)
                w16 = o.iq_convert_ci16(y, wire_cfg.ci16_scale).reshape(-1, 2).astype(np.int32)
                g16 = d_iq[0, 0, : w16.shape[0]].cpu().numpy().astype(np.int32)
                ok = bool(np.abs(g16 - w16).max() <= 1)
            elif ok:
                # the dominant kernel's output: slot 0, every port, against the oracle's modulator (north-star tolerance 1e-5)
                ref_iq = o.ofdm_slot(ofdm, want, 0)
                got_iq = d_iq[0].cpu().numpy().view(np.complex64).reshape(nof_ports, -1)[:, : ref_iq.shape[1]]
                err = float(np.abs(got_iq - ref_iq).max() / np.abs(ref_iq).max())
                out["iq_rel_err_vs_oracle"] = float("%.3g" % err)
                ok = bool(err < 1e-5)
            out["verified_vs_oracle"] = ok
            out["verified"] = "grid 0 bit-exact (bf16) and its IQ (%s) against the CPU oracle" % (
                "int16, one LSB" if wire else "f32, 1e-5 relative, all ports")
        except Exception as e:  # the oracle is optional at bench time
            out["verified_vs_oracle"] = "unavailable: %s" % e
        out["_first_pdu"] = (pdus[0], last[: pdus[0].tb_size_bytes].cpu().numpy().copy(), nof_ports, nof_subc, ofdm)
    plan.close()
    oplan.close()
    del tb_sets, d_grid, d_iq
    torch.cuda.empty_cache()
    return out


def shim_entry(env):
    """Through-the-shim figures ("host spans, boundary A/C semantics, reported separately"): config-3-sized
    live traffic -- every PDU / slot differs -- through the host-span seams, PCIe included.  Bounded: about two seconds."""
    import ctypes as C
    import shim_latency
    ctx, lib = env["ctx"], env["lib"]
    h = ctx.lib
    rng = np.random.default_rng(7)
    items = shim_latency.live_pdu_pool(rng)
    max_tb = max(p.tb_size_bytes for p, _ in items)
    # seam A alone, asynchronous, 4 PDUs in flight: transport block down, whole grid up per PDU
    done_fn = C.cast(h.nrphy_pdsch_async_count_done, C.c_void_p)
    qh = C.c_void_p()
    assert h.nrphy_pdsch_async_create(ctx.handle, 4, 4, 273 * 12, max_tb, C.byref(qh)) == 0
    count = C.c_uint64(0)
    refs = [(C.byref(p), tb.ctypes.data) for p, tb in items]

    def pump(n):
        for k in range(n):
            pr, tbp = refs[k % len(refs)]
            while True:
                rc = h.nrphy_pdsch_async_submit(qh, pr, tbp, done_fn, C.byref(count))
                if rc == 0:
                    break
                assert rc == 4, rc
                h.nrphy_pdsch_async_wait_slot(qh)
        h.nrphy_pdsch_async_wait(qh)

    pump(64)
    count.value = 0
    total = 800
    t0 = time.perf_counter()
    pump(total)
    dt = time.perf_counter() - t0
    assert count.value == total
    h.nrphy_pdsch_async_destroy(qh)
    grid_bytes = 4 * 14 * 273 * 12 * 4
    legs = shim_latency.dl_slot_pipeline(ctx, depths=(1, 4), total=400, verbose=False)
    return {
        "workload": "config-3-sized live traffic (64 PDU shapes; slot index, RNTI, identities, MCS, allocation all changing), host spans",
        "seam_a_async_4_in_flight": {"pdus_per_sec": round(total / dt, 1), "ms_per_pdu": round(1e3 * dt / total, 4),
                                     "pcie_bytes_down": int(np.mean([p.tb_size_bytes for p, _ in items])) + 4096,
                                     "pcie_bytes_up": grid_bytes,
                                     "entry": "nrphy_pdsch_async_submit (pdsch_processor::process; grid merged on the host)"},
        "seams_a_c_slot_pipeline": legs,
        "entry": "nrphy_dl_slot_open / _pdsch / _modulate / _wait / _close (grid stays in HBM; pdxch_processor hand-over semantics)",
    }


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args):
    """--gpus N > 1 without a launcher around us: start the N ranks as a CHILD `python -m torch.distributed.run` (what the
    reference's harness does with its T worker threads, pdsch_processor_benchmark.cpp:684-737) before this process has
    made any GPU call -- a process that touched the GPU must never be replaced or re-executed -- and leave with its exit
    code.  Rank 0 of the child prints the one JSON line on the stdout we share."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """The N-rank path without device work: process group (gloo), slot sharding / cell-affine placement, barriers, totals
    and the per-rank gather, one line from rank 0.  What the CPU tests run; never a measurement (value 0, "dry_run": true)."""
    import torch
    import backends
    sharding = backends.load_package().sharding
    dist = None
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    device = torch.device("cpu")
    if args.config == 4:
        mine = [i for i in range(args.slots * world) if sharding.cell_affine_rank(i % 4, i // 4, world, 4) == rank]
        slots = len(mine)
    else:
        slots = sharding.shard_slots(args.slots * world, rank, world)[1]
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    total_slots, _, dt = sharding.aggregate(dist, device, slots * args.steps, 0, dt)
    per_rank = sharding.gather(dist, device, [float(slots), dt, 0.0, 0.0])
    if rank == 0:
        print(json.dumps({
            "metric": "pdsch_slots_per_sec", "value": 0.0, "unit": "slots/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 0.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "none", "data": "none (dry run: no device work, not a measurement)", "dry_run": True,
            "config": {"workload": "dry run of BASELINE config %d" % args.config, "slots_per_gpu_per_step": args.slots,
                       "parallelism": "slot-sharded x%d, no data-path collective" % world},
            "total_slots": total_slots, "per_rank": {"slots_per_step": [int(r[0]) for r in per_rank]},
            "collective_backend": "gloo (torch.distributed)" if dist is not None else "none (single process)",
        }), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # Defaults: 30 untimed steps (30 ms) bring the device to its steady clocks -- with 3 the compute-bound codeblock launch reads
    # 11 % slow and the whole step 4-5 % (profiles/r02_codeblock_experiments.txt, "warm-up") -- then 50 timed steps.
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--settle", type=int, default=30,
                    help="untimed steps in all before the timed region: when --warmup is smaller, settling steps make up the difference "
                         "(the engine clocks need ~30 ms of load; 0 = warm-up steps only)")
    ap.add_argument("--slots", type=int, default=1024, help="slots (config 4: cell-slots) per GPU per step")
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5],
                    help="BASELINE config measured as the line's value: 3 = the headline workload (default)")
    ap.add_argument("--wire", action="store_true", help="OFDM output as complex int16 (amplitude controller + conversion fused)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary entries (configs 2, 4, 5, wire format)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch, sharding and aggregation only (gloo, no device work): the N-rank path on a machine without GPUs")
    ap.add_argument("--rehearse-one-device", action="store_true",
                    help="rehearsal of the N-rank path on a ONE-GPU box: every rank works on cuda:0 and the process group is gloo "
                         "(RCCL refuses two ranks on one device); the line says so and is not a scaling measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if "RANK" not in os.environ:
        if args.gpus > 1:
            sys.exit(self_launch(args))  # nothing in this process has touched a GPU
    elif int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ.get("WORLD_SIZE")),
              file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    if args.dry_run:
        dry_run(args, rank, world)
        return
    if args.config == 5:
        # the receive-side chain has its own script (profiles/rx_chain_bench.py); same JSON schema, one GPU
        if world != 1:
            print("bench.py: --config 5 is a one-GPU measurement", file=sys.stderr)
            sys.exit(2)
        import rx_chain_bench
        print(json.dumps(rx_chain_bench.run(argparse.Namespace(
            slots=args.slots, iterations=8, steps=args.steps, warmup=args.warmup, snr_db=32.0))),
            flush=True)
        return

    import torch
    import backends
    import cases
    lib = backends.pkg.lib

    if args.rehearse_one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ:
        # under torch.distributed.run, also at N=1: RCCL initialisation, barrier and the totals' all-reduce all run
        import torch.distributed as dist
        if args.rehearse_one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    env = {"torch": torch, "lib": lib, "abi": backends.abi, "cases": cases, "sharding": backends.pkg.sharding,
           "ctx": lib.Context(local_rank), "dist": dist, "rank": rank, "world": world,
           "device": torch.device("cpu") if args.rehearse_one_device else torch.device("cuda", local_rank),
           "settle": max(0, args.settle)}

    head = run_downlink(env, args.config, args.slots, args.steps, args.warmup, wire=args.wire)
    if rank == 0:
        first = head.pop("_first_pdu")
        out = {
            "metric": head["metric"], "value": head["value"], "unit": head["unit"], "n_gpus": world, "steps": args.steps,
            "warmup": head["warmup"], "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 bit-packed GF(2) + bf16 grid + %s IQ" % ("ci16" if args.wire else "f32"),
            "data": "synthetic",
        }
        out.update({k: v for k, v in head.items() if k not in out})
        out["collective_backend"] = ("none (single process)" if dist is None else
                                     "gloo (rehearsal: %d ranks on ONE device, not a scaling measurement)" % world
                                     if args.rehearse_one_device else "rccl (torch.distributed nccl)")
    sec = {}
    s_steps, s_warm = max(3, args.steps), max(3, args.warmup)  # the same count as the headline: short runs read 5-10 % slow
    if world == 1 and not args.no_secondary and args.config == 3 and not args.wire:
        # Secondary measurements, same process and GPU; every entry is verified against the oracle.
        for name, (cfg, slots, wire) in {"config3_wire_ci16": (3, args.slots, True), "config2": (2, 1000, False),
                                         "config4": (4, 1024, False)}.items():
            e = run_downlink(env, cfg, slots, s_steps, s_warm, wire=wire)
            e.pop("_first_pdu")
            sec[name] = e
        import rx_chain_bench
        sec["config5"] = rx_chain_bench.run_all(s_steps, s_warm)
        sec["shim"] = shim_entry(env)
    elif world > 1 and not args.no_secondary and args.config == 3 and not args.wire:
        # N ranks: BASELINE config 4 is the one quoted as a stream sharded over the GPUs -- placed by cell affinity, every
        # rank its share, whole-job totals like the headline.
        e = run_downlink(env, 4, 1024, s_steps, s_warm)
        if rank == 0:
            e.pop("_first_pdu")
            sec["config4"] = e
    if rank == 0:
        if sec:
            out["secondary"] = sec
        if world == 1 and not args.no_cpu_baseline and args.config == 3:
            out["cpu_baseline"] = cpu_baseline(*first)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
