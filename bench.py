#!/usr/bin/env python3
"""Headline benchmark: PDSCH slots/s (+ IQ Gsamples/s) at 100 MHz / 4 layers / 256-QAM on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path (TB CRC -> segmentation -> LDPC -> rate matching -> scrambling -> QAM ->
layer mapping/precoding/RE mapping -> DM-RS -> OFDM) over one batch of --slots BASELINE-config-3 slots per GPU with
distinct transport blocks already resident in HBM.  Slots are independent, so ranks just own disjoint slot batches
(weak scaling, no data-path collective); RCCL is used for the barrier and the max-over-ranks time only.
Rank 0 prints ONE JSON line.
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


def cpu_baseline(pdu, tb, nof_ports, nof_subc, ofdm, budget_s=20.0):
    """Times the CPU path on the host cores for a bounded sample of the same workload (rank 0, N=1 only).
    Uses the compiled reference (oracle/_ref, kind "reference") when that library travelled with the repository,
    the C oracle (kind "port") otherwise."""
    import ctypes as C
    import backends
    cores = min(len(os.sched_getaffinity(0)), 16)
    r = backends.ref()
    if r is not None:
        kind = "reference"

        def run(threads, reps):
            return r.lib.ref_bench_pdsch(C.byref(pdu), tb.ctypes.data, nof_ports, nof_subc, C.byref(ofdm), threads,
                                         reps, 1)
    else:
        kind = "port"
        o = backends.oracle()

        def run(threads, reps):
            return o.lib.oracle_bench(C.byref(pdu), tb.ctypes.data, nof_ports, nof_subc, C.byref(ofdm), threads, reps)
    run(cores, 1)       # warms caches / page-faults the buffers
    t1 = run(cores, 4) / 4.0  # calibration: seconds per slot per thread
    reps = int(max(2, min(20000, budget_s / max(t1, 1e-4))))
    dt = run(cores, reps)
    return {
        "value": cores * reps / dt,
        "unit": "slots/s",
        "cores": cores,
        "kind": kind,
        "sample": "%d threads x %d config-3 slots each (PDSCH %s + OFDM generic radix-2 DFT), %.1f s" % (
            cores, reps, "AVX2 LDPC/precoder" if kind == "reference" else "scalar C oracle", dt),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--slots", type=int, default=1024, help="slots (config 4: cell-slots) per GPU per step")
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5],
                    help="BASELINE config: 3 = the headline workload (default); 2, 4 and 5 (receive-side add-on, one GPU) "
                         "are secondary measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.config == 5:
        # the receive-side chain has its own script (profiles/rx_chain_bench.py); same JSON schema, one GPU
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        import rx_chain_bench
        sys.argv = [sys.argv[0], "--steps", str(args.steps), "--warmup", str(args.warmup)] + (
            ["--slots", str(args.slots)] if args.slots != 1024 else [])
        rx_chain_bench.main()
        return

    import torch
    import backends
    import cases
    lib = backends.pkg.lib

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ctx = lib.Context(local_rank)
    slots = args.slots
    if args.config == 4:
        # four cells x four UEs (68 PRB each, QPSK / 16 / 64 / 256-QAM): one grid per cell-slot, four PDUs per grid
        pdus, grid_of = [], []
        for i in range(slots):
            cell, nof_ports, nof_subc = cases.mixed_cell(i % 4, slot_index=(i // 4) % 20)
            pdus += cell
            grid_of += [i] * len(cell)
        ofdm = cases.baseline_config(3)[3]
        workload = ("BASELINE config 4: 100 MHz cell-slots, 4 cells x 4 UEs (68 PRB each; QPSK 120, 16-QAM 658, "
                    "64-QAM 873, 256-QAM 948; 4 layers) + 4-port OFDM")
    else:
        nof_ports, nof_subc, ofdm = cases.baseline_config(args.config)[1:]
        pdus = [cases.baseline_config(args.config, slot_index=i % (20 if args.config == 3 else 10))[0]
                for i in range(slots)]
        grid_of = list(range(slots))
        d0 = lib.derive(pdus[0])
        workload = {
            3: "BASELINE config 3: 100 MHz (FFT 4096, 30 kHz SCS, 273 PRB grid) 4-layer 256-QAM R=948/1024 full-TBS "
               "PDSCH (TBS %d bit, %d CB, BG1 Zc %d) + 4-port OFDM",
            2: "BASELINE config 2: 20 MHz (FFT 2048, 15 kHz SCS, 106 PRB) 2-layer 64-QAM R=873/1024 PDSCH "
               "(TBS %d bit, %d CB, BG1 Zc %d) + 2-port OFDM",
        }[args.config] % (8 * pdus[0].tb_size_bytes, d0["nof_codeblocks"], d0["lifting_size"])
    pdu0 = pdus[0]
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
    d_tb = tb_sets[0]
    plan = lib.PdschPlan(ctx, pdus, tb_offsets, grid_of, slots, nof_ports, nof_subc)
    oplan = lib.OfdmPlan(ctx, ofdm, nof_ports)
    d_grid = torch.zeros((slots, nof_ports, 14, nof_subc), dtype=torch.int32, device="cuda")
    d_iq = torch.zeros((slots, nof_ports, oplan.slot_stride, 2), dtype=torch.float32, device="cuda")
    sps = 2 if args.config != 2 else 1  # slots per subframe
    d_slot = torch.tensor([i % sps for i in range(slots)], dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    step_no = [0]

    def step():
        plan.run(tb_sets[step_no[0] % len(tb_sets)], d_grid, zero_grids=True)
        oplan.run(slots, d_grid, d_iq, d_slot_index=d_slot)
        step_no[0] += 1

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    plan.enable_timing(args.steps)
    oplan.enable_timing(args.steps)
    barrier()
    torch.cuda.synchronize()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    # Whole-job totals: slots and IQ samples summed over ranks, time = max over ranks (RCCL all-reduce of 3 numbers).
    samples_per_slot = nof_ports * oplan.slot_stride
    total_slots, total_samples, dt = backends.pkg.sharding.aggregate(
        dist, torch.device("cuda", local_rank), slots * args.steps, slots * args.steps * samples_per_slot, dt)

    (ms_crc, ms_cb, ms_dmrs, ms_run), _ = plan.kernel_times()
    ms_ofdm, _ = oplan.kernel_time()

    if rank == 0:
        # Algorithmic bytes per slot: TB read + grid written once (incl. zeros) +
        # grid read by the OFDM modulator + IQ write.
        grid_bytes = nof_ports * 14 * nof_subc * 4
        alg_pdsch = tb_bytes_per_step / slots + grid_bytes
        alg_ofdm = grid_bytes + samples_per_slot * 8
        ofdm_name = "ofdm_kernel<%d>" % ofdm.dft_size
        kernels = {
            # name: (avg ms per launch, algorithmic bytes per launch)
            ofdm_name: (ms_ofdm, slots * alg_ofdm),
            # The codeblock launch also carries the DM-RS and zero-fill waves: it writes every grid word exactly once.
            "codeblock_kernel": (ms_cb, tb_bytes_per_step + slots * grid_bytes),
        }
        dom = max(kernels, key=lambda k: kernels[k][0])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if args.config == 3 and tj.get("slots") == slots and dom in tj.get("hbm_bytes_per_launch", {}):
                    traffic = tj["hbm_bytes_per_launch"][dom]
            except Exception:
                traffic = None

        def roof(name):
            ms, nbytes = kernels[name]
            gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"kernel": name, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "ms_per_launch": round(ms, 4),
                    "algorithmic_bytes_per_launch": int(nbytes)}

        roofline = roof(dom)
        roofline["traffic"] = traffic
        out = {
            "metric": "pdsch_slots_per_sec",
            "value": round(total_slots / dt, 1),
            "unit": "slots/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 bit-packed GF(2) + bf16 grid + f32 IQ",
            "data": "synthetic",
            "config": {"workload": workload, "slots_per_gpu_per_step": slots, "parallelism": "slot-sharded x%d, no data-path collective" % world},
            "iq_gsamples_per_sec": round(total_samples / dt / 1e9, 3),
            "whole_path_hbm_frac": round(total_slots * (alg_pdsch + alg_ofdm) / dt / 1e9 / (HBM_PEAK_GBS * world), 4),
            "kernel_ms": {"prologue_tbcrc_scrambling_seq": round(ms_crc, 4), "codeblock_dmrs_zerofill": round(ms_cb, 4),
                          "separate_dmrs": round(ms_dmrs, 4), "pdsch_run": round(ms_run, 4), "ofdm": round(ms_ofdm, 4)},
            "roofline": roofline,
            "roofline_other": [roof(k) for k in kernels if k != dom],
        }
        # Sanity: slot 0 of the last step against the CPU oracle (checker only, outside the timed region).
        try:
            o = backends.oracle()
            want = None
            for k, q in enumerate(pdus):  # the PDUs of grid 0 map disjoint RE: their grids OR together
                if grid_of[k] != 0:
                    break
                last = tb_sets[(step_no[0] - 1) % len(tb_sets)]
                tbk = last[tb_offsets[k]: tb_offsets[k] + q.tb_size_bytes].cpu().numpy()
                part = o.pdsch_process(q, tbk, nof_ports, nof_subc)
                want = part if want is None else np.bitwise_or(want, part)
            got = d_grid[0].cpu().numpy().view(np.uint16).reshape(want.shape)
            out["verified_vs_oracle"] = bool(np.array_equal(got, want))
        except Exception as e:  # the oracle is optional at bench time
            out["verified_vs_oracle"] = "unavailable: %s" % e
        if world == 1 and not args.no_cpu_baseline and args.config == 3:
            tb_host = d_tb[: pdu0.tb_size_bytes].cpu().numpy().copy()
            out["cpu_baseline"] = cpu_baseline(pdus[0], tb_host, nof_ports, nof_subc, ofdm)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
