#!/usr/bin/env python3
"""Performance sanity sweep over PDU shapes (not a bench line): per shape, a plan of --slots identical-shape PDUs,
HIP-event time of the PDSCH launches, and oracle verification of slot 0.  Catches shapes whose work lists degenerate
(e.g. zero-fill of comb patterns).  Usage (GPU box, repository root): python3 profiles/shape_sweep.py [--slots 512]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=512)
    args = ap.parse_args()
    import torch
    import backends
    import cases
    lib, abi = backends.pkg.lib, backends.abi
    ctx = lib.Context(0)
    oracle = backends.oracle()
    w4 = cases.codebook("four_layer_four_ports_0_0")
    w2 = cases.codebook("two_layer_two_ports_0")
    w1 = cases.codebook("single_port")
    comb = [(range(0, 273), [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0])]

    def pdu(qm, rate, w, n_prb, bw=273, **kw):
        layers = w.shape[2]
        nsym = kw.get("nof_symbols", 12)
        tb_bits = cases.tbs(nsym, 36, qm, rate, layers, n_prb)
        bg = 2 if (rate <= 256 or tb_bits <= 292 or (tb_bits <= 3824 and rate <= 686)) else 1
        args_ = dict(qm=qm, bwp_size_rb=bw, dmrs_symbols=(2, 7, 11), prb_start=0, prb_count=n_prb, nof_symbols=12,
                     base_graph=bg, precoding=w, tb_size_bytes=tb_bits // 8)
        args_.update(kw)
        return abi.make_pdu(**args_)

    shapes = {
        "cfg3 4L 256QAM 270PRB": lambda i: pdu(8, 948, w4, 270, slot_index=i % 20),
        "4L QPSK R=120 270PRB (BG2, repetition)": lambda i: pdu(2, 120, w4, 270),
        "4L 16QAM R=658 270PRB rv2": lambda i: pdu(4, 658, w4, 270, rv=2),
        "4L 256QAM 270PRB + CSI-RS comb reserved": lambda i: pdu(8, 948, w4, 270, reserved=comb),
        "2L 64QAM 273PRB cdm2 (zero other group)": lambda i: pdu(6, 873, w2, 273),
        "2L 64QAM 273PRB cdm1 (data beside DM-RS)": lambda i: pdu(6, 873, w2, 273, nof_cdm_groups_without_data=1),
        "1L QPSK 52PRB (cfg1 shape, bw 52)": lambda i: pdu(2, 120, w1, 52, bw=52),
        "1L 256QAM 273PRB": lambda i: pdu(8, 948, w1, 273),
        "4L 64QAM 4PRB (tiny)": lambda i: pdu(6, 600, w4, 4),
    }
    for name, make in shapes.items():
        slots = args.slots
        pdus = [make(i) for i in range(slots)]
        ports, subc = pdus[0].nof_ports, pdus[0].bwp_size_rb * 12
        stride = (pdus[0].tb_size_bytes + 255) & ~255
        d_tb = torch.randint(0, 256, (slots * stride,), dtype=torch.uint8, device="cuda")
        plan = lib.PdschPlan(ctx, pdus, [i * stride for i in range(slots)], list(range(slots)), slots, ports, subc)
        d_grid = torch.zeros((slots, ports, 14, subc), dtype=torch.int32, device="cuda")
        for _ in range(2):
            plan.run(d_tb, d_grid, zero_grids=True)
        plan.enable_timing(5)
        for _ in range(5):
            plan.run(d_tb, d_grid, zero_grids=True)
        ctx.synchronize()
        torch.cuda.synchronize()
        (ms_pro, ms_cb, ms_dmrs, ms_run), _ = plan.kernel_times()
        tb0 = d_tb[: pdus[0].tb_size_bytes].cpu().numpy()
        want = oracle.pdsch_process(pdus[0], tb0, ports, subc)
        got = d_grid[0].cpu().numpy().view(np.uint16).reshape(want.shape)
        d = lib.derive(pdus[0])
        print("%-44s CB %3d  prologue %.3f  codeblock %.3f  run %.3f ms  -> %7.0f k slots/s  %s" % (
            name, d["nof_codeblocks"], ms_pro, ms_cb, ms_run, slots / ms_run, "ok" if np.array_equal(got, want) else "MISMATCH"),
            flush=True)
        del plan, d_grid, d_tb


if __name__ == "__main__":
    main()
