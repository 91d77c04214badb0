"""LDPC decoder throughput (BASELINE config 5 shape: the 104 codeblocks of a config-3 slot, BG1 Zc=384, E=8960).
Usage: python3 profiles/decoder_bench.py [n_slots] [iterations]   (run on the GPU box)"""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("srsran-edgeric-5g_amd")
abi, lib = pkg.abi, pkg.lib


def main():
    n_slots = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    ctx = lib.Context(0)
    bg, zc, e, n_cb = 1, 384, 8960, 104 * n_slots
    k = 22 * zc
    rng = np.random.default_rng(5)
    msgs = rng.integers(0, 256, (n_cb, k // 8), dtype=np.uint8)
    d_msg = torch.from_numpy(msgs).cuda()
    d_cb = torch.zeros((n_cb, e // 8), dtype=torch.uint8, device="cuda")
    ctx.ldpc_encode(bg, zc, d_msg, k // 8, e, d_cb, e // 8, n_cb)
    torch.cuda.synchronize()
    bits = np.unpackbits(d_cb.cpu().numpy(), axis=1).astype(np.float32)
    nof_llr = 66 * zc
    llr = np.zeros((n_cb, nof_llr), np.int8)
    llr[:, :e] = np.clip(np.rint((1 - 2 * bits) * 20 + rng.normal(0, 8.0, bits.shape)), -120, 120).astype(np.int8)
    d_llr = torch.from_numpy(llr).cuda()
    out = torch.zeros((n_cb, k // 8), dtype=torch.uint8, device="cuda")
    its = torch.zeros((n_cb,), dtype=torch.int32, device="cuda")
    res = {}
    for crc, label in ((0, "fixed_iterations"), (0x24B, "early_stop")):
        cfg = abi.LdpcDecoderCfg(bg, zc, 0, crc, nof_llr, iters, 0.8)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            for _ in range(2):
                ctx.ldpc_decode(cfg, n_cb, d_llr, nof_llr, out, k // 8, its, stream.cuda_stream)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            reps = 5
            for _ in range(reps):
                ctx.ldpc_decode(cfg, n_cb, d_llr, nof_llr, out, k // 8, its, stream.cuda_stream)
            b.record(stream)
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        res[label] = {"ms": ms, "codeblocks_per_s": n_cb / ms * 1e3, "slots_per_s": n_slots / ms * 1e3,
                      "info_gbps": n_cb * (k - 24) / ms * 1e-6, "mean_iterations": float(its.float().mean())}
    # rate dematcher in front (256-QAM, rv 0, new data, then a retransmission combined into the same soft buffers)
    rm_in = torch.from_numpy(np.ascontiguousarray(llr[:, :e])).cuda()
    soft = torch.zeros((n_cb, nof_llr), dtype=torch.int8, device="cuda")
    dcfg = abi.LdpcRateDematcherCfg(bg, zc, 0, 8, 0, 72, e)
    for new_data, label in ((True, "rate_dematch_new_data"), (False, "rate_dematch_combine")):
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            ctx.ldpc_rate_dematch(dcfg, n_cb, rm_in, e, soft, nof_llr, new_data, stream.cuda_stream)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(5):
                ctx.ldpc_rate_dematch(dcfg, n_cb, rm_in, e, soft, nof_llr, new_data, stream.cuda_stream)
            b.record(stream)
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        gb = n_cb * (e + nof_llr * (1 if new_data else 2)) * 1e-9
        res[label] = {"ms": ms, "codeblocks_per_s": n_cb / ms * 1e3, "algorithmic_GBps": gb / ms * 1e3}
    # without a CRC in the random messages early stop never fires: the second leg is the CRC cost on top
    # CPU context: the reference's own decoder (oracle/_ref, AVX2 and generic) on a few of the same codeblocks, 1 thread
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    try:
        import backends
        ref = backends.ref()
        if ref is None:
            raise RuntimeError("oracle/_ref not built")
        for simd, label in ((1, "reference_avx2"), (0, "reference_generic")):
            t0, n = time.perf_counter(), 0
            while time.perf_counter() - t0 < 2.0:
                ref.ldpc_decode(bg, zc, 0, 0, iters, 0.8, llr[n % n_cb], simd=simd)
                n += 1
            dt = time.perf_counter() - t0
            res[label] = {"codeblocks_per_s": n / dt, "info_gbps": n * (k - 24) / dt * 1e-9, "threads": 1}
    except Exception as e:  # the compiled reference is optional here
        res["reference"] = "unavailable: %s" % e
    print(json.dumps({"n_slots": n_slots, "n_cb": n_cb, "max_iterations": iters, **res}))


if __name__ == "__main__":
    main()
