#!/usr/bin/env python3
"""Phase timeline of one wave of the LDS-message decoder (ldpc_decode_pairs_lm_kernel): build the library with -DNRPHY_DEC_TRACE
(add it to FLAGS in srsran-edgeric-5g_amd/build.py, `build.py --force`), then on the GPU box: python3 profiles/probes/decoder_trace_run.py
Prints the cycles one wave of the middle codeblock spent between the marks TR(k) of ldpc_decoder.hip, summed over the layers
of iterations 2 .. 8 (28 layers); every mark costs ~300 cycles of its own (s_memtime + wait)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib
lib = importlib.import_module("srsran-edgeric-5g_amd.lib"); abi = importlib.import_module("srsran-edgeric-5g_amd.abi")
ctx = lib.Context()
bg, zc, nf = 1, 384, 72
n_cb = 6656
nof_llr = 26*384 - 2*384
rng = np.random.default_rng(1)
llr = rng.integers(-20, 21, (n_cb, nof_llr), dtype=np.int8)
cfg = abi.LdpcDecoderCfg(bg, zc, nf, 0, nof_llr, 8, 0.8)
d_llr = torch.from_numpy(llr).cuda()
out = torch.zeros((n_cb, 22*384//8), dtype=torch.uint8, device="cuda")
its = torch.zeros(n_cb, dtype=torch.int32, device="cuda")
sb = ctx.ldpc_decoder_scratch_bytes(cfg, n_cb)
scratch = torch.zeros(sb, dtype=torch.uint8, device="cuda")
for _ in range(3):
    ctx.ldpc_decode(cfg, n_cb, d_llr, nof_llr, out, out.shape[1], its, d_scratch=scratch)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); ctx.ldpc_decode(cfg, n_cb, d_llr, nof_llr, out, out.shape[1], its, d_scratch=scratch); e1.record(); torch.cuda.synchronize()
print("ms", e0.elapsed_time(e1))
t = scratch[:80].cpu().numpy().view(np.uint64)
names = ["enter(prev barrier->fn)", "addresses", "soft loads", "msg loads", "forward", "lookups+prep", "backward", "barrier"]
tot = t[:8].sum()
for n, v in zip(names, t[:8]): print("%-26s %8d  per layer %7.0f  %4.1f%%" % (n, v, v / 28.0, 100.0 * v / max(tot,1)))
print("total per layer", tot / 28.0)
