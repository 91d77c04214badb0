#!/usr/bin/env python3
"""What the compiler made of the headline RE loop (codeblock_kernel_t<8, 4>, four ports, wideband precoding): compiles
csrc/pdsch_kernels.hip with the library's flags (no GPU needed), cuts the kernel's last loop with sixteen scalar-operand
v_pk_mul_f32 out of the ISA and counts what must not be there -- scalar loads of the weights and v_readlane / v_writelane
spill traffic (round 2: one s_load_dwordx8 + s_waitcnt per port and 64 RE, 2,387 v_readlane in the kernel).
Usage: python3 profiles/check_re_loop.py > profiles/r03_re_loop_check.txt"""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "srsran-edgeric-5g_amd", "csrc")
with tempfile.TemporaryDirectory() as tmp:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + C,
                    "-ffp-contract=off", "-x", "hip", "-c", os.path.join(C, "pdsch_kernels.hip"), "-o", os.path.join(tmp, "o.o"), "-save-temps"],
                   cwd=tmp, check=True, capture_output=True)
    isa = open(os.path.join(tmp, "pdsch_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
begin = next(i for i, l in enumerate(isa) if l.startswith("_ZN5nrphy18codeblock_kernel_tILi8ELi4EEE"))
end = next(i for i in range(begin, len(isa)) if "s_endpgm" in isa[i])
k = isa[begin:end + 1]
code = [l for l in k if l.startswith("\t") and not l.strip().startswith(";")]
print("codeblock_kernel_t<8, 4>: %d instructions, v_readlane %d, v_writelane %d (the v_readlane are the wave reductions' row folds)" % (
    len(code), sum("v_readlane" in l for l in code), sum("v_writelane" in l for l in code)))
idx = [i for i, l in enumerate(k) if "v_pk_mul_f32" in l and ", s[" in l][-16:]
start = max(i for i in range(idx[0]) if "Loop Header" in k[i])
stop = next(i for i in range(idx[-1], len(k)) if "s_cbranch" in k[i] or "s_endpgm" in k[i])
body = [l for l in k[start:stop + 1] if l.startswith("\t") and not l.strip().startswith(";")]
print("grid loop, four ports (from its loop header to the branch behind the last product): %d instructions" % len(body))
for what in ("v_pk_mul_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_cvt_pk_bf16_f32", "buffer_store_dword", "ds_read", "global_load", "s_load", "v_readlane", "v_writelane"):
    print("  %-20s %d" % (what, sum(what in l for l in body)))
print("  scalar loads in the loop region (symbol change and the rare group that straddles a symbol; none per 64 RE, none of weights):")
for l in body:
    if "s_load" in l:
        print("     " + l.strip())
