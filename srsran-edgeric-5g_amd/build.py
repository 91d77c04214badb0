"""Builds csrc/libmi355nrphy.so for gfx950 with hipcc (cross-compiles without a GPU).

    python srsran-edgeric-5g_amd/build.py [--force]

The library is built in-tree so that it travels with the repository snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmi355nrphy.so")
SOURCES = ["nrphy_host.cpp", "dl_control_host.cpp", "pdsch_async.cpp", "dl_slot_async.cpp", "pdsch_kernels.hip", "ofdm_kernels.hip", "ldpc_decoder.hip",
           "ldpc_dematcher.hip", "pusch_decoder.hip", "csi_rs_kernels.hip", "dl_control_kernels.hip", "lower_phy_kernels.hip", "demod_kernels.hip"]
HEADERS = ["nrphy_internal.h", "nrphy_host_internal.h", "nrphy_trace.h", "bits_device.h", "ldpc_device.h", "nr_ldpc_bg.inc", "nr_polar_tables.inc",
           os.path.join(ROOT, "include", "mi355_nrphy.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# Contraction is off everywhere: the kernels say where they want a fused multiply-add (explicit __fmaf_rn / v_pk_fma_f32).  Until
# late round 4 the FFT file was built with -ffp-contract=fast; under that flag the backend fuses across `#pragma clang fp
# contract(off)`, which cost the wire-format sink its exact rounding (profiles/r04_fuzz_sweep_summary.txt), and the transforms of
# the sizes that matter (powers of two) come out instruction for instruction the same without it -- the 3 * 2^k sizes lose 4-12 of
# ~630-880 vector instructions per symbol.
CONTRACT = {}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compiles every HIP source for gfx950 and links the shared library; returns its path."""
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [HIPCC] + FLAGS + [CONTRACT.get(src, "-ffp-contract=off"), "-x", "hip", "-c", os.path.join(CSRC, src),
               "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for cmd, p in procs:
        out, _ = p.communicate(timeout=1800)
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.run(cmd, check=True, timeout=600)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
