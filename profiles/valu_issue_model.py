#!/usr/bin/env python3
"""Vector-issue cost model of the kernels the bench prices against the vector-issue roof.

The roof used until round 3 -- one vector instruction per FOUR cycles and SIMD -- is a convention: profiles/r01_valu_rate.txt
(profiles/probes/valu_rate.hip, measured on MI355X at four waves per SIMD) shows 2.57 cycles for the plain VOP2 integer
instructions, 3.44 for v_fma_f32, 4.2-4.4 for most VOP3, 4.76-5.42 for packed FP32.  This script prices a kernel by its own
instruction mix: the static histogram of vector mnemonics in the kernel's ISA (hipcc -save-temps; the hot loops are unrolled, so
they dominate the static code as they dominate the dynamic count) x the measured cost of each class = average issue cycles per
vector instruction; bench.py multiplies the dynamic instruction count of the PMC profile (SQ_INSTS_VALU, profiles/traffic.json)
by it: issue_cycles = instructions x average cost, frac = issue_cycles / (1,024 SIMDs x launch time x 2.4 GHz) -- the probe's
cycles are defined on the nominal 2.4 GHz clock, so the roof uses the same; the clock the engines really hold under the kernel
(SQ_BUSY_CYCLES / 32 / time) is reported next to it.

No GPU needed.  Usage: python3 profiles/valu_issue_model.py            -> JSON on stdout (merged into profiles/traffic.json by
                       python3 profiles/valu_issue_model.py --update)"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "srsran-edgeric-5g_amd", "csrc")

# cycles per instruction and SIMD with several waves (profiles/r01_valu_rate.txt, r04_valu_rate.txt); classes by mnemonic
COSTS = [
    (r"^v_(add|sub|subrev)_(u32|i32|co_u32)|^v_(xor|and|or|not)_b32|^v_mov_b32|^v_(lshlrev|lshrrev|ashrrev)_b32", 2.57, "plain VOP1/VOP2 integer"),
    (r"^v_bitop3_b32", 2.81, "v_bitop3"),
    (r"^v_(fma|mul|add|sub|mac|fmac|max|min)_f32|^v_cvt_(f32|u32|i32)_", 3.44, "scalar FP32"),
    (r"^v_pk_(fma)_f32", 5.42, "packed FP32 fma"),
    (r"^v_pk_(add|mul)_f32", 4.76, "packed FP32 add / mul"),
    (r"^v_pk_", 4.36, "packed 16-bit (VOP3P; profiles/r04_valu_rate.txt: 4.32-4.40)"),
    (r"^v_(cmp|cmpx)_", 4.73, "compare"),
    (r"^v_(readlane|readfirstlane|writelane)", 4.35, "cross-lane"),
]
DEFAULT = (4.33, "other VOP3 / DPP / SDWA")   # v_med3, v_alignbit, v_mad_u32_u24, v_bfe, v_lshl_or, v_add3, v_cndmask, v_perm ...

KERNELS = {   # bench.py's name -> (source, mangled-name pattern, contraction flag)
    "codeblock_kernel": ("pdsch_kernels.hip", r"codeblock_kernel_tILi8ELi4E", "-ffp-contract=off"),
    "prologue_kernel": ("pdsch_kernels.hip", r"prologue_kernel", "-ffp-contract=off"),
    "ofdm_kernel<4096>": ("ofdm_kernels.hip", r"ofdm_kernelILi4096ELi1ELb0E", "-ffp-contract=off"),
    "ofdm_kernel<4096, ci16>": ("ofdm_kernels.hip", r"ofdm_kernelILi4096ELi2ELb1E", "-ffp-contract=off"),
    "ldpc_decode_msg_bg1_kernel": ("ldpc_decoder.hip", r"ldpc_decode_msg_bg1_kernel", "-ffp-contract=off"),
    "ldpc_decode_msg_bg2_slot_kernel": ("ldpc_decoder.hip", r"ldpc_decode_msg_bg2_slot_kernel", "-ffp-contract=off"),
}


def source_sha():
    """sha256 over the device sources: bench.py refuses a traffic.json measured on other kernels."""
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h", ".inc")):
            h.update(name.encode())
            h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()


def isa_of(source, contract):
    tmp = tempfile.mkdtemp()
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                    "-I" + CSRC, contract, "-x", "hip", "-c", os.path.join(CSRC, source), "-o", os.path.join(tmp, "o.o"), "-save-temps"],
                   cwd=tmp, check=True, capture_output=True)
    for f in os.listdir(tmp):
        if f.endswith("gfx950.s"):
            return open(os.path.join(tmp, f)).read()
    raise RuntimeError("no ISA")


def price(body):
    hist, classes, total, cycles = {}, {}, 0, 0.0
    for line in body.split("\n"):
        m = re.match(r"\s+(v_[a-z0-9_]+)", line)
        if not m:
            continue
        mn = m.group(1)
        for pat, cost, label in COSTS:
            if re.match(pat, mn):
                break
        else:
            cost, label = DEFAULT
        hist[mn] = hist.get(mn, 0) + 1
        c = classes.setdefault(label, {"instructions": 0, "cycles_each": cost})
        c["instructions"] += 1
        total += 1
        cycles += cost
    return total, cycles, classes, hist


def shipped_isa():
    """Disassembly of the code objects inside the built library (profiles/disasm_lib.py): what really runs -- a separate
    `hipcc -c -save-temps` compile can differ from it (round 4: a multiply-add fused only in the library).  None without a library."""
    lib = os.path.join(CSRC, "libmi355nrphy.so")
    if not os.path.exists(lib):
        return None
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import disasm_lib
    text = ""
    for co in disasm_lib.code_objects(lib):
        dis = subprocess.run([os.path.join(disasm_lib.LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
        # objdump's "0000... <symbol>:" headers in the shape of the assembler listing's "symbol:" labels
        text += re.sub(r"^[0-9a-f]+ <(\S+)>:$", r"\1:", dis, flags=re.M)
    return text


def model():
    out, isa = {}, {}
    shipped = shipped_isa()
    for name, (src, pat, contract) in KERNELS.items():
        if shipped is not None:
            text = shipped
        else:
            if src not in isa:
                isa[src] = isa_of(src, contract)
            text = isa[src]
        m = re.search(r"^(_ZN5nrphy\w*%s\w*):" % pat, text, re.M)
        if not m:
            continue
        a = m.end()
        body = text[a: text.index("s_endpgm", a)]
        total, cycles, classes, hist = price(body)
        out[name] = {"static_vector_instructions": total, "avg_issue_cycles_per_instruction": round(cycles / total, 3),
                     "classes": classes, "top_mnemonics": dict(sorted(hist.items(), key=lambda kv: -kv[1])[:12])}
    return {"kernel_source_sha256": source_sha(), "valu_issue_model": out,
            "valu_issue_model_source": "profiles/valu_issue_model.py: static instruction mix of the %s x profiles/r01_valu_rate.txt costs" % (
                "code objects inside the built library" if shipped is not None else "kernels' ISA (hipcc -save-temps)")}


if __name__ == "__main__":
    res = model()
    if "--update" in sys.argv:
        path = os.path.join(ROOT, "profiles", "traffic.json")
        tj = json.load(open(path))
        tj.update({"valu_issue_model": res["valu_issue_model"], "valu_issue_model_source": res["valu_issue_model_source"]})
        json.dump(tj, open(path, "w"), indent=1)
        print("updated", path)
    else:
        print(json.dumps(res, indent=1))
