#!/usr/bin/env python3
"""What actually ships: the gfx950 code objects inside libmi355nrphy.so (section .hip_fatbin -> clang offload bundles -> ELF).
`hipcc -c -save-temps` listings and -Rpass-analysis remarks come from a separate compile and can differ from the library's own
code (round 4: a multiply-add fused only in the real build).  No GPU needed.

  python3 profiles/disasm_lib.py                      resources of every kernel (from the code objects' metadata notes)
  python3 profiles/disasm_lib.py KERNEL_SUBSTRING     disassembly of the matching kernels on stdout
  python3 profiles/disasm_lib.py --lib path/to/variant.so ..."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib):
    tmp = tempfile.mkdtemp()
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    data, magic, out, pos = open(fat, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, size, tsize = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24:off + 24 + tsize].decode()
            off += 24 + tsize
            if "gfx950" in triple and size:
                path = os.path.join(tmp, "co%d.elf" % len(out))
                open(path, "wb").write(data[i + o:i + o + size])
                out.append(path)
        pos = i + 24
    return out


def demangle(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]


def resources(lib):
    rows = []
    for co in code_objects(lib):
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        for block in notes.split("- .agpr_count")[1:]:
            get = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, block) or [None, "?"])[1]
            rows.append((demangle(get("name")), get("sgpr_count"), get("vgpr_count"), get("sgpr_spill_count"), get("vgpr_spill_count"),
                         get("private_segment_fixed_size"), get("group_segment_fixed_size")))
    return rows


if __name__ == "__main__":
    args = sys.argv[1:]
    lib = os.path.join(ROOT, "srsran-edgeric-5g_amd", "csrc", "libmi355nrphy.so")
    if args[:1] == ["--lib"]:
        lib, args = args[1], args[2:]
    if not args:
        print("%-72s %5s %5s %10s %10s %7s %7s" % ("kernel", "SGPR", "VGPR", "SGPR spill", "VGPR spill", "scratch", "LDS"))
        for r in sorted(resources(lib)):
            print("%-72s %5s %5s %10s %10s %7s %7s" % ((r[0][-72:],) + r[1:]))
    else:
        for co in code_objects(lib):
            text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
            for m in re.finditer(r"^[0-9a-f]+ <(\S+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", text, re.S | re.M):
                if args[0] in m.group(1) or args[0] in demangle(m.group(1)):
                    print("; %s" % demangle(m.group(1)))
                    print(m.group(2))
