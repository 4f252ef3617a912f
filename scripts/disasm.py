#!/usr/bin/env python
"""Disassemble kernels of the gfx950 code objects inside nnlm_amd/libnnlm_mi355x.so (or of one object file of nnlm_amd/build/).

    python scripts/disasm.py 'kl_tile_kernel<10, 2, 4, false>' > /tmp/k.s      # substring of the demangled name
    python scripts/disasm.py --list | grep kl_tile
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path, d):
    fat = os.path.join(d, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, path])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
    out = []
    for n, (b0, b1) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        part, co = os.path.join(d, f"fat{n}.bin"), os.path.join(d, f"dev{n}.co")
        open(part, "wb").write(blob[b0:b1])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        out.append(co)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    so = os.path.join(ROOT, "nnlm_amd", "libnnlm_mi355x.so")
    for a in sys.argv[1:]:
        if a.startswith("--obj="):
            so = a[6:]
    with tempfile.TemporaryDirectory() as d:
        for co in code_objects(so, d):
            syms = subprocess.run(["nm", "--defined-only", co], capture_output=True, text=True, check=True).stdout
            names = [ln.split()[-1] for ln in syms.splitlines() if " T " in ln or " t " in ln]
            dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
            for nm, dn in zip(names, dem):
                if "--list" in sys.argv:
                    print(dn)
                    continue
                if any(a in dn for a in args):
                    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--disassemble-symbols=" + nm, co],
                                         capture_output=True, text=True, check=True).stdout
                    print("; ====", dn)
                    print(txt)


if __name__ == "__main__":
    main()
