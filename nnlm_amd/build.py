"""Build libnnlm_mi355x.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "nnlm_mi355x.hip")
OUT = os.path.join(HERE, "libnnlm_mi355x.so")


def sources():
    d = os.path.join(HERE, "csrc")
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".h"))] + [
        os.path.join(HERE, "..", "include", "nnlm_mi355x.h")]


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in sources()):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-mfma-vgpr-form: keep MFMA C/D operands in VGPRs (gfx950 has a unified register file); the sweep kernel
    # reads and rewrites single accumulator entries between MFMAs and would otherwise shuttle whole tiles VGPR<->AGPR
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form",
           "-shared", "-fPIC", "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
