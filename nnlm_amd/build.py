"""Build libnnlm_mi355x.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library is a handful of translation units (nnlm_amd/csrc/*.hip) compiled in parallel and linked into one shared object: the SCD
sweep kernels of k_sweep_q.h (one heavy instantiation per block count, mask and arithmetic mode) take as long as everything else
together, so they have units of their own (tu_sweepq.hip, tu_sweepqw.hip, tu_sweepf.hip)."""
from __future__ import annotations

import hashlib
import os
import signal
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libnnlm_mi355x.so")
OBJ = os.path.join(HERE, "build")


def units():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))] + [
        os.path.join(HERE, "..", "include", "nnlm_mi355x.h")]


def unit_deps(src):
    """The unit's source and every header it includes (quoted includes, transitively)."""
    import re
    seen, todo = set(), [os.path.abspath(src)]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(f):
            continue
        seen.add(f)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(f).read(), re.M):
            todo.append(os.path.normpath(os.path.join(os.path.dirname(f), inc)))
    return seen


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in sources()):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    # -amdgpu-mfma-vgpr-form: keep MFMA C/D operands in VGPRs (gfx950 has a unified register file); the sweep kernel
    # reads and rewrites single accumulator entries between MFMAs and would otherwise shuttle whole tiles VGPR<->AGPR
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fPIC"]
    # an object is fresh when it is newer than its unit and every header the unit includes AND was built by this compiler with these
    # flags (recorded next to it); objects are written under a temporary name and renamed on success, so a killed compiler cannot leave
    # a truncated file that the next run would trust
    stamp = hashlib.sha256((" ".join([os.path.realpath(hipcc)] + flags)).encode()).hexdigest()
    procs, fresh = [], []
    for src in units():  # every stale unit at once: a few processes, each minutes long
        obj = os.path.join(OBJ, os.path.splitext(os.path.basename(src))[0] + ".o")
        tag = obj + ".flags"
        same_flags = os.path.exists(tag) and open(tag).read() == stamp
        if not force and same_flags and os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in unit_deps(src)):
            fresh.append(obj)  # (the unit and its headers are older than its object)
            continue
        cmd = [hipcc] + flags + ["-c", src, "-o", obj + ".tmp"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, obj, subprocess.Popen(cmd, start_new_session=True)))  # (own process group: hipcc is a wrapper around clang children)
    objs = []
    for cmd, obj, p in procs:
        if p.wait() != 0:
            for _, _, q in procs:
                if q.poll() is None:
                    try:
                        os.killpg(q.pid, signal.SIGKILL)
                    except ProcessLookupError:
                        pass
            raise subprocess.CalledProcessError(p.returncode, cmd)
        os.replace(obj + ".tmp", obj)
        with open(obj + ".flags", "w") as f:
            f.write(stamp)
        objs.append(obj)
    objs += fresh
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
