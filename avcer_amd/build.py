"""Build recipe for libavcer_hip.so (hipcc, gfx950 only, in-tree so that the .so travels with the repo snapshot)."""
from __future__ import annotations

import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libavcer_hip.so")
SOURCES = ("gemm.hip", "fused.hip", "kernels.hip", "api.hip")
HEADERS = ("common.h", "gemm_dev.h", "split_dev.h", os.path.join("..", "..", "include", "avcer_hip.h"))
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libavcer_hip.so cannot be built")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None, tag: str = "") -> str:
    """Compile every translation unit to an object and link the shared library. Returns the .so path.
    `extra_flags` / `out` / `tag`: lab builds only (tools/build_lab.sh: the SAME sources and flags plus e.g. a -D switch, objects
    suffixed with `tag`, library written to `out`) -- the product build passes none of them."""
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    lib = out or LIB
    objdir = os.path.dirname(lib) if out else CSRC
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", tag + ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        outp, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), outp))
        if verbose and outp.strip():
            print(outp)
    if force or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return lib


if __name__ == "__main__":
    print(build(force="--force" in os.sys.argv, verbose=True))
