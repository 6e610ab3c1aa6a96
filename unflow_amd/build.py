"""Builds unflow_amd/csrc/libunflow_hip.so for gfx950 with hipcc (in-tree, so the .so travels with
the repo snapshot).  Counterpart of the reference's JIT nvcc/g++ build in src/e2eflow/ops.py:21-48."""
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libunflow_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(out, deps):
    return (not os.path.exists(out)) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force=False, verbose=False):
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(CSRC, "..", "..", "include", "unflow_hip.h")]
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
