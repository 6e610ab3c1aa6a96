"""Builds unflow_amd/csrc/libunflow_hip.so for gfx950 with hipcc (in-tree, so the .so travels with
the repo snapshot).  Counterpart of the reference's JIT nvcc/g++ build in src/e2eflow/ops.py:21-48."""
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libunflow_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize -fno-vectorize: NO compiler-formed packed-fp32 instructions (v_pk_mul/add/fma_f32) anywhere in the
# code object (checked: `llvm-objdump -d | grep -c "v_pk_.*_f32"` = 0; with only the SLP vectoriser off the loop vectoriser
# still left 98 of them in the warp / augmentation / resize kernels).  hipcc (ROCm 7.2) allocates some of them with the
# destination pair overlapping a source pair that another half reads through op_sel, e.g.
#   v_pk_mul_f32 v[80:81], v[74:75], v[80:81] op_sel:[0,1]      (lo = v74 * v81, hi = v75 * v81 -> v81)
# and on gfx950 the LOW half of exactly that instruction came out wrong in ~10 % of the replays of the two-branch backward
# graph — only while a kernel of the other branch shared the SIMD, never alone, never eagerly; the inputs were verified
# intact after the replay (tools/debug/wgstream_flake.py: 13 wrong sums in 120 replays with packed ops, 0 in 100 without;
# the stress test stays in the GPU suite: tests/test_engine_gpu.py).  The packed forms buy nothing here (the step is 0.6 %
# FASTER without them: they are an anti-lever beside MFMAs, MI355X_MICROARCH.md).
FLAGS = ["--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-fno-vectorize", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


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
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]      # dlopen: csrc/comm_rccl.hip
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
