"""Where the waves of one workgroup of corr_fwd_rw_kernel spend a step (diagnostic build).  Build a traced copy of the library and run:

    hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fno-vectorize -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden \\
          -DUNFLOW_CORR_TRACE=16 -c unflow_amd/csrc/correlation_planes.hip -o scratch/corr_trace.o        # 16: an interior row group
    hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libunflow_trace.so scratch/corr_trace.o <the other .o files> -ldl
    UNFLOW_LIB_PATH=scratch/libunflow_trace.so python tools/debug/corr_phase_trace.py

Every wave of that workgroup sums shader cycles (s_memtime) per phase in registers: prologue, waiting for the tile, barrier, dispatch,
tail + blocks without products, blocks with products (+ finish), and counts its product steps.  The stamps are scheduling barriers, so
the traced build interleaves a little less than the shipped one; small phases are only indicative (profiles/r04_corr_fwd_rw.txt)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd import _lib
from unflow_amd._lib import ptr, stream, check, planes_of
from unflow_amd.core import layers as L
dev = torch.device("cuda:0")
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
pt = L.PT.alloc((8, 48, 64, 256), dev, 3)
pt.t.copy_(torch.randn(8, 48, 64, 256, generator=g).to(dev))
L.planes_from_f32(pt.t, pt.pl)
f = pt.t
st = stream()
co = torch.zeros((8, 48, 64, 441), device=dev)
for i in range(5):
    check(lib.unflow_correlation_nhwc_fwd_pl(ptr(f), ptr(f), 256, planes_of(pt.pl), planes_of(pt.pl), 4, ptr(co), 441, 8, 256, 48, 64, 1, 20, 20, 1, 2, st))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
raw = ctypes.CDLL(os.environ["UNFLOW_LIB_PATH"])
print("rc", raw.unflow_debug_corr_trace(buf))
names = ["prologue", "wait tile", "barrier", "next+masks", "tail/other", "MF+FIN blk", "MF blk", "n MF steps"]
print("wave " + " ".join("%11s" % n for n in names) + "   total")
for w in range(8):
    v = [buf[w * 8 + i] for i in range(8)]
    print("%4d " % w + " ".join("%11d" % x for x in v) + "   %d" % sum(v[:7]))
