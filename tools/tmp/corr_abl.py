import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd import _lib
from unflow_amd._lib import ptr, stream, check, planes_of
from unflow_amd.core import layers as L
from bench_ops import timeit
dev = torch.device("cuda:0")
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
pt = L.PT.alloc((8, 48, 64, 256), dev, 3)
pt.t.copy_(torch.randn(8, 48, 64, 256, generator=g).to(dev))
L.planes_from_f32(pt.t, pt.pl)
f = pt.t
st = stream()
co = torch.zeros((8, 48, 64, 441), device=dev)
def run():
    check(lib.unflow_correlation_nhwc_fwd_pl(ptr(f), ptr(f), 256, planes_of(pt.pl), planes_of(pt.pl), 4, ptr(co), 441, 8, 256, 48, 64, 1, 20, 20, 1, 2, st))
names = {0: "full", 64: "af loads all out of range", 512: "no af loads", 128: "no finish", 256: "no steps (prologue only)", 256|8: "no steps, no zero-fill", 256|8|512: "no steps, no zero-fill, no af loads",
         128|8: "no finish no zero-fill", 128|8|512: "no finish, zero-fill, af", 128|8|512|1: "... and no MFMA",  128|8|512|1|2: "... and no DMA", 128|8|512|1|2|16: "... and no B reads"}
for rep in range(2):
    for d, nm in names.items():
        _lib.set_option("corr_dbg", d)
        us = timeit(run, reps=30)
        print("%-40s %6.1f us" % (nm, us))
_lib.set_option("corr_dbg", 0)
