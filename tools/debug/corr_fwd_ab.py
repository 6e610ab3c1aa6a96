"""Same-box A/B of the step's forward correlation (8 x 48 x 64 x 256 -> 441 channels, NaN-prefilled output): option corr_rw 0 (corr_fwd_wb_kernel)
against 1 (corr_fwd_rw_kernel), three interleaved rounds, median of 40 launches each; prints the largest difference of the two results."""
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
outs = {}
for rep in range(3):
    for rw in (0, 1):
        _lib.set_option("corr_rw", rw)
        co = torch.full((8, 48, 64, 441), float('nan'), device=dev)
        us = timeit(lambda: check(lib.unflow_correlation_nhwc_fwd_pl(ptr(f), ptr(f), 256, planes_of(pt.pl), planes_of(pt.pl), 4, ptr(co), 441, 8, 256, 48, 64, 1, 20, 20, 1, 2, st)), reps=40)
        outs[rw] = co
        print(json.dumps({"corr_rw": rw, "us": round(us, 1)}))
d = (outs[0] - outs[1]).abs().max().item()
print("max |wb - rw| =", d, " nan:", torch.isnan(outs[1]).any().item(), " scale:", outs[0].abs().max().item())
