"""Times the narrow-band forward correlation (81 channels, 16 x 96 x 128 x 256) — one JSON line; A/B with UNFLOW_OPT_CORR_NB=0."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd import _lib
from unflow_amd._lib import ptr, stream, check, planes_of
from unflow_amd.core import layers as L
dev = torch.device("cuda:0")
lib = _lib.lib()
N2, h2, w2 = 16, 96, 128
F2 = L.PT.alloc((N2, h2, w2, 256), dev, 3)
F2.t.copy_(torch.randn(N2, h2, w2, 256, device=dev))
L.planes_from_f32(F2.t, F2.pl)
co2 = torch.empty(N2, h2, w2, 84, device=dev)
st = stream()
def run():
    check(lib.unflow_correlation_nhwc_fwd_pl(ptr(F2.t), ptr(F2.t), 256, planes_of(F2.pl), planes_of(F2.pl), N2 // 2, ptr(co2), 84, N2, 256, h2, w2, 1, 4, 4, 1, 1, st))
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort()
print(json.dumps({"nb": os.environ.get("UNFLOW_OPT_CORR_NB", "1"), "us": round(ts[10] * 1e3, 1)}))
