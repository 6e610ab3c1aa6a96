"""Times the step's forward correlation (441 channels, 8 x 48 x 64 x 256) — one JSON line; A/B with UNFLOW_OPT_CORR_WB=0."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd import _lib
from unflow_amd._lib import ptr, stream, check, planes_of
from unflow_amd.core import layers as L
dev = torch.device("cuda:0")
lib = _lib.lib()
F = L.PT.alloc((8, 48, 64, 256), dev, 3)
F.t.copy_(torch.randn(8, 48, 64, 256, device=dev))
L.planes_from_f32(F.t, F.pl)
co = torch.empty(8, 48, 64, 441, device=dev)
st = stream()
def run():
    check(lib.unflow_correlation_nhwc_fwd_pl(ptr(F.t), ptr(F.t), 256, planes_of(F.pl), planes_of(F.pl), 4, ptr(co), 441, 8, 256, 48, 64, 1, 20, 20, 1, 2, st))
for _ in range(10): run()
torch.cuda.synchronize()
ts = []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort()
print(json.dumps({"wb": os.environ.get("UNFLOW_OPT_CORR_WB", "1"), "us": round(ts[15] * 1e3, 1)}))
