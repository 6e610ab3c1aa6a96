"""Times the step's backward correlation (441 channels, 8 x 48 x 64 x 256, fused g0 + g1) and the 81-channel point — one JSON
line; A/B with UNFLOW_OPT_CORR_BWD_B128=0."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd import _lib
from unflow_amd._lib import ptr, stream, check, planes_of
from unflow_amd.core import layers as L
dev = torch.device("cuda:0")
lib = _lib.lib()
st = stream()
def med(run):
    for _ in range(10): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    return round(ts[15] * 1e3, 1)
def case(N, h, w, md, s2, oc_ld):
    F = L.PT.alloc((N, h, w, 256), dev, 3)
    F.t.copy_(torch.randn(N, h, w, 256, device=dev))
    L.planes_from_f32(F.t, F.pl)
    gco = torch.randn(N, h, w, oc_ld, device=dev)
    gf = torch.empty_like(F.t)
    return med(lambda: check(lib.unflow_correlation_nhwc_bwd_pl(ptr(gco), oc_ld, ptr(F.t), ptr(F.t), 256, planes_of(F.pl), planes_of(F.pl), N // 2, ptr(gf), ptr(None), 256, 1, N, 256, h, w, 1, md, md, 1, s2, st)))
print(json.dumps({"b128": os.environ.get("UNFLOW_OPT_CORR_BWD_B128", "1"), "bwd441_us": case(8, 48, 64, 20, 2, 441), "bwd81_us": case(16, 96, 128, 4, 1, 84)}))
