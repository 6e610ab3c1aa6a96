import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd import _lib
from unflow_amd._lib import ptr, stream, check, planes_of
from unflow_amd.core import layers as L
dev = torch.device("cuda:0")
lib = _lib.lib()
st = stream()
tag = sys.argv[1]
g = torch.Generator().manual_seed(int(tag) + 3)
for (N, h, w) in ((2, 16, 24), (8, 48, 64)):
    feat = torch.randn(N, h, w, 256, generator=g).to(dev)
    pt = L.PT.alloc((N, h, w, 256), dev, 3)
    pt.t.copy_(feat)
    L.planes_from_f32(pt.t, pt.pl)
    for rw in (1, 0):
        _lib.set_option("corr_rw", rw)
        ref = None; bad = 0; worst = 0.0
        t0 = time.time()
        for it in range(300):
            co = torch.full((N, h, w, 476), float('nan'), device=dev)
            check(lib.unflow_correlation_nhwc_fwd_pl(ptr(pt.t), ptr(pt.t), 256, planes_of(pt.pl), planes_of(pt.pl), N // 2, ptr(co[..., 32:473]), 476, N, 256, h, w, 1, 20, 20, 1, 2, st))
            out = co[..., 32:473]
            if ref is None: ref = out.clone()
            elif not torch.equal(out, ref):
                bad += 1; worst = max(worst, (out - ref).abs().max().item())
        torch.cuda.synchronize()
        print("proc", tag, (N, h, w), "rw", rw, "mismatching runs:", bad, "of 299, worst", worst, "nan in ref:", torch.isnan(ref).any().item(), "%.1fs" % (time.time() - t0), flush=True)
