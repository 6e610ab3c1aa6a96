"""The forward correlation kernels run to run and across batch compositions: a sample's output must not depend on the batch it sits in
(bit-identical), with either kernel (option corr_rw)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd import _lib
from unflow_amd._lib import ptr, stream, check, planes_of
from unflow_amd.core import layers as L
dev = torch.device("cuda:0")
lib = _lib.lib()
st = stream()
def run(feat, shift, rw):
    N, h, w, C = feat.shape
    _lib.set_option("corr_rw", rw)
    pt = L.PT.alloc((N, h, w, C), dev, 3)
    pt.t.copy_(feat)
    L.planes_from_f32(pt.t, pt.pl)
    co = torch.full((N, h, w, 476), float('nan'), device=dev)
    check(lib.unflow_correlation_nhwc_fwd_pl(ptr(pt.t), ptr(pt.t), C, planes_of(pt.pl), planes_of(pt.pl), shift, ptr(co[..., 32:473]), 476, N, C, h, w, 1, 20, 20, 1, 2, st))
    torch.cuda.synchronize()
    return co[..., 32:473].clone()
g = torch.Generator().manual_seed(3)
for (h, w) in ((16, 24), (48, 64), (12, 16)):
    f2 = torch.randn(2, h, w, 256, generator=g).to(dev)            # one pair, both directions
    f4 = torch.cat([f2[:1], f2[:1] * 0.5 + 1, f2[1:], f2[1:] * 0.25 - 1], 0)   # two pairs: samples (0,2) and (1,3)
    for rw in (0, 1):
        a = run(f2, 1, rw); b = run(f2, 1, rw)
        c = run(f4, 2, rw)
        same_run = torch.equal(a, b)
        per_sample = torch.equal(c[0], a[0]) and torch.equal(c[2], a[1])
        print(h, w, "rw", rw, "run-to-run identical:", same_run, " sample in batch of 4 == in batch of 2:", per_sample, " nan:", torch.isnan(a).any().item(),
              " max diff:", (c[0] - a[0]).abs().max().item())
