"""Runs the four correlation kernels of the planes path a few times each (for rocprofv3 PMC passes: FETCH_SIZE / WRITE_SIZE
per launch): the step's 441-channel shape (8 x 48 x 64 x 256) and the 81-channel point (16 x 96 x 128 x 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd import _lib
from unflow_amd._lib import ptr, stream, check, planes_of
from unflow_amd.core import layers as L
dev = torch.device("cuda:0")
lib = _lib.lib()
st = stream()
for (N, h, w, md, s2, ld) in ((8, 48, 64, 20, 2, 441), (16, 96, 128, 4, 1, 84)):
    F = L.PT.alloc((N, h, w, 256), dev, 3)
    F.t.copy_(torch.randn(N, h, w, 256, device=dev))
    L.planes_from_f32(F.t, F.pl)
    co = torch.empty(N, h, w, ld, device=dev)
    gco = torch.randn(N, h, w, ld, device=dev)
    gf = torch.empty_like(F.t)
    for _ in range(4):
        check(lib.unflow_correlation_nhwc_fwd_pl(ptr(F.t), ptr(F.t), 256, planes_of(F.pl), planes_of(F.pl), N // 2, ptr(co), ld, N, 256, h, w, 1, md, md, 1, s2, st))
        check(lib.unflow_correlation_nhwc_bwd_pl(ptr(gco), ld, ptr(F.t), ptr(F.t), 256, planes_of(F.pl), planes_of(F.pl), N // 2, ptr(gf), ptr(None), 256, 1, N, 256, h, w, 1, md, md, 1, s2, st))
    torch.cuda.synchronize()
