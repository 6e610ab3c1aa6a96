"""Same-box A/B of FlowNetC's first layer (8 x 384 x 512 x 4 -> 64, planes-only output): option conv1_direct 1 (csrc/conv_first.hip)
against 0 (the gather kernel's rgb4 form), three interleaved rounds, median of 40 launches each."""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd import _lib
from unflow_amd._lib import check, stream
from unflow_amd.core import layers as L
from bench_ops import timeit
dev = torch.device("cuda:0")
B, H, W, Cout = 8, 384, 512, 64
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, 4, generator=g); x[..., 3] = 0
w = (torch.randn(7, 7, 4, Cout, generator=g) / 12).to(dev).contiguous()
b = (torch.randn(Cout, generator=g) * 0.1).to(dev)
X = L.PT(x.to(dev), torch.zeros(3, B, H, W, 4, dtype=torch.int16, device=dev))
L.planes_from_f32(X.t, X.pl, C=4)
w_dir = torch.zeros(3, 7, 28, Cout, dtype=torch.int16, device=dev)
w_tr = torch.zeros(3, 7, Cout, 32, dtype=torch.int16, device=dev)
check(_lib.lib().unflow_weight_planes_batched(1, (ctypes.c_void_p * 1)(w.data_ptr()), (ctypes.c_int * 1)(7), (ctypes.c_int * 1)(28), (ctypes.c_int * 1)(Cout),
                                              (ctypes.c_void_p * 1)(w_dir.data_ptr()), (ctypes.c_void_p * 1)(w_tr.data_ptr()), 3, stream()), "weight_planes")
Y = L.PT.alloc((B, H // 2, W // 2, Cout), dev, 3)
outs = {}
for rep in range(3):
    for d in (0, 1):
        _lib.set_option("conv1_direct", d)
        us = timeit(lambda: L.conv_fwd(X, w, w_tr, b, Y, 2, True, planes_only=True), reps=40)
        outs[d] = Y.pl.clone()
        print(json.dumps({"conv1_direct": d, "us": round(us, 1)}))
print("planes identical:", torch.equal(outs[0], outs[1]), " differing elements:", (outs[0] != outs[1]).sum().item(), "of", outs[0].numel())
