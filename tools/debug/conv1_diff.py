"""Where conv_first.hip's output differs grossly from the gather kernel's (debugging aid): positions of |diff| > 1e-3."""
import sys, os, ctypes, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd import _lib
from unflow_amd._lib import check, stream
from unflow_amd.core import layers as L
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
def val(pl):
    return ((pl.to(torch.int32) & 0xffff) << 16).view(torch.float32).double().sum(0)
def run(d):
    _lib.set_option("conv1_direct", d)
    Y = L.PT.alloc((B, H // 2, W // 2, Cout), dev, 3)
    L.conv_fwd(X, w, w_tr, b, Y, 2, True, planes_only=True)
    torch.cuda.synchronize()
    return Y.pl.clone()
ref = val(run(0))
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    pl = run(1)
    d = (val(pl) - ref).abs()
    bad = (d > 1e-3).nonzero()
    print("run", rep, "bad elements:", bad.shape[0], "max", d.max().item())
    if bad.shape[0]:
        bb = bad.cpu()
        print("  batch", collections.Counter(bb[:, 0].tolist()).most_common(8))
        print("  y % 8", sorted(collections.Counter((bb[:, 1] % 8).tolist()).items()))
        print("  y // 8", collections.Counter((bb[:, 1] // 8).tolist()).most_common(6))
        print("  x % 32", sorted(collections.Counter((bb[:, 2] % 32).tolist()).items())[:40])
        print("  x // 32", sorted(collections.Counter((bb[:, 2] // 32).tolist()).items()))
        print("  c", sorted(collections.Counter(bb[:, 3].tolist()).items()))
        print("  first", bb[:6].tolist())
