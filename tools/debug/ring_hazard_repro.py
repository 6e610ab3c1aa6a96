#!/usr/bin/env python3
"""The 81-channel +-4 forward (corr_fwd_ring_kernel) replayed under a bandwidth hog on a second stream: how many output elements differ from
the quiet result, per replay (tests/test_fullsize_gpu.py::test_ring_correlation_bit_identical_under_a_bandwidth_hog asserts zero).
UNFLOW_LIB_PATH=<round-5 library> reproduces ADVICE r5's hazard: fragment registers copied before their inline-asm loads landed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd import _lib
from unflow_amd._lib import check, ptr, planes_of, stream
from unflow_amd.core import layers as L

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(11)
N, h, w, C = 16, 96, 128, 256
F = L.PT.alloc((N, h, w, C), dev, 3)
F.t.copy_(torch.randn(N, h, w, C, generator=g).to(dev))
L.planes_from_f32(F.t, F.pl)
lib = _lib.lib()


def corr(out):
    check(lib.unflow_correlation_nhwc_fwd_pl(ptr(F.t), ptr(F.t), C, planes_of(F.pl), planes_of(F.pl), N // 2, ptr(out), 84, N, C, h, w, 1, 4, 4, 1, 1,
                                             stream()), "correlation")


quiet = torch.zeros(N, h, w, 84, device=dev)
corr(quiet)
torch.cuda.synchronize()
again = torch.zeros_like(quiet)
corr(again)
print("library:", _lib.LIB_PATH)
print("quiet replay differs in %d elements" % (again != quiet).sum().item())
n = 150_000_000
src, dst = torch.randn(n, device=dev), torch.empty(n, device=dev)
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for _ in range(40):
        dst.copy_(src)
bad = []
for _ in range(20):
    out = torch.zeros_like(quiet)
    corr(out)
    d = out != quiet
    bad.append((int(d.sum().item()), float((out - quiet).abs().max().item())))
torch.cuda.synchronize()
print("under the hog, 20 replays: elements that differ / max |difference| per replay:")
print("  " + "  ".join("%d/%.3g" % b for b in bad))
print("value scale: max |out| = %.3f" % quiet.abs().max().item())
