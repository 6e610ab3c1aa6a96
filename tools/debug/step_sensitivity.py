"""How much of the gradient moves when ONE conv3 weight moves by 3.6e-6 (FlowNetC, 128 x 192, eager, both forward correlation kernels):
the evidence that the loss is discontinuous (hard masks) behind profiles/r04_two_rank_test_sensitivity.txt."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd import _lib
from unflow_amd.core.engine import FlowNetEngine, DEFAULT_PARAMS
dev = torch.device("cuda:0")
H, W = 128, 192
g = torch.Generator().manual_seed(11)
a = (torch.rand(1, H, W, 3, generator=g) * 255).to(dev); b = (torch.rand(1, H, W, 3, generator=g) * 255).to(dev)
def grads(eng):
    eng.set_input(a, b); eng.G.zero_(); eng.forward_net(); eng.forward_loss(with_grad=True); eng.backward_net(); torch.cuda.synchronize()
    return eng.G.clone()
for rw in (0, 1, 0, 1):
    _lib.set_option("corr_rw", rw)
    eng = FlowNetEngine(1, H, W, params=dict(DEFAULT_PARAMS, flownet='C'), device=dev, seed=7)
    g0 = grads(eng)
    g0b = grads(eng)
    l = eng.by_name['conv3']
    lo = (l.w.data_ptr() - eng.P.data_ptr()) // 4
    eng.P[lo + 12345] += 3.6e-6
    eng._wplanes_version = None
    eng.refresh_weight_planes(force=True)
    g1 = grads(eng)
    d = (g1 - g0).abs()
    sc = g0.abs().max().item()
    lc = eng.by_name['conv3']; glo = (lc.dw.data_ptr() - eng.G.data_ptr()) // 4
    dc = d[glo:glo + lc.dw.numel()]; gc = g0[glo:glo + lc.dw.numel()].abs()
    print("rw", rw, "repeat identical:", torch.equal(g0, g0b), " max|dG|/max|G| =", d.max().item() / sc, " conv3: n rel change > 5%%: %d, > 1%%: %d" % ((dc > 0.05 * gc).sum().item(), (dc > 0.01 * gc).sum().item()),
          " all: n > 1e-6*max: %d" % (d > 1e-6 * sc).sum().item())
