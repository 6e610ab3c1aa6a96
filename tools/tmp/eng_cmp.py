import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd import _lib
from unflow_amd.core.engine import FlowNetEngine, DEFAULT_PARAMS
dev = torch.device("cuda:0")
H, W = 128, 192
g = torch.Generator().manual_seed(11)
a = (torch.rand(2, H, W, 3, generator=g) * 255).to(dev); b = (torch.rand(2, H, W, 3, generator=g) * 255).to(dev)
def grads(B, ia, ib):
    eng = FlowNetEngine(B, H, W, params=dict(DEFAULT_PARAMS, flownet='C'), device=dev, seed=7)
    eng.set_input(ia, ib)
    eng.G.zero_()
    eng.forward_net()
    loss = eng.forward_loss(with_grad=True)
    eng.backward_net()
    torch.cuda.synchronize()
    return eng, eng.G.clone()
for rw in (0, 1, 0, 1):
    _lib.set_option("corr_rw", rw)
    e0, g0 = grads(1, a[:1], b[:1])
    e1, g1 = grads(1, a[1:], b[1:])
    e2, g2 = grads(2, a, b)
    gm = (g0 + g1) * 0.5
    d = (gm - g2).abs()
    sc = g2.abs().max().item()
    print("rw", rw, "max |mean(g0,g1) - g2| =", d.max().item(), " rel to max|g|:", d.max().item() / sc, " n > 1e-6*max:", (d > 1e-6 * sc).sum().item(), " first:", (d > 1e-6 * sc).nonzero().flatten()[:6].tolist())
    # layer names of the worst entries
    idx = d.argmax().item()
    for l in e2.layers if hasattr(e2, 'layers') else []:
        pass
