"""Bitwise comparison of all gradients: filter gradients inline vs on the second stream (eager and hipGraph)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
from unflow_amd.core.engine import FlowNetCEngine
from parity_util import images, graph_step

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda:0')
eng = FlowNetCEngine(B, 384, 512, device=dev, seed=None)
eng.init_params(seed=100 + B)
im1, im2 = images(B, 384, 512, 200 + B)
im1, im2 = im1.to(dev), im2.to(dev)
side = torch.cuda.Stream(dev)


def run(group, graph, unique=False, tiny=False):
    eng.wgrad_unique_ws, eng.wgrad_inline_tiny = unique, tiny
    eng.wgrad_group = group
    eng.wgrad_stream = side if group > 0 else None
    if graph:
        graph_step(eng, im1, im2)
    else:
        eng.set_input(im1, im2)
        eng.G.zero_()
        eng.forward_net(); eng.forward_loss(with_grad=True); eng.backward_net()
        torch.cuda.synchronize()
    return eng.G.clone()


ref = run(0, False)
def diff(g):
    bad = []
    for l in eng.layers:
        lo = (l.dw.data_ptr() - eng.G.data_ptr()) // 4
        n = l.dw.numel()
        a, b = ref[lo:lo + n], g[lo:lo + n]
        if not torch.equal(a, b):
            bad.append((l.name.split('/')[-1], ((a - b).abs().max() / a.abs().max()).item()))
    return bad


def dump(g):
    l = [l for l in eng.layers if l.name.endswith('flow3_up2')][0]
    lo = (l.dw.data_ptr() - eng.G.data_ptr()) // 4
    a, b = ref[lo:lo + 64].cpu(), g[lo:lo + 64].cpu()
    d = (b - a)
    print("   ref", [round(v, 3) for v in a.tolist()])
    print("   dif", [round(v, 4) for v in d.tolist()])


def chk(name, t):
    return "%s %.10e" % (name, t.double().sum().item())


st = eng.stages[-1]
for group, graph in [(100, False), (100, False), (100, True), (100, True), (0, True)]:
    g = run(group, graph)
    bad = diff(g)
    print("group %d graph %d: %d differing %s" % (group, graph, len(bad), bad[:8]), flush=True)
    print("   ", chk("cat2g[192:196]", st.grad['cat2'][..., 192:196]), chk("flow3", st.act['flow3']),
          chk("cat5g[1024:1028]", st.grad['cat5'][..., 1024:1028]), chk("flow6", st.act['flow6']))
    if bad:
        dump(g)
