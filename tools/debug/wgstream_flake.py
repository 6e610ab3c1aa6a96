import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, '' + os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests") + '')
import torch
from unflow_amd.core.engine import FlowNetCEngine
from parity_util import images, graph_step
B=2
dev = torch.device('cuda:0')
eng = FlowNetCEngine(B, 384, 512, device=dev, seed=None)
eng.init_params(seed=5)
im1, im2 = images(B, 384, 512, 77)
im1, im2 = im1.to(dev), im2.to(dev)
side = torch.cuda.Stream(dev)
def run(group, graph):
    eng.wgrad_group, eng.wgrad_stream = group, (side if group > 0 else None)
    if graph:
        graph_step(eng, im1, im2)
    else:
        eng.set_input(im1, im2); eng.G.zero_()
        eng.forward_net(); eng.forward_loss(with_grad=True); eng.backward_net(); torch.cuda.synchronize()
    return eng.G.clone()
ref = run(0, False)
for rep in range(40):
    for group in (3, 4, 6):
        got = run(group, True)
        for l in eng.layers:
            lo = (l.dw.data_ptr() - eng.G.data_ptr()) // 4
            a, b = ref[lo:lo + l.dw.numel()], got[lo:lo + l.dw.numel()]
            if not torch.equal(a, b):
                idx = (a != b).nonzero().flatten()
                st = eng.stages[-1]
                op = [o for o in st.ops if o.kind == 'layer' and o.l is l][0]
                x = st.pt(op.src).t.detach().cpu().double()
                dz = st.pt(op.dst, True).t.detach().cpu().double()[..., :l.cout]
                if l.kind == 'deconv' and l.cout == 2:
                    N, H, W, _ = x.shape
                    dzp = torch.nn.functional.pad(dz, (0, 0, 1, 2, 1, 2))          # oy = 2 iy + ky - 1 in [-1, 2H + 1]
                    dw = torch.zeros(4, 4, l.cout, x.shape[-1], dtype=torch.float64)
                    for ky in range(4):
                        for kx in range(4):
                            g = dzp[:, ky:ky + 2 * H:2, kx:kx + 2 * W:2, :]          # [N,H,W,co] at (2iy+ky-1, 2ix+kx-1)
                            dw[ky, kx] = torch.einsum('nhwo,nhwi->oi', g, x)
                    j = idx[0].item()
                    tp, co, ci = j // 4, (j // 2) % 2, j % 2
                    ky, kx = tp // 4, tp % 4
                    g = dzp[:, ky:ky + 2 * H:2, kx:kx + 2 * W:2, co]
                    contrib = (g * x[..., ci]).flatten()
                    d = (b[j] - a[j]).item()
                    ratio = d / contrib
                    near = (ratio - ratio.round()).abs() < 2e-3
                    cand = [(int(s), round(ratio[s].item(), 4), contrib[s].item()) for s in near.nonzero().flatten().tolist() if abs(ratio[s].round().item()) in (1.0, 2.0)]
                    print("   idx %d = tap (%d,%d) co %d ci %d: got - ref = %.9f; sites whose contribution explains it (site, multiple, contribution): %s" % (j, ky, kx, co, ci, d, cand[:6]))
                    for s, m, c in cand[:2]:
                        n_, r_ = divmod(s, H * W); iy, ix = divmod(r_, W)
                        print("      site n=%d iy=%d ix=%d: x = %s, dz at its tap pixel = %s" % (n_, iy, ix, x[n_, iy, ix].tolist(), dzp[n_, 2 * iy + ky, 2 * ix + kx].tolist()))
                    print("   recomputed from the buffers after the replay: %.9f (ref %.9f, got %.9f)" % (dw.flatten()[j].item(), a[j].item(), b[j].item()))
                print("rep %d group %d %s: %d of %d differ, first idx %s, ref %s got %s, tensor offset %d (mod 32: %d)" % (
                    rep, group, l.name, idx.numel(), a.numel(), idx[:8].tolist(), a[idx[:4]].tolist(), b[idx[:4]].tolist(), lo, lo % 32), flush=True)
print("done")
