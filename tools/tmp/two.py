import os, sys, subprocess, socket, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
child = r'''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from unflow_amd.core.engine import FlowNetEngine, DEFAULT_PARAMS
from unflow_amd.core.train import StepRunner
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
H, W = 128, 192
NS = int(os.environ.get("NSTEPS", "1")); UG = os.environ.get("UG", "1") == "1"
g = torch.Generator().manual_seed(11)
steps = [((torch.rand(2, H, W, 3, generator=g) * 255), (torch.rand(2, H, W, 3, generator=g) * 255)) for _ in range(NS)]
eng = FlowNetEngine(1, H, W, params=dict(DEFAULT_PARAMS, flownet='C'), device=dev, seed=7)
run = StepRunner(eng, world, use_graph=UG)
for a, b in steps:
    run.step(a[rank:rank + 1].to(dev), b[rank:rank + 1].to(dev), 1e-4)
torch.cuda.synchronize()
dist.barrier()
if rank == 0:
    one = FlowNetEngine(2, H, W, params=dict(DEFAULT_PARAMS, flownet='C'), device=dev, seed=7)
    r1 = StepRunner(one, 1, use_graph=UG)
    for a, b in steps: r1.step(a.to(dev), b.to(dev), 1e-4)
    torch.cuda.synchronize()
    d = (one.P - eng.P).abs()
    gd = (one.G - eng.G).abs()
    bad = []
    for l in one.layers:
        lo = (l.dw.data_ptr() - one.G.data_ptr()) // 4
        c = (d[lo:lo + l.dw.numel()] > 2e-6).sum().item()
        if c: bad.append((l.name.split('/')[-1], c, l.dw.numel()))
    nb = (d[one.n_weights:] > 2e-6).sum().item()
    print("RES layers", bad, "biases", nb)
    print("RES steps", NS, "graph", UG, "params > 2e-6:", (d > 2e-6).sum().item(), "max", d.max().item(), " | G diff max", gd.max().item(), "rel", gd.max().item() / one.G.abs().max().item(), "n G rel>1e-5:", (gd > 1e-5 * one.G.abs().max()).sum().item())
dist.barrier(); dist.destroy_process_group()
''' % ROOT
open("/tmp/child.py", "w").write(child)
for rw, ns, ug in ((1, 1, 1), (0, 1, 1)):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), UNFLOW_OPT_CORR_RW=str(rw), NSTEPS=str(ns), UG=str(ug))
        procs.append(subprocess.Popen([sys.executable, "/tmp/child.py"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    print("rw", rw, [l for l in outs[0][0].splitlines() if l.startswith("RES")], outs[0][1][-500:] if procs[0].returncode else "")
