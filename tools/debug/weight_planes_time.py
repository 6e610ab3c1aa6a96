"""Times the per-step re-split of all weights into operand planes (unflow_weight_planes_batched through the engine) — one line."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd.core.engine import FlowNetCEngine, DEFAULT_PARAMS
dev = torch.device("cuda:0")
eng = FlowNetCEngine(4, 384, 512, params=dict(DEFAULT_PARAMS, flownet='C'), device=dev, seed=0)
fn = [getattr(eng, n) for n in ("refresh_weight_planes", "_refresh_weight_planes", "update_weight_planes") if hasattr(eng, n)][0]
for _ in range(5): fn(force=True)
torch.cuda.synchronize()
ts = []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(force=True); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort()
print(json.dumps({"weight_planes_us": round(ts[15] * 1e3, 1)}))
