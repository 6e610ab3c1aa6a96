#!/usr/bin/env python3
"""BASELINE configs[0]: FlowNetS forward only, batch 1, 2 x (3 x 384 x 512) random frames, on the CPU oracle
(oracle/model_ref.py — the restatement of the reference's TF1 graph; the reference itself cannot run here).
Plumbing figure for BASELINE.md §3: 2 warm-up + median of 5; prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
import torch
from oracle import model_ref as M

torch.set_num_threads(os.cpu_count())
P = M.init_params_spec('S', 0)
g = torch.Generator().manual_seed(1234)
im1 = torch.rand(1, 384, 512, 3, generator=g) * 255
im2 = torch.rand(1, 384, 512, 3, generator=g) * 255
ts = []
with torch.no_grad():
    for it in range(7):
        t0 = time.perf_counter()
        flows = M.flownet(P, im1, im2, 'S', backward_flow=False)
        ts.append(time.perf_counter() - t0)
t = sorted(ts[2:])[2]
print(json.dumps({"config": "FlowNetS forward only, B=1, 384x512 (BASELINE configs[0])", "image_pairs_per_s": round(1 / t, 3),
                  "ms": round(t * 1e3, 1), "threads": os.cpu_count(), "kind": "port (oracle/model_ref.py)"}))
