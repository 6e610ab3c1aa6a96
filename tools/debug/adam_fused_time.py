#!/usr/bin/env python3
"""Optimizer tail of the step, alone: adam_kernel + weight_planes_kernel (two launches) against adam_planes_kernel (one), FlowNetC B = 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd.core.engine import FlowNetCEngine

dev = torch.device("cuda:0")
eng = FlowNetCEngine(4, 384, 512, device=dev, seed=0)
eng.G.normal_(0, 1e-3)
eng.defer_l2 = True


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def two():
    eng.adam_range(0, eng.n_params, 1e-4)
    eng.refresh_weight_planes(force=True)


def one():
    eng.adam_ranges_fused([(0, eng.n_params)], 1e-4)


for _ in range(2):
    print("adam + weight_planes: %.1f us   fused adam_planes: %.1f us" % (t(two), t(one)))
