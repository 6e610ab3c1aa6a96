#!/usr/bin/env python3
"""Every conv / conv_transpose launch of one training step timed on its own (HIP events, REPS back-to-back
launches of the same call on the current stream): layer, pass, shapes, us, algorithmic TFLOP/s.

    python tools/per_layer_bench.py [--batch 4 --height 384 --width 512 --flownet C --dtype f32] > profiles/rNN_per_layer.txt

The calls are recorded from FlowNetEngine.fwd_bwd() exactly as bench.py's roofline leg records them, so tile choice,
split-K plan and epilogue fusions are the ones of the benchmarked step; a call re-run in isolation finds its weights in
L2, which the step does not — read the table as an upper bound per layer, the class total of bench.py as the truth."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--flownet", default="C")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--filter", default="", help="only time calls whose 'pass in out' line contains one of these |-separated substrings")
    args = ap.parse_args()
    if args.dtype == "f16":
        os.environ["UNFLOW_CONV_MATH"] = "f16"
    import torch
    from unflow_amd.core import layers as L
    from unflow_amd.core.engine import FlowNetCEngine, DEFAULT_PARAMS
    dev = torch.device("cuda", 0)
    eng = FlowNetCEngine(args.batch, args.height, args.width, params=dict(DEFAULT_PARAMS, flownet=args.flownet), device=dev, seed=0)
    g = torch.Generator().manual_seed(1234)
    im1 = (torch.rand(args.batch, args.height, args.width, 3, generator=g) * 255).to(dev)
    im2 = (torch.rand(args.batch, args.height, args.width, 3, generator=g) * 255).to(dev)
    eng.set_input(im1, im2)
    eng.fwd_bwd()
    torch.cuda.synchronize()
    names = ["conv_fwd", "conv_bwd_data", "conv_bwd_filter", "deconv_fwd", "deconv_bwd_data", "deconv_bwd_filter",
             "flow_wgrad_batched"]
    orig = {n: getattr(L, n) for n in names}
    calls = []

    def wrap(n, fn):
        def inner(*a, **k):
            calls.append((n, fn, a, k))
            return fn(*a, **k)
        return inner

    try:
        for n in names:
            setattr(L, n, wrap(n, orig[n]))
        eng.fwd_bwd()
        torch.cuda.synchronize()
    finally:
        for n in names:
            setattr(L, n, orig[n])

    def shape(x):
        t = x.t if isinstance(x, L.PT) else x
        return tuple(t.shape)

    tot_us = tot_gf = 0.0
    print("%-18s %-22s %-22s %5s %9s %8s" % ("pass", "in [B,H,W,C]", "out [B,H,W,C]", "k", "us", "TFLOP/s"))
    for n, fn, a, k in calls:
        if n == "flow_wgrad_batched":       # all Cout = 2 filter gradients of the decoder: one batch
            jobs = a[0]
            gf = sum(2.0 * shape(j[2])[0] * shape(j[2])[1] * shape(j[2])[2] * (9 if j[0] == 'conv' else 4) * shape(j[1])[3] * 2
                     for j in jobs) / 1e9
            for _ in range(3):
                fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn(*a, **k)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.reps
            tot_us += us
            tot_gf += gf
            print("%-18s %-22s %-22s %5d %9.1f %8.1f" % ("flow_wgrad_batch", "%d layers" % len(jobs), "Cout=2", 0, us, gf / us * 1e3))
            continue
        if n in ("conv_fwd", "deconv_fwd"):
            si, so = shape(a[0]), shape(a[4])
            w = a[1]
        elif n in ("conv_bwd_data", "deconv_bwd_data"):
            so, si = shape(a[0]), shape(a[3])       # dz = layer output, dx = layer input
            w = a[1]
        else:
            si, so = shape(a[0]), shape(a[1])
            w = a[2]
        kk = w.shape[0]
        if args.filter and not any(f in "%s %s %s" % (n, "x".join(map(str, si)), "x".join(map(str, so))) for f in args.filter.split("|")):
            continue
        if n.startswith("conv"):
            gf = 2.0 * so[0] * so[1] * so[2] * kk * kk * si[3] * so[3] / 1e9
        else:
            gf = 2.0 * so[0] * so[1] * so[2] * 4 * si[3] * so[3] / 1e9
        for _ in range(3):
            fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        tot_us += us
        tot_gf += gf
        print("%-18s %-22s %-22s %5d %9.1f %8.1f" % (n, "x".join(map(str, si)), "x".join(map(str, so)), kk, us, gf / us * 1e3 if us else 0))
    print("total %.1f us  %.1f GFLOP -> %.1f TFLOP/s (%d calls)" % (tot_us, tot_gf, tot_gf / tot_us * 1e3, len(calls)))


if __name__ == "__main__":
    main()
