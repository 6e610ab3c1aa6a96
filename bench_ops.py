#!/usr/bin/env python3
"""Op-level roofline microbenchmarks: HBM-bound ops at full-resolution shapes, where the bandwidth
fraction is meaningful (inside the training step they run on <= 96x128 pyramid levels and are launch-latency bound).
Prints one JSON object per op: algorithmic bytes (SURVEY 8d per-pixel figures), time (HIP events on the launch
stream, median of 20), GB/s and the fraction of the 8 TB/s HBM3E peak (6.3 TB/s is the measured streaming ceiling)."""
import ctypes
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

from unflow_amd import _lib, ops
from unflow_amd._lib import ptr, cf, cl, stream, check
from unflow_amd.core.image_warp import image_warp

HBM_PEAK = 8000.0


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2] * 1e3   # us


# `python bench_ops.py --driver`: the short subset bench.py embeds in the driver's line (secondary["ops"]: the warps, downsample,
# Adam, both correlation points; <= 3 s of GPU time); otherwise an optional substring filter on the row names
DRIVER = "--driver" in sys.argv[1:]
_args = [a for a in sys.argv[1:] if not a.startswith("--")]
ONLY = _args[0] if _args else ""      # print only the rows whose name contains this (every row is still measured)


def report(name, nbytes, us, extra=None):
    if ONLY and ONLY not in name:
        return
    gbs = nbytes / us / 1e3
    out = {"op": name, "algorithmic_MB": round(nbytes / 1e6, 1), "us": round(us, 1), "GB/s": round(gbs, 1),
           "frac_of_8TBs": round(gbs / HBM_PEAK, 3)}
    if extra:
        out.update(extra)
    print(json.dumps(out), flush=True)


def main():
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    g = torch.Generator().manual_seed(0)
    N, H, W = 16, 768, 1024
    npx = N * H * W
    im = torch.rand(N, H, W, 3, generator=g).to(dev)
    flow_iid = (torch.randn(N, H, W, 2, generator=g) * 4).to(dev)     # i.i.d. per pixel: worst case for the gathers
    # smooth field of the same magnitude (what a flow network produces): coarse noise, bilinearly upsampled
    flow = torch.nn.functional.interpolate(torch.randn(N, 2, H // 32, W // 32, generator=g) * 4, size=(H, W), mode='bilinear',
                                           align_corners=False).permute(0, 2, 3, 1).contiguous().to(dev)
    out3 = torch.empty_like(im)
    st = stream()
    # image_warp forward: (2C+2)*4 B/px  (read image once, flow, write warped)
    report("image_warp_fwd C=3 %dx%dx%d" % (N, H, W), npx * 32,
           timeit(lambda: check(lib.unflow_image_warp_fwd(ptr(im), 3, ptr(flow), cf(1.0), ptr(out3), ptr(None), 0, N, H, W, 3, st))))
    if not DRIVER:
        report("image_warp_fwd C=3, i.i.d. N(0,4^2) flow (gather worst case)", npx * 32,
               timeit(lambda: check(lib.unflow_image_warp_fwd(ptr(im), 3, ptr(flow_iid), cf(1.0), ptr(out3), ptr(None), 0, N, H, W, 3, st))))
    report("backward_warp_fwd C=3", npx * 32,
           timeit(lambda: check(lib.unflow_backward_warp_fwd(ptr(im), ptr(flow), ptr(out3), N, H, W, 3, st))))
    # the same op on a locally constant field (a translation): the rate of the gather itself, without the extra cache lines a
    # spatially varying field makes every wave touch (the field above changes by ~8 px across a wave's 64 pixels)
    flow_c = torch.full_like(flow, 2.5)
    report("backward_warp_fwd C=3, constant flow (2.5, 2.5)", npx * 32,
           timeit(lambda: check(lib.unflow_backward_warp_fwd(ptr(im), ptr(flow_c), ptr(out3), N, H, W, 3, st))))
    del flow_c
    dfl = torch.empty_like(flow)
    gout = torch.rand(N, H, W, 3, generator=g).to(dev)
    # backward wrt flow: read dout (3), image (3 via taps), flow (2), write dflow (2): (3C+... ) = 40 B/px
    report("backward_warp_bwd C=3", npx * 40,
           timeit(lambda: check(lib.unflow_backward_warp_bwd(ptr(gout), ptr(im), ptr(flow), ptr(dfl), N, H, W, 3, st))))
    gray1 = torch.empty(N, H, W, device=dev)
    gray2 = torch.empty_like(gray1)
    report("warp_gray_fwd (warp + grayscale fused)", npx * (12 + 8 + 4),
           timeit(lambda: check(lib.unflow_warp_gray_fwd(ptr(im), 3, ptr(flow), cf(1.0), ptr(gray2), N // 2, N, H, W, st))))
    check(lib.unflow_rgb_to_gray255(ptr(im), 3, ptr(gray1), cl(npx), st))
    mask = torch.ones(1, H, W, device=dev)
    dist = torch.empty_like(gray1)
    acc = torch.zeros(1, device=dev)
    dgray = torch.empty_like(gray1)
    for D in (() if DRIVER else (1, 3)):
        report("ternary_fwd D=%d (census %dx%d)" % (D, 2 * D + 1, 2 * D + 1), npx * 12,
               timeit(lambda: check(lib.unflow_ternary_fwd(ptr(gray1), ptr(gray2), ptr(mask), 1, ptr(dist), ptr(acc), cf(1.0), cf(npx), D, N, H, W, st))))
        report("ternary_bwd D=%d" % D, npx * 16,
               timeit(lambda: check(lib.unflow_ternary_bwd(ptr(gray1), ptr(gray2), ptr(mask), 1, ptr(dist), ptr(dgray), cf(1.0), cf(npx), D, N, H, W, st))))
    if not DRIVER:
        report("second_order fwd+bwd", npx * 16,
               timeit(lambda: check(lib.unflow_second_order_fwd_bwd(ptr(flow), cf(1.0), ptr(acc), ptr(dfl), 0, cf(1.0), cf(npx), N, H, W, st))))
    for s in (2, 4):
        o = torch.empty(N, H // s, W // s, 3, device=dev)
        report("downsample scale=%d C=3" % s, int(npx * 12 * (1 + 1.0 / (s * s))),
               timeit(lambda: check(lib.unflow_downsample_fwd(ptr(im), ptr(o), N, H, W, 3, s, st))))
    # forward_warp (ops/forward_warp_op.cu.cc:16-125): 12 B/px forward (flow in, splat sum out), 20 B/px backward; the work is
    # the <= 81 taps per source pixel, not the bytes — the HBM fraction is reported all the same (DESIGN.md has the tap-rate bound)
    fw_out = fw_ws = flow_50 = None
    if not DRIVER:
        fw_out = torch.empty(N, H, W, 1, device=dev)
        fw_ws = torch.empty(lib.unflow_forward_warp_workspace_bytes(N, H, W, 1) // 4 + 64, dtype=torch.float32, device=dev)   # 64-bit sums + the far-source bins
        flow_50 = (torch.rand(N, H, W, 2, generator=g) * 100 - 50).to(dev)
    for nm, fl in (() if DRIVER else (("smooth field, sigma 4 px", flow), ("i.i.d. N(0,4^2)", flow_iid), ("i.i.d. U(-50,50) px", flow_50))):
        for det in (1, 0):
            report("forward_warp_fwd %s, %s" % ("deterministic (64-bit fixed-point sums)" if det else "float atomics (reference contract)", nm),
                   npx * 12, timeit(lambda: check(lib.unflow_forward_warp_fwd(ptr(fl), ptr(fw_out), N, H, W, det, ptr(fw_ws),
                                                                              _lib.csz(fw_ws.numel() * 4), st)), reps=5))
        report("forward_warp_bwd %s" % nm, npx * 20,
               timeit(lambda: check(lib.unflow_forward_warp_bwd(ptr(gray1), ptr(fl), ptr(dfl), N, H, W, st)), reps=5))
    del fw_out, fw_ws, flow_50
    n = 39_200_000
    p, gr, m, v = (torch.randn(n, generator=g).to(dev) for _ in range(4))
    v.abs_()
    report("adam_step (39.2 M params, fused L2)", n * 4 * 7,
           timeit(lambda: check(lib.unflow_adam_step(ptr(p), ptr(gr), ptr(m), ptr(v), cl(n), cl(n), cf(1.0), cf(4e-4), cf(1e-4), cf(.9), cf(.999), cf(1e-8), st))))
    # correlation, FlowNetC configuration, N=8 directed samples (compute-leaning: report both rooflines).  The step's entry
    # points: features with their bf16 operand planes (written by the producing convolution in the step; here by
    # unflow_planes_from_f32 outside the timed region), six-term products on the bf16 matrix cores.
    from unflow_amd.core import layers as L
    from unflow_amd._lib import planes_of

    def with_planes(t):
        pt = L.PT.alloc(tuple(t.shape), dev, 3)
        pt.t.copy_(t)
        L.planes_from_f32(pt.t, pt.pl)
        return pt
    F = with_planes(torch.randn(8, 48, 64, 256, generator=g).to(dev))
    f = F.t
    co = torch.empty(8, 48, 64, 441, device=dev)
    us = timeit(lambda: check(lib.unflow_correlation_nhwc_fwd_pl(ptr(f), ptr(f), 256, planes_of(F.pl), planes_of(F.pl), 4, ptr(co), 441, 8, 256, 48, 64, 1, 20, 20, 1, 2, st)))
    report("correlation_nhwc_fwd_pl 441ch N=8", 8 * 11.71e6, us, {"GFLOP_algorithmic": 5.55, "TFLOP/s": round(5.55e3 / us, 1)})
    gco = torch.randn(8, 48, 64, 441, generator=g).to(dev)
    gf = torch.empty_like(f)
    us = timeit(lambda: check(lib.unflow_correlation_nhwc_bwd_pl(ptr(gco), 441, ptr(f), ptr(f), 256, planes_of(F.pl), planes_of(F.pl), 4, ptr(gf), ptr(None), 256, 1, 8, 256, 48, 64, 1, 20, 20, 1, 2, st)))
    report("correlation_nhwc_bwd_pl 441ch N=8 (fused g0+g1)", 8 * 18.0e6, us, {"GFLOP_algorithmic": 11.1, "TFLOP/s": round(11.1e3 / us, 1)})
    # the north star's +-4-displacement 81-channel cost volume (HBM-bound: AI = 17.5 FLOP/B), 1/8 resolution of 768x1024
    N2, h2, w2 = 16, 96, 128
    F2 = with_planes(torch.randn(N2, h2, w2, 256, generator=g).to(dev))
    f2 = F2.t
    co2 = torch.empty(N2, h2, w2, 84, device=dev)
    nb = N2 * h2 * w2 * (2 * 256 + 81) * 4
    gfl = 2 * 81 * 256 * N2 * h2 * w2 / 1e9
    if not DRIVER:
        us = timeit(lambda: check(lib.unflow_correlation_nhwc_fwd(ptr(f2), ptr(f2), 256, N2 // 2, ptr(co2), 84, N2, 256, h2, w2, 1, 4, 4, 1, 1, st)))
        report("correlation_nhwc_fwd 81ch (md=4, stride_2=1) N=16 96x128, fp32 entry", nb, us, {"GFLOP_algorithmic": round(gfl, 2), "TFLOP/s": round(gfl * 1e3 / us, 1)})
    # planes entry: the ALGORITHMIC bytes are SURVEY 8(d)'s fp32 figure, (2 * 256 + 81) * 4 B per pixel, whatever format the kernel
    # reads; the bytes of the format it does read (3 bf16 planes: 6 B per feature value) are reported beside it
    nbp = N2 * h2 * w2 * (2 * 256 * 6 + 81 * 4)
    us = timeit(lambda: check(lib.unflow_correlation_nhwc_fwd_pl(ptr(f2), ptr(f2), 256, planes_of(F2.pl), planes_of(F2.pl), N2 // 2, ptr(co2), 84, N2, 256, h2, w2, 1, 4, 4, 1, 1, st)))
    report("correlation_nhwc_fwd_pl 81ch (md=4, stride_2=1) N=16 96x128", nb, us,
           {"GFLOP_algorithmic": round(gfl, 2), "TFLOP/s": round(gfl * 1e3 / us, 1), "plane_format_MB": round(nbp / 1e6, 1),
            "frac_of_8TBs_on_plane_bytes": round(nbp / us / 1e3 / HBM_PEAK, 3)})
    gco2 = torch.randn(N2, h2, w2, 84, generator=g).to(dev)
    gf2 = torch.empty_like(f2)
    nbb = N2 * h2 * w2 * (81 + 2 * 256 + 256) * 4      # dOut + both features read, the (fused, shared-tensor) gradient written once
    if not DRIVER:
        us = timeit(lambda: check(lib.unflow_correlation_nhwc_bwd(ptr(gco2), 84, ptr(f2), ptr(f2), 256, N2 // 2, ptr(gf2), ptr(None), 256, 1, N2, 256, h2, w2, 1, 4, 4, 1, 1, st)))
        report("correlation_nhwc_bwd 81ch N=16 96x128 (fused g0+g1), fp32 entry", nbb, us, {"GFLOP_algorithmic": round(2 * gfl, 2), "TFLOP/s": round(2 * gfl * 1e3 / us, 1)})
    nbbp = N2 * h2 * w2 * (81 * 4 + 2 * 256 * 6 + 256 * 4)
    us = timeit(lambda: check(lib.unflow_correlation_nhwc_bwd_pl(ptr(gco2), 84, ptr(f2), ptr(f2), 256, planes_of(F2.pl), planes_of(F2.pl), N2 // 2, ptr(gf2), ptr(None), 256, 1, N2, 256, h2, w2, 1, 4, 4, 1, 1, st)))
    report("correlation_nhwc_bwd_pl 81ch N=16 96x128 (fused g0+g1)", nbb, us,
           {"GFLOP_algorithmic": round(2 * gfl, 2), "TFLOP/s": round(2 * gfl * 1e3 / us, 1), "plane_format_MB": round(nbbp / 1e6, 1),
            "frac_of_8TBs_on_plane_bytes": round(nbbp / us / 1e3 / HBM_PEAK, 3)})
    if DRIVER:
        return
    # the reference op's own boundary: two NCHW fp32 tensors in, NCHW out (ops/correlation_op.cc) — transposes to NHWC, operand
    # planes built in the workspace, the planes kernels, transpose back.  With only the fp32 part of the workspace: the fp32 kernels.
    del F2, f2, co2, gco2, gf2
    for (nm, Bc, Cc, Hc, Wc, md, s2, oc) in (("81ch (md=4, stride_2=1) B=8 96x128", 8, 256, 96, 128, 4, 1, 81), ("441ch (md=20, stride_2=2) B=4 48x64", 4, 256, 48, 64, 20, 2, 441)):
        xa = torch.randn(Bc, Cc, Hc, Wc, generator=g).to(dev)
        xb = torch.randn(Bc, Cc, Hc, Wc, generator=g).to(dev)
        oo = torch.empty(Bc, oc, Hc, Wc, device=dev)
        lib.unflow_correlation_workspace_bytes.restype = ctypes.c_size_t
        wsb = lib.unflow_correlation_workspace_bytes(Bc, Cc, Hc, Wc, 1, md, md, 1, s2)
        ws = torch.empty(wsb // 4 + 64, device=dev)
        fp32_only = (4 * xa.numel() + oo.numel()) * 4
        nbr = (2 * xa.numel() + oo.numel()) * 4
        for label, nbytes in (("planes in the workspace", wsb), ("fp32 part of the workspace only", fp32_only)):
            us = timeit(lambda: check(lib.unflow_correlation_fwd(ptr(xa), ptr(xb), ptr(oo), Bc, Cc, Hc, Wc, 1, md, md, 1, s2, ptr(ws), ctypes.c_size_t(nbytes), st)))
            report("correlation_fwd NCHW fp32 tensors (reference op boundary) %s, %s" % (nm, label), nbr, us)
        go = torch.randn(Bc, oc, Hc, Wc, generator=g).to(dev)
        ga, gb = torch.empty_like(xa), torch.empty_like(xb)
        for label, nbytes in (("planes in the workspace", wsb), ("fp32 part of the workspace only", fp32_only)):
            us = timeit(lambda: check(lib.unflow_correlation_bwd(ptr(go), ptr(xa), ptr(xb), ptr(ga), ptr(gb), Bc, Cc, Hc, Wc, 1, md, md, 1, s2, ptr(ws), ctypes.c_size_t(nbytes), st)))
            report("correlation_bwd NCHW fp32 tensors (reference op boundary) %s, %s" % (nm, label), nbr + 2 * xa.numel() * 4, us)
        del xa, xb, oo, ws, go, ga, gb


if __name__ == "__main__":
    main()
