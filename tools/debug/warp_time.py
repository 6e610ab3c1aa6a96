"""Times backward_warp / image_warp / warp_gray forward at 16 x 768 x 1024 x 3 (the bench_ops.py shapes) — one line each."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench_ops as bo
from unflow_amd import _lib
from unflow_amd._lib import ptr, cf, stream, check
dev = torch.device("cuda:0")
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
N, H, W = 16, 768, 1024
npx = N * H * W
im = torch.rand(N, H, W, 3, generator=g).to(dev)
flow = torch.nn.functional.interpolate(torch.randn(N, 2, H // 32, W // 32, generator=g) * 4, size=(H, W), mode='bilinear',
                                       align_corners=False).permute(0, 2, 3, 1).contiguous().to(dev)
out3 = torch.empty_like(im)
st = stream()
gray2 = torch.empty(N, H, W, device=dev)
dfl = torch.empty_like(flow)
gout = torch.rand(N, H, W, 3, generator=g).to(dev)
bo.report("backward_warp_fwd C=3", npx * 32, bo.timeit(lambda: check(lib.unflow_backward_warp_fwd(ptr(im), ptr(flow), ptr(out3), N, H, W, 3, st))))
bo.report("image_warp_fwd C=3", npx * 32, bo.timeit(lambda: check(lib.unflow_image_warp_fwd(ptr(im), 3, ptr(flow), cf(1.0), ptr(out3), ptr(None), 0, N, H, W, 3, st))))
bo.report("warp_gray_fwd", npx * 24, bo.timeit(lambda: check(lib.unflow_warp_gray_fwd(ptr(im), 3, ptr(flow), cf(1.0), ptr(gray2), N // 2, N, H, W, st))))
bo.report("backward_warp_bwd C=3", npx * 40, bo.timeit(lambda: check(lib.unflow_backward_warp_bwd(ptr(gout), ptr(im), ptr(flow), ptr(dfl), N, H, W, 3, st))))
zf = torch.zeros_like(flow)
bo.report("backward_warp_fwd C=3, zero flow", npx * 32, bo.timeit(lambda: check(lib.unflow_backward_warp_fwd(ptr(im), ptr(zf), ptr(out3), N, H, W, 3, st))))
cf_ = torch.full_like(flow, 2.5)
bo.report("backward_warp_fwd C=3, constant flow 2.5", npx * 32, bo.timeit(lambda: check(lib.unflow_backward_warp_fwd(ptr(im), ptr(cf_), ptr(out3), N, H, W, 3, st))))
