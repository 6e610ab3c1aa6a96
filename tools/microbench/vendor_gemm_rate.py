#!/usr/bin/env python3
"""Calibration only (not on the product path): what the vendor bf16 GEMM (hipBLASLt through torch.matmul) reaches on this
GPU at the implicit-GEMM shapes of the big FlowNetC layers (M = sites, N = Cout, K = taps x Cin) and at a large square —
the practical ceiling for one bf16 MFMA term on MI355X, to read the bf16x3 kernels' 416.7 TFLOP/s peak against."""
import torch


def rate(M, N, K, dtype=torch.bfloat16, iters=30):
    a = torch.randn(M, K, device="cuda", dtype=dtype)
    b = torch.randn(K, N, device="cuda", dtype=dtype)
    for _ in range(5):
        torch.matmul(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(a, b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * M * N * K / us / 1e6


if __name__ == "__main__":
    shapes = [("conv3_1 (B=8)", 24576, 256, 4284), ("conv2 (B=8)", 98304, 128, 1600), ("conv4_1 (B=8)", 6144, 512, 4608),
              ("conv6_1 (B=8)", 384, 1024, 9216), ("square 8192", 8192, 8192, 8192), ("square 4096", 4096, 4096, 4096)]
    print("%-16s %8s %6s %6s %10s %10s %8s" % ("shape", "M", "N", "K", "us", "TFLOP/s", "of 2500"))
    for name, M, N, K in shapes:
        us, tf = rate(M, N, K)
        print("%-16s %8d %6d %6d %10.1f %10.1f %7.1f%%" % (name, M, N, K, us, tf, 100 * tf / 2500))
