#!/usr/bin/env python3
"""Where a wave of the halo kernel spends a K tile (diagnostic build).  Build a traced copy of the library and run one layer:

    hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fno-vectorize -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden \\
          -DUNFLOW_PHASE_TRACE=<workgroup id> -c unflow_amd/csrc/conv_planes.hip -o scratch/conv_planes_trace.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libunflow_trace.so scratch/conv_planes_trace.o <the other .o files>
    UNFLOW_LIB_PATH=scratch/libunflow_trace.so python tools/phase_trace.py

The waves of that workgroup read s_memtime (shader cycles on gfx950) at the phase boundaries of every K tile and sum the
phases in registers: MFMA phase (fragment reads + 48 MFMAs + the next tile's loads issued), first barrier, waiting for the
next tile's loads, LDS stores, second barrier, loop tail.  Prints per wave the mean cycles of each phase per tile."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from unflow_amd import _lib
    from unflow_amd.core import layers as L
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dev = torch.device("cuda:0")
    B, H, W, Cin, Cout, k = 8, 48, 64, 476, 256, 3          # conv3_1 forward of the benchmarked step
    P = 1 if os.environ.get("PHASE_TRACE_F16") else 3      # PHASE_TRACE_F16=1: the fp16 mode (8 MFMAs per wave and tile), B = 16 images
    if P == 1:
        B = 16
    g = torch.Generator().manual_seed(1)
    X = L.PT.alloc((B, H, W, Cin), dev, P)
    X.t.copy_(torch.randn(B, H, W, Cin, generator=g))
    L.planes_from_f32(X.t, X.pl)
    w = (torch.randn(k, k, Cin, Cout, generator=g) / (k * k * Cin) ** 0.5).to(dev)
    r8 = lambda c: (c + 7) // 8 * 8                        # noqa: E731
    d = torch.zeros(P, k * k, Cin, r8(Cout), dtype=torch.int16, device=dev)
    t = torch.zeros(P, k * k, Cout, r8(Cin), dtype=torch.int16, device=dev)
    _lib.check(_lib.lib().unflow_weight_planes_batched(1, (ctypes.c_void_p * 1)(w.data_ptr()), (ctypes.c_int * 1)(k * k),
                                                      (ctypes.c_int * 1)(Cin), (ctypes.c_int * 1)(Cout),
                                                      (ctypes.c_void_p * 1)(d.data_ptr()), (ctypes.c_void_p * 1)(t.data_ptr()), P,
                                                      _lib.stream()), "wp")
    Y = L.PT.alloc((B, H, W, Cout), dev, P)
    bias = torch.zeros(Cout, device=dev)
    DZ = L.PT.alloc((B, H, W, Cout), dev, P)
    DZ.t.copy_(torch.randn(B, H, W, Cout, generator=g))
    L.planes_from_f32(DZ.t, DZ.pl)
    dw = torch.zeros(k, k, Cin, Cout, device=dev)
    lib = _lib.lib()

    def trace(label, fn, names):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lib.unflow_debug_phase_trace(None, 1)
        fn()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (8 * 8))()
        _lib.check(lib.unflow_debug_phase_trace(buf, 0), "trace")
        a = np.frombuffer(buf, dtype=np.uint64).reshape(8, 8).astype(np.float64)
        print(label)
        for wv in range(8):
            n = a[wv, 6]
            if n < 2:
                continue
            ph = a[wv, :6] / np.array([n, n, n, n, n, n - 1])
            tot = ph.sum()
            print("  wave %d: %d tiles, %.0f cycles per tile: " % (wv, n, tot) +
                  ", ".join("%s %.0f (%.0f%%)" % (nm, v, 100 * v / tot) for nm, v in zip(names, ph) if nm))

    if os.environ.get("PHASE_TRACE_DEEP"):
        # conv6_1 forward (8 x 6 x 8 x 1024 -> 1024): the plain gather kernel, 18 K tiles per workgroup
        Bd, Hd, Wd, Cd = 8, 6, 8, 1024
        Xd = L.PT.alloc((Bd, Hd, Wd, Cd), dev, 3)
        Xd.t.copy_(torch.randn(Bd, Hd, Wd, Cd, generator=g))
        L.planes_from_f32(Xd.t, Xd.pl)
        wd = (torch.randn(3, 3, Cd, Cd, generator=g) / (9 * Cd) ** 0.5).to(dev)
        dd = torch.zeros(3, 9, Cd, Cd, dtype=torch.int16, device=dev)
        td = torch.zeros(3, 9, Cd, Cd, dtype=torch.int16, device=dev)
        _lib.check(_lib.lib().unflow_weight_planes_batched(1, (ctypes.c_void_p * 1)(wd.data_ptr()), (ctypes.c_int * 1)(9),
                                                          (ctypes.c_int * 1)(Cd), (ctypes.c_int * 1)(Cd),
                                                          (ctypes.c_void_p * 1)(dd.data_ptr()), (ctypes.c_void_p * 1)(td.data_ptr()), 3,
                                                          _lib.stream()), "wp")
        Yd = L.PT.alloc((Bd, Hd, Wd, Cd), dev, 3)
        bd = torch.zeros(Cd, device=dev)
        trace("gather kernel, conv6_1 forward (48 MFMAs per wave and tile):", lambda: L.conv_fwd(Xd, wd, td, bd, Yd, 1, True),
              ["mfma phase", "barrier 1", "wait loads", "lds stores", "barrier 2", "loop tail"])
        return
    if P == 1:
        trace("halo kernel, conv3_1 forward, fp16 (8 MFMAs per wave and tile):", lambda: L.conv_fwd(X, w, t, bias, Y, 1, True),
              ["mfma phase", "barrier 1", "wait loads", "lds stores", "barrier 2", "loop tail"])
        return
    trace("halo kernel, conv3_1 forward (48 MFMAs per wave and tile):", lambda: L.conv_fwd(X, w, t, bias, Y, 1, True),
          ["mfma phase", "barrier 1", "wait loads", "lds stores", "barrier 2", "loop tail"])
    if _lib.get_option("wgrad_pp") > 0:
        trace("ping-pong filter gradient, conv3_1 (2 x 24 MFMAs per wave and stage):", lambda: L.conv_bwd_filter(X, DZ, dw, 1),
              ["read slot 0", "barrier", "24 mfma", "barrier", "read slot 1 + barrier", "24 mfma + barrier"])
    else:
        trace("LDS-DMA filter gradient, conv3_1 (24 MFMAs per wave and stage):", lambda: L.conv_bwd_filter(X, DZ, dw, 1),
              ["issue dma", "24 mfma (+ read wait)", "wait dma", "barrier", "issue reads", "loop tail"])


if __name__ == "__main__":
    main()
