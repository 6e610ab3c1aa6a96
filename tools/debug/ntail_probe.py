#!/usr/bin/env python3
"""What the 4-column N tail of the concat widths costs: the conv_transpose data / filter gradients of the decoder at Cin = 388 / 772 / 1028
(the concat widths: conv + deconv + 2 flow channels + 2 pad) against Cin = 384 / 768 / 1024 (the last 128-wide N tile gone)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unflow_amd.core import layers as L

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


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


def wplanes(w):
    import ctypes
    from unflow_amd import _lib
    from unflow_amd._lib import check, stream
    k, _, R, Cc = w.shape
    r8 = lambda c: (c + 7) // 8 * 8
    d = torch.zeros(3, k * k, R, r8(Cc), dtype=torch.int16, device=dev)
    tt = torch.zeros(3, k * k, Cc, r8(R), dtype=torch.int16, device=dev)
    check(_lib.lib().unflow_weight_planes_batched(1, (ctypes.c_void_p * 1)(w.data_ptr()), (ctypes.c_int * 1)(k * k), (ctypes.c_int * 1)(R),
                                                  (ctypes.c_int * 1)(Cc), (ctypes.c_void_p * 1)(d.data_ptr()), (ctypes.c_void_p * 1)(tt.data_ptr()), 3,
                                                  stream()))
    return d, tt


for (H, W, Cout, cins) in ((48, 64, 64, (388, 384)), (24, 32, 128, (772, 768)), (12, 16, 256, (1028, 1024))):
    for Cin in cins:
        x = L.PT.alloc((8, H, W, Cin), dev, 3)
        x.t.normal_()
        L.planes_from_f32(x.t, x.pl)
        dz = L.PT.alloc((8, 2 * H, 2 * W, Cout), dev, 3)
        dz.t.normal_()
        L.planes_from_f32(dz.t, dz.pl)
        w = torch.randn(4, 4, Cout, Cin, device=dev) * 0.02
        wd, wt = wplanes(w)
        dx = L.PT.alloc((8, H, W, Cin), dev, 3)
        dw = torch.empty_like(w)
        us_d = t(lambda: L.deconv_bwd_data(dz, w, wt, dx, False, x, 0, Cin))
        us_w = t(lambda: L.deconv_bwd_filter(x, dz, dw))
        us_f = t(lambda: L.deconv_fwd(x, w, wd, None, dz, True))
        print("deconv %4d -> %3d at %dx%d: data gradient %6.1f us   filter gradient %6.1f us   forward %6.1f us" % (Cin, Cout, H, W, us_d, us_w, us_f))
