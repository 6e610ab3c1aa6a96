"""-m gpu: the operand-plane kernels of csrc/conv_planes.hip through the C ABI (unflow_*_pl) vs fp64 torch-CPU:
plane producers (exactness of the 3-way bf16 split, fp16 rounding, weight planes in both layouts), conv / conv_transpose
fwd, dgrad (stride 1 and the 4 parity classes of stride 2), wgrad (the ds_read_b64_tr_b16 path), every tile config, the
split-K reduce, odd channel counts with zero-padded plane tails, channel-slice views, the fused epilogues and the output
planes they write.

Tolerances: n_planes = 3 is fp32-equivalent (six exact bf16 products, fp32 accumulation): 2e-5 of the output scale like the
fp32-MFMA kernels (tests/test_conv_gpu.py).  n_planes = 1 (fp16 operands, fp32 accumulate): operands carry 2^-11 relative
rounding, a K-term dot product of zero-mean terms ~ 2^-11 * sqrt(2/K) * sum|a||b| / sqrt(K): stated bound 2e-3 of the
output scale."""
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def bf16_bits_to_f64(t):
    return (t.to(torch.int32) << 16).view(torch.float32).double()


def planes_value(pl):
    """fp64 value an int16 planes tensor [P, ...] encodes (3: hi + mid + lo bf16; 1: fp16)."""
    if pl.shape[0] == 1:
        return pl[0].view(torch.float16).double()
    return bf16_bits_to_f64(pl[0]) + bf16_bits_to_f64(pl[1]) + bf16_bits_to_f64(pl[2])


TOL = {3: 2e-5, 1: 2e-3}


def make_pt(x, dev, P, extra=0, offset=0):
    """PT of x (NHWC) living as a channel slice [offset, offset + C) of a wider zero buffer; planes filled by the
    library (unflow_planes_from_f32)."""
    from unflow_amd.core import layers as L
    B, H, W, C = x.shape
    buf = L.PT.alloc((B, H, W, offset + C + extra), dev, P)
    buf.t[..., offset:offset + C] = x.to(dev)
    pt = buf.sl(offset, offset + C) if (offset or extra) else buf
    L.planes_from_f32(pt.t, pt.pl)
    return pt


def weight_planes(w, dev, P):
    """(direct, transposed) planes of W[k,k,R,Cc] via the batched library call."""
    import ctypes
    from unflow_amd import _lib
    from unflow_amd._lib import check, stream
    k, _, R, Cc = w.shape
    r8 = lambda c: (c + 7) // 8 * 8                                         # noqa: E731
    d = torch.zeros(P, k * k, R, r8(Cc), dtype=torch.int16, device=dev)
    t = torch.zeros(P, k * k, Cc, r8(R), dtype=torch.int16, device=dev)
    wd = w.to(dev).contiguous()
    check(_lib.lib().unflow_weight_planes_batched(1, (ctypes.c_void_p * 1)(wd.data_ptr()), (ctypes.c_int * 1)(k * k),
                                                  (ctypes.c_int * 1)(R), (ctypes.c_int * 1)(Cc),
                                                  (ctypes.c_void_p * 1)(d.data_ptr()), (ctypes.c_void_p * 1)(t.data_ptr()), P,
                                                  stream()), "weight_planes")
    return wd, d, t


@pytest.mark.parametrize("P", [3, 1])
def test_plane_producers(P, dev):
    from unflow_amd.core import layers as L
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 5, 7, 13, generator=g) * (10.0 ** torch.randint(-6, 6, (2, 5, 7, 13), generator=g).float())
    pt = make_pt(x, dev, P, extra=3, offset=4)
    val = planes_value(pt.pl.cpu())
    if P == 3:
        assert torch.equal(val[..., :13], x.double())                      # hi + mid + lo == x EXACTLY
    else:
        assert torch.equal(val[..., :13].float(), x.half().float())        # round-to-nearest-even fp16
    assert torch.all(val[..., 13:16] == 0)                                   # zero tail up to the next multiple of 8
    # weights, both layouts
    w = torch.randn(3, 3, 10, 12, generator=g)
    wd, d, t = weight_planes(w, dev, P)
    vd, vt = planes_value(d.cpu()), planes_value(t.cpu())                    # [9,10,16], [9,12,16]
    ref = w.double().reshape(9, 10, 12)
    if P == 1:
        ref = ref.float().half().double()
    assert torch.equal(vd[..., :12], ref) and torch.all(vd[..., 12:] == 0)
    assert torch.equal(vt[..., :10], ref.transpose(1, 2)) and torch.all(vt[..., 10:] == 0)


# (B, H, W, Cin, Cout, k, stride)
CONV_CASES = [
    (2, 24, 32, 4, 64, 7, 2),      # conv1: 4 -> 8 plane channels, one tap per K granule, 128x64 tile
    (2, 16, 24, 64, 128, 5, 2),    # conv2
    (2, 48, 64, 128, 128, 3, 1),   # 6144 sites: 128x128 tiles, no split
    (1, 12, 16, 476, 256, 3, 1),   # conv3_1: 476 -> 480 plane channels (granules straddle taps? 60 per tap)
    (1, 12, 16, 388, 64, 3, 1),    # odd granule count per tap (49): K tiles straddle taps
    (2, 6, 8, 256, 512, 3, 2),     # conv4-like, asymmetric pad
    (8, 6, 8, 512, 1024, 3, 2),    # conv6: split-K + reduce epilogue
    (8, 6, 8, 1024, 1024, 3, 1),   # conv6_1: split-K
    (2, 12, 16, 256, 32, 1, 1),    # conv_redir: 64x64 tile (fwd), pointwise kernel (dgrad)
    (1, 10, 14, 40, 44, 3, 1),     # M and N tails
    (2, 12, 16, 96, 12, 1, 1),     # 3/8-width conv_redir: Cout 12 (K tail granule of the dgrad reads zeros)
    # halo kernel (source stride 1, maps >= 8 x 32): 4 x 32-site tiles, chunk-major K
    (2, 64, 64, 64, 128, 5, 2),    # data gradient: 4 parity classes (3x3 / 3x2 / 2x3 / 2x2 taps), 32 x 32 site grid
    (2, 8, 32, 256, 256, 3, 1),    # few tiles: split over chunks + reduce epilogue
    (1, 16, 32, 388, 64, 3, 1),    # 49 granules per tap: partial last chunk; N = 64 tile
    (1, 10, 40, 128, 96, 3, 1),    # tile tails in both directions (10 = 2.5 x 4, 40 = 1.25 x 32), N tail
    (1, 32, 32, 64, 64, 1, 1),     # 1x1: halo = tile
    # halo kernel over four ACCUMULATING parity classes (source stride 2: forward of stride-2 convs)
    (1, 38, 70, 40, 96, 3, 2),     # 3x3 s2: classes 2x2 / 2x1 / 1x2 / 1x1 taps, ragged 19 x 35 site grid, N tail
    (2, 64, 128, 16, 64, 7, 2),    # 7x7 s2 (FlowNetS conv1-like, Cin 14 -> 16): classes 4x4 .. 3x3, half-filled chunk, N = 64 tile
    (8, 96, 128, 128, 256, 5, 2),  # conv3 at the step's shape
]


@pytest.mark.parametrize("P", [3, 1])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_planes_vs_fp64(case, P, dev, lib_option):
    lib_option("halo_s2", 2)       # the accumulating-class halo form wherever it applies (by default only from 384 tiles up)
    _conv_case_vs_fp64(case, P, dev)


@pytest.mark.parametrize("k64", [0, 2])
@pytest.mark.parametrize("case", [(2, 48, 64, 128, 128, 3, 1), (1, 12, 16, 476, 256, 3, 1), (1, 16, 32, 388, 64, 3, 1), (2, 64, 64, 64, 128, 5, 2),
                                  (1, 38, 70, 40, 96, 3, 2), (1, 10, 40, 128, 96, 3, 1), (2, 8, 32, 256, 256, 3, 1)])
def test_conv_planes_f16_k64_vs_fp64(case, k64, dev, lib_option):
    """fp16 halo launches with K tiles of 64 channels (f16_k64: two consecutive 32-channel chunks staged as two planes, round 6) forced
    everywhere / off: forward and data gradient of halo-kernel cases incl. a half-filled last chunk (476 -> 480 = 7.5 x 64, 388, 40
    channels), split over chunks, parity and accumulating classes — the same fp64 bound as the K32 form."""
    lib_option("f16_k64", k64)
    lib_option("halo_s2", 2)
    _conv_case_vs_fp64(case, 1, dev)


@pytest.mark.parametrize("case", [(2, 24, 32, 772, 128), (1, 9, 40, 128, 72), (2, 24, 64, 388, 64)])
def test_deconv_planes_f16_k64_vs_fp64(case, dev, lib_option):
    """conv_transpose forward (the launches f16_k64's rule takes: four classes of 2 x 2 taps) and its data gradient in fp16 with K64 tiles."""
    lib_option("f16_k64", 2)
    _deconv_case_vs_fp64(case, 1, dev)


@pytest.mark.parametrize("case", [(2, 48, 64, 128, 256, 3, 1), (1, 12, 16, 476, 256, 3, 1), (2, 8, 32, 256, 256, 3, 1), (4, 48, 64, 256, 512, 3, 2),
                                  (1, 10, 40, 388, 160, 3, 1), (1, 38, 70, 136, 264, 3, 2)])
@pytest.mark.parametrize("db", [0, 1])
def test_conv_planes_f16_tall_tile_vs_fp64(case, db, dev, lib_option):
    """fp16 halo launches on the 256-site x 128 workgroup tile (csrc/conv_halo_tall.hip, option f16_tall forced wherever it can run): forward
    and data gradient incl. N tails (160, 264, 388, 476 columns), split K, ragged site grids (heights 10, 12, 38: partial 8-row tiles),
    parity and accumulating classes — the same fp64 bound as the 128 x 128 tile."""
    lib_option("f16_tall", 2)
    lib_option("f16_db", db)
    lib_option("f16_k64", 0)
    lib_option("halo_s2", 2)
    _conv_case_vs_fp64(case, 1, dev)


@pytest.mark.parametrize("case", [(2, 24, 32, 772, 256), (1, 9, 40, 264, 136)])
def test_deconv_planes_f16_tall_tile_vs_fp64(case, dev, lib_option):
    lib_option("f16_tall", 2)
    lib_option("f16_k64", 0)
    _deconv_case_vs_fp64(case, 1, dev)


@pytest.mark.parametrize("case", [(2, 48, 64, 128, 128, 3, 1), (1, 12, 16, 476, 256, 3, 1), (1, 16, 32, 388, 64, 3, 1), (2, 64, 64, 64, 128, 5, 2),
                                  (1, 38, 70, 40, 96, 3, 2), (1, 10, 40, 128, 96, 3, 1), (2, 8, 32, 256, 256, 1, 1)])
def test_conv_planes_f16_double_buffered_vs_fp64(case, dev, lib_option):
    """fp16 halo launches with the weight tile double-buffered in LDS (option f16_db: the next tile's loads issued at the end of the
    previous iteration, one barrier per K tile, two where a chunk's halo is replaced): 128- and 64-wide tiles, one tap per chunk (1 x 1:
    a new halo every tile), parity and accumulating classes, split K."""
    lib_option("f16_db", 1)
    lib_option("f16_k64", 0)
    lib_option("halo_s2", 2)
    _conv_case_vs_fp64(case, 1, dev)


@pytest.mark.parametrize("case", [(2, 24, 32, 772, 128), (2, 24, 64, 388, 64)])
def test_deconv_planes_f16_double_buffered_vs_fp64(case, dev, lib_option):
    lib_option("f16_db", 1)
    lib_option("f16_k64", 0)
    _deconv_case_vs_fp64(case, 1, dev)


# the persistent stream-K halo kernel (csrc/conv_streamk.hip) forced wherever it is eligible (source stride 1, N > 64, bf16 x 3):
# forward and data gradient of the cases below run through it — single and multiple tap classes (even and uneven tap counts),
# ragged tiles, an N tail, launches with far fewer chunk units than workgroups (empty ranges, items cut into up to 8 segments)
# and the step's own shapes (conv3_1; the 4 / 2 / 2 / 1-tap parity classes of conv4's data gradient)
STREAMK_CASES = [
    (2, 48, 64, 128, 128, 3, 1),
    (2, 8, 32, 256, 256, 3, 1),
    (1, 10, 40, 128, 96, 3, 1),
    (8, 96, 128, 128, 256, 5, 2),
    (4, 48, 64, 256, 512, 3, 2),
    (4, 48, 64, 476, 256, 3, 1),
    (1, 38, 70, 40, 96, 3, 2),      # forward = four ACCUMULATING classes (2x2 / 2x1 / 1x2 / 1x1 taps) walked inside one item, ragged grid
    (2, 96, 128, 64, 128, 5, 2),    # conv2-like forward: accumulating 3x3 / 3x2 / 2x3 / 2x2 classes, 2 chunks each
]


def _streamk_mode(lib_option, form):
    """form 8: the 8-wave ping-pong stream-K kernel forced wherever eligible, 0: the one-shot kernels."""
    lib_option("streamk", 2 if form == 8 else 0)
    lib_option("halo_s2", 2)                             # stride-2 forwards as accumulating classes wherever the form applies


def _streamk_check(run, lib_option, form):
    from unflow_amd import _lib
    _streamk_mode(lib_option, form)
    _lib.lib().unflow_debug_streamk_timeouts()          # clear
    first = run()
    assert _lib.lib().unflow_debug_streamk_timeouts() == 0
    again = run()                                        # fixed-order sums: bit-identical run to run
    assert all(torch.equal(a, b) for a, b in zip(first, again))
    _streamk_mode(lib_option, 0)                         # and within fp32 rounding of the one-shot kernels' results
    plain = run()
    assert all(rel_err(a, b) < 2e-6 for a, b in zip(first, plain))


@pytest.mark.parametrize("case", STREAMK_CASES)
def test_conv_planes_streamk_vs_fp64(case, dev, lib_option):
    _streamk_check(lambda: _conv_case_vs_fp64(case, 3, dev), lib_option, 8)


def _conv_case_vs_fp64(case, P, dev):
    from unflow_amd.core import layers as L
    B, H, W, Cin, Cout, k, stride = case
    g = torch.Generator().manual_seed(zlib.crc32(str(case).encode()))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(k, k, Cin, Cout, generator=g) * (1.0 / np.sqrt(k * k * Cin))
    b = torch.randn(Cout, generator=g) * 0.1
    if P == 1:      # compare against the fp16-rounded operands' exact result?  No: against the true fp32 problem.
        pass
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    pt_, pb = ((k - 1) // 2, k // 2) if stride == 1 else (None, None)
    from oracle import model_ref as M
    y_ref = M.conv2d(xr.permute(0, 3, 1, 2), wr, br, stride, act=True).permute(0, 2, 3, 1)
    gy = torch.randn(y_ref.shape, generator=g).double()
    dz_ref = gy * torch.where(y_ref.detach() > 0, 1.0, 0.1)
    y_ref.backward(gy)

    X = make_pt(x, dev, P, extra=8)
    wd, w_dir, w_tr = weight_planes(w, dev, P)
    Ho, Wo = L.out_hw(H, W, stride)
    Y = L.PT.alloc((B, Ho, Wo, Cout + 8), dev, P)
    Y.t.fill_(7.0)
    Yv = Y.sl(0, Cout)
    L.conv_fwd(X, wd, w_tr, b.to(dev), Yv, stride, True)
    assert rel_err(Yv.t, y_ref) < TOL[P]
    assert torch.all(Y.t[..., Cout:] == 7.0)                                # neighbours of the slice untouched
    # the output planes are the split of the fp32 output, element for element
    got_pl = planes_value(Y.pl.cpu())[..., :Cout]
    if P == 3:
        assert torch.equal(got_pl, Yv.t.cpu().double())
    else:
        assert torch.equal(got_pl.float(), Yv.t.cpu().half().float())

    DZ = make_pt(dz_ref.float(), dev, P, extra=4)
    Cp4 = (Cin + 3) // 4 * 4
    DX = L.PT.alloc((B, H, W, Cp4), dev, P)
    DX.t.fill_(3.0)
    L.conv_bwd_data(DZ, wd, w_dir, DX, stride, accumulate=False)
    assert rel_err(DX.t[..., :Cin], xr.grad) < TOL[P]
    # accumulate + activation-derivative epilogue + planes only for the activated range
    base = torch.randn(B, H, W, Cp4, generator=g).to(dev)
    src = torch.randn(B, H, W, Cp4, generator=g).to(dev)
    DX2 = L.PT.alloc((B, H, W, Cp4), dev, P)
    DX2.t.copy_(base)
    hi = Cp4 // 2 // 4 * 4
    L.conv_bwd_data(DZ, wd, w_dir, DX2, stride, accumulate=True, act_src=src, act_lo=0, act_hi=hi)
    expect = base + DX.t
    slope = torch.where(src > 0, torch.ones_like(src), torch.full_like(src, 0.1))
    expect[..., :hi] *= slope[..., :hi]
    assert rel_err(DX2.t, expect) < TOL[P]
    pl2 = planes_value(DX2.pl.cpu())
    if P == 3:
        assert torch.equal(pl2[..., :hi], DX2.t.cpu().double()[..., :hi])
    assert torch.all(pl2[..., hi:] == 0)                                     # outside [act_lo, act_hi): planes untouched

    dw = torch.full((k, k, Cin, Cout), 9.0, device=dev)
    L.conv_bwd_filter(X, DZ, dw, stride)
    assert rel_err(dw, wr.grad) < (3e-5 if P == 3 else TOL[P])
    return Yv.t.clone(), DX.t.clone(), DX2.t.clone()


@pytest.mark.parametrize("P", [3, 1])
@pytest.mark.parametrize("case", [(2, 24, 32, 64), (1, 17, 70, 24), (3, 64, 128, 64)])
def test_conv1_two_pixel_granules(case, P, dev):
    """FlowNetC's first layer in the two-pixel-granule form (rgb4_form of csrc/conv_planes.hip): input planes with row
    length 4, weight planes of W[7,7,4,Cout] read as [7][28][Cout]; forward and filter gradient vs fp64, including the image
    borders (granule pairs at x = -2,-1 and W, W+1 ... must read zeros) and an odd height."""
    import ctypes
    from unflow_amd import _lib
    from unflow_amd._lib import check, stream
    from unflow_amd.core import layers as L
    from oracle import model_ref as M
    B, H, W, Cout = case
    g = torch.Generator().manual_seed(zlib.crc32(str(case).encode()))
    x = torch.randn(B, H, W, 4, generator=g)
    x[..., 3] = 0
    w = torch.randn(7, 7, 4, Cout, generator=g) * (1.0 / np.sqrt(147))
    b = torch.randn(Cout, generator=g) * 0.1
    xr, wr, br = x.double(), w.double().requires_grad_(), b.double()
    y_ref = M.conv2d(xr.permute(0, 3, 1, 2), wr, br, 2, act=True).permute(0, 2, 3, 1)
    gy = torch.randn(y_ref.shape, generator=g).double()
    dz_ref = gy * torch.where(y_ref.detach() > 0, 1.0, 0.1)
    y_ref.backward(gy)

    X = L.PT(x.to(dev), torch.zeros(P, B, H, W, 4, dtype=torch.int16, device=dev))
    L.planes_from_f32(X.t, X.pl, C=4)
    wd = w.to(dev).contiguous()
    w_dir = torch.zeros(P, 7, 28, Cout, dtype=torch.int16, device=dev)
    w_tr = torch.zeros(P, 7, Cout, 32, dtype=torch.int16, device=dev)
    check(_lib.lib().unflow_weight_planes_batched(1, (ctypes.c_void_p * 1)(wd.data_ptr()), (ctypes.c_int * 1)(7),
                                                  (ctypes.c_int * 1)(28), (ctypes.c_int * 1)(Cout),
                                                  (ctypes.c_void_p * 1)(w_dir.data_ptr()),
                                                  (ctypes.c_void_p * 1)(w_tr.data_ptr()), P, stream()), "weight_planes")
    Ho, Wo = L.out_hw(H, W, 2)
    Y = L.PT.alloc((B, Ho, Wo, Cout), dev, P)
    L.conv_fwd(X, wd, w_tr, b.to(dev), Y, 2, True)
    assert rel_err(Y.t, y_ref) < TOL[P]
    DZ = make_pt(dz_ref.float(), dev, P)
    dw = torch.full((7, 7, 4, Cout), 9.0, device=dev)
    L.conv_bwd_filter(X, DZ, dw, 2)
    assert rel_err(dw, wr.grad) < (3e-5 if P == 3 else TOL[P])
    # same numbers as the one-pixel-granule form
    X8 = make_pt(x, dev, P)
    _, _, w_tr8 = weight_planes(w, dev, P)
    Y8 = L.PT.alloc((B, Ho, Wo, Cout), dev, P)
    L.conv_fwd(X8, wd, w_tr8, b.to(dev), Y8, 2, True)
    assert rel_err(Y.t, Y8.t) < 1e-5


def _planes_value(pl):
    """fp32 value of bf16 x 3 operand planes [3, ...]: hi + mid + lo (exact in fp64)."""
    v = (pl.to(torch.int32) & 0xffff) << 16
    return v.view(torch.float32).double().sum(0)


@pytest.mark.parametrize("leaky", [True, False])
@pytest.mark.parametrize("case", [(2, 64, 128), (1, 50, 76), (3, 16, 64), (8, 384, 512)])
def test_conv_first_layer_kernel_planes_only_vs_fp64(case, leaky, dev, lib_option):
    """FlowNetC's first layer as the step runs it — planes-only output, rgb4 input — on its own kernel (csrc/conv_first.hip):
    against fp64 and against the gather kernel (option conv1_direct = 0), full and ragged tiles, with and without leaky ReLU; the
    planes must be the exact 3-way split of an fp32 value (hi + mid + lo representable)."""
    import ctypes
    from unflow_amd import _lib
    from unflow_amd._lib import check, stream
    from unflow_amd.core import layers as L
    from oracle import model_ref as M
    B, H, W = case
    Cout = 64
    g = torch.Generator().manual_seed(zlib.crc32(str(case).encode()))
    x = torch.randn(B, H, W, 4, generator=g)
    x[..., 3] = 0
    w = torch.randn(7, 7, 4, Cout, generator=g) * (1.0 / np.sqrt(147))
    b = torch.randn(Cout, generator=g) * 0.1
    y_ref = M.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), 2, act=leaky).permute(0, 2, 3, 1)
    X = L.PT(x.to(dev), torch.zeros(3, B, H, W, 4, dtype=torch.int16, device=dev))
    L.planes_from_f32(X.t, X.pl, C=4)
    wd = w.to(dev).contiguous()
    w_dir = torch.zeros(3, 7, 28, Cout, dtype=torch.int16, device=dev)
    w_tr = torch.zeros(3, 7, Cout, 32, dtype=torch.int16, device=dev)
    check(_lib.lib().unflow_weight_planes_batched(1, (ctypes.c_void_p * 1)(wd.data_ptr()), (ctypes.c_int * 1)(7),
                                                  (ctypes.c_int * 1)(28), (ctypes.c_int * 1)(Cout),
                                                  (ctypes.c_void_p * 1)(w_dir.data_ptr()),
                                                  (ctypes.c_void_p * 1)(w_tr.data_ptr()), 3, stream()), "weight_planes")
    Ho, Wo = L.out_hw(H, W, 2)
    outs = []
    for direct in (1, 0):
        lib_option("conv1_direct", direct)
        Y = L.PT.alloc((B, Ho, Wo, Cout), dev, 3)
        Y.pl.fill_(0x7fc0)                      # bf16 NaN pattern: every plane element must be written
        L.conv_fwd(X, wd, w_tr, b.to(dev), Y, 2, leaky, planes_only=True)
        outs.append(_planes_value(Y.pl).cpu())
        if direct and B * H * W >= 8 * 384 * 512:   # the benchmarked shape: six tiles per workgroup, replayed for bit-identity
            # round 4's freely scheduled build of this kernel stored 0x40000000 into a few hundred plane elements per launch,
            # different ones every run (the VMEM store-data hazard behind an SGPR soffset, igemm_shared.h buf_st16_held): 500
            # launches per activation mode = 1,000 full-size launches, every one bit-identical to the first
            first = Y.pl.clone()
            bad = torch.zeros((), dtype=torch.int64, device=dev)
            for _ in range(500):
                Y.pl.fill_(0x7fc0)
                L.conv_fwd(X, wd, w_tr, b.to(dev), Y, 2, leaky, planes_only=True)
                bad += (Y.pl != first).sum()
            assert bad.item() == 0, bad.item()
    scale = y_ref.abs().max().item()
    for o in outs:
        assert not torch.isnan(o).any()
        assert (o - y_ref).abs().max().item() <= 5e-6 * scale
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-6 * scale


# (B, H, W, Cin, Cout)   H,W = INPUT size; output is 2H x 2W
DECONV_CASES = [
    (8, 6, 8, 1024, 512),     # deconv5
    (2, 12, 16, 1028, 256),   # deconv4: 1028 -> 1032 plane channels
    (1, 24, 32, 772, 128),    # deconv3
    (1, 24, 32, 388, 64),     # deconv2
    (2, 12, 16, 76, 12),      # 3/8-width full_res deconv1: Cout 12
    (2, 24, 64, 388, 64),     # data gradient through the halo kernel (accumulating 2x2-tap classes): 24 x 64 site grid
    (1, 9, 40, 128, 40),      # ... ragged tiles
]


@pytest.mark.parametrize("P", [3, 1])
@pytest.mark.parametrize("case", DECONV_CASES)
def test_deconv_planes_vs_fp64(case, P, dev, lib_option):
    lib_option("halo_s2", 2)
    _deconv_case_vs_fp64(case, P, dev)


@pytest.mark.parametrize("case", [(1, 24, 32, 772, 128), (4, 24, 32, 772, 128), (1, 9, 40, 128, 72), (2, 24, 64, 388, 64),
                                  (1, 9, 40, 128, 40)])
def test_deconv_planes_streamk_vs_fp64(case, dev, lib_option):
    """conv_transpose on the persistent stream-K halo kernel: forward = four output-parity classes of 2 x 2 taps; data gradient
    (the last two cases: N = Cin > 64) = four ACCUMULATING 2 x 2-tap classes on the parity sub-lattices of dz."""
    _streamk_check(lambda: _deconv_case_vs_fp64(case, 3, dev), lib_option, 8)


def _deconv_case_vs_fp64(case, P, dev):
    from unflow_amd.core import layers as L
    from oracle import model_ref as M
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(zlib.crc32(str(case).encode()))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(4, 4, Cout, Cin, generator=g) * (1.0 / np.sqrt(4 * Cin))
    b = torch.randn(Cout, generator=g) * 0.1
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    y_ref = M.conv2d_transpose(xr.permute(0, 3, 1, 2), wr, br, act=True).permute(0, 2, 3, 1)
    gy = torch.randn(y_ref.shape, generator=g).double()
    dz_ref = gy * torch.where(y_ref.detach() > 0, 1.0, 0.1)
    y_ref.backward(gy)

    X = make_pt(x, dev, P)
    wd, w_dir, w_tr = weight_planes(w, dev, P)
    Y = L.PT.alloc((B, 2 * H, 2 * W, Cout + 4), dev, P)
    Yv = Y.sl(0, Cout)
    L.deconv_fwd(X, wd, w_dir, b.to(dev), Yv, True)
    assert rel_err(Yv.t, y_ref) < TOL[P]
    if P == 3:
        assert torch.equal(planes_value(Y.pl.cpu())[..., :Cout], Yv.t.cpu().double())
    DZ = make_pt(dz_ref.float(), dev, P)
    DX = L.PT.alloc((B, H, W, Cin), dev, P)
    DX.t.fill_(5.0)
    L.deconv_bwd_data(DZ, wd, w_tr, DX, accumulate=False)
    assert rel_err(DX.t, xr.grad) < TOL[P]
    DX2 = L.PT.alloc((B, H, W, Cin), dev, P)
    DX2.t.fill_(1.0)
    L.deconv_bwd_data(DZ, wd, w_tr, DX2, accumulate=True)
    assert rel_err(DX2.t, DX.t + 1.0) < TOL[P]
    dw = torch.full((4, 4, Cout, Cin), 9.0, device=dev)
    L.deconv_bwd_filter(X, DZ, dw)
    assert rel_err(dw, wr.grad) < (3e-5 if P == 3 else TOL[P])
    return Y.t.clone(), DX.t.clone()


def test_planes_kernels_match_inline_split_bitwise_close(dev):
    """The plane kernels and conv_igemm.hip's in-kernel split compute the SAME six bf16 product terms; only the fp32
    accumulation order inside a K tile differs (term-major vs plane order) — results agree to fp32 rounding."""
    from unflow_amd.core import layers as L
    B, H, W, Cin, Cout, k = 2, 48, 64, 128, 128, 3
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(k, k, Cin, Cout, generator=g) / np.sqrt(k * k * Cin)
    X = make_pt(x, dev, 3)
    wd, w_dir, w_tr = weight_planes(w, dev, 3)
    y_pl = torch.zeros(B, H, W, Cout, device=dev)
    L.conv_fwd(X, wd, w_tr, None, y_pl, 1, False)
    y_in = torch.zeros(B, H, W, Cout, device=dev)
    L.conv2d_fwd(X.t, wd, None, y_in, 1, False)
    assert rel_err(y_pl, y_in) < 2e-6


CORR_PL_CASES = [
    # N (directed batch), C, H, W, attrs
    (8, 256, 48, 64, dict(kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2)),   # the step's shape
    (4, 96, 16, 24, dict(kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2)),    # 3/8-width features
    (2, 64, 12, 40, dict(kernel_size=1, max_displacement=4, pad=4, stride_1=1, stride_2=1)),      # 81 channels, 2 site tiles
    (2, 32, 9, 11, dict(kernel_size=1, max_displacement=3, pad=5, stride_1=1, stride_2=1)),       # pad > displacement
    # narrow-band tiling (r <= 6 over several site tiles: 32 - 2r owned sites per tile, one Gram per displacement row)
    (2, 64, 6, 131, dict(kernel_size=1, max_displacement=4, pad=4, stride_1=1, stride_2=1)),      # 81 ch, 6 tiles, ragged
    (2, 48, 5, 100, dict(kernel_size=1, max_displacement=6, pad=6, stride_1=1, stride_2=1)),      # r = 6: 20 owned sites
    (2, 32, 7, 101, dict(kernel_size=1, max_displacement=8, pad=8, stride_1=1, stride_2=2)),      # r = 4 in two classes
    (2, 32, 6, 90, dict(kernel_size=1, max_displacement=2, pad=4, stride_1=1, stride_2=1)),       # pad > displacement
    (2, 256, 6, 70, dict(kernel_size=1, max_displacement=4, pad=4, stride_1=1, stride_2=1)),      # 4 waves (K split), 3 tiles
    (2, 128, 5, 60, dict(kernel_size=1, max_displacement=6, pad=6, stride_1=1, stride_2=1)),      # 2 waves, r = 6
    # wide band by DMA (C % 64 == 0): row pairs, neighbour column tiles inside / outside the image, odd row counts
    (2, 128, 11, 150, dict(kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2)),  # 3 tiles per class
    (2, 128, 7, 80, dict(kernel_size=1, max_displacement=10, pad=10, stride_1=1, stride_2=1)),    # r = 10 at stride_2 = 1
    (2, 128, 9, 40, dict(kernel_size=1, max_displacement=8, pad=12, stride_1=1, stride_2=1)),     # pad > displacement
    (2, 192, 5, 33, dict(kernel_size=1, max_displacement=14, pad=14, stride_1=1, stride_2=2)),    # 3 waves, r = 7
    # a wave pair per output row (C = 128 / 256): groups of four rows, ragged last group, two column tiles, both stride_2
    (2, 256, 13, 70, dict(kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2)),   # 7 + 6 rows per class, 2 tiles
    (2, 256, 9, 30, dict(kernel_size=1, max_displacement=10, pad=10, stride_1=1, stride_2=1)),    # 9 rows: groups of 4, 4, 1
    (4, 256, 10, 64, dict(kernel_size=1, max_displacement=16, pad=18, stride_1=1, stride_2=2)),   # r = 8, pad > displacement
]
# the cases corr_fwd_rw_kernel takes by default; with corr_rw = 0 they run on corr_fwd_wb_kernel (its A/B, and the C = 64 / 192 kernel)
CORR_RW_CASES = [c for c in CORR_PL_CASES if c[1] in (128, 256) and not (c[4]['max_displacement'] // c[4]['stride_2'] <= 6 and c[3] / c[4]['stride_2'] > 32)]


# the cases the row-shared narrow-band kernel takes by default (stride_2 = 1, r <= 4, several site tiles, C % 32 == 0); with
# corr_rs = 0 they run on corr_fwd_nb_kernel (C % 64 == 0) or the streaming kernel
CORR_RS_CASES = [c for c in CORR_PL_CASES if c[4]['stride_2'] == 1 and c[4]['max_displacement'] <= 4 and c[3] > 32 and c[1] % 32 == 0]


@pytest.mark.parametrize("case", CORR_RS_CASES + [(4, 256, 21, 200, dict(kernel_size=1, max_displacement=4, pad=4, stride_1=1, stride_2=1)),
                                                   (2, 96, 10, 77, dict(kernel_size=1, max_displacement=3, pad=3, stride_1=1, stride_2=1)),
                                                   # the ring kernel (r = 4): one chunk, three chunks, pad > displacement
                                                   (2, 32, 9, 50, dict(kernel_size=1, max_displacement=4, pad=4, stride_1=1, stride_2=1)),
                                                   (2, 96, 17, 33, dict(kernel_size=1, max_displacement=4, pad=4, stride_1=1, stride_2=1)),
                                                   (2, 64, 9, 40, dict(kernel_size=1, max_displacement=4, pad=6, stride_1=1, stride_2=1))])
def test_correlation_planes_fwd_row_shared_and_narrow_band_kernels_vs_oracle(case, dev, oracle_lib, lib_option):
    """The +-4 / 81-channel cost volume (and r = 2, 3): the default kernels (r = 4: f1 rows streamed through an LDS ring, row
    groups of 8 with a ragged last group, 9 site tiles, a ragged last tile; r < 4: the row-shared kernel, groups of 4; C = 96:
    three chunks), into a concat buffer and as a dense cost volume (16-byte stores); with corr_rs = 1 the row-shared kernel at
    r = 4 too, with corr_rs = 0 the kernels both replaced."""
    test_correlation_planes_fwd_vs_oracle(case, dev, oracle_lib)
    test_correlation_planes_fwd_vs_oracle(case, dev, oracle_lib, ld_extra=0)
    lib_option("corr_rs", 1)
    test_correlation_planes_fwd_vs_oracle(case, dev, oracle_lib)
    lib_option("corr_rs", 0)
    test_correlation_planes_fwd_vs_oracle(case, dev, oracle_lib)


@pytest.mark.parametrize("case", CORR_RW_CASES)
def test_correlation_planes_fwd_wide_band_kernel_vs_oracle(case, dev, oracle_lib, lib_option):
    """The K-split wide-band kernel on the shapes the wave-pair kernel takes by default."""
    lib_option("corr_rw", 0)
    test_correlation_planes_fwd_vs_oracle(case, dev, oracle_lib)


@pytest.mark.parametrize("case", CORR_PL_CASES)
def test_correlation_planes_fwd_vs_oracle(case, dev, oracle_lib, ld_extra=3):
    """unflow_correlation_nhwc_fwd_pl (bf16 matrix cores, six-term split) vs the scalar C oracle of CorrelateData
    (ops/correlation_op.cu.cc:51-117), paired like the training step (sample n with (n + N/2) % N).  The output rows carry
    `ld_extra` foreign channels behind the cost volume (the step writes it into a concat buffer); 0 = a dense cost volume."""
    from unflow_amd import _lib
    from unflow_amd._lib import check, ptr, stream
    N, C, H, W, attrs = case
    B = N // 2
    rs = np.random.RandomState(zlib.crc32(str(case).encode()))
    feat = torch.from_numpy(rs.randn(N, H, W, C).astype(np.float32))
    F = make_pt(feat, dev, 3, extra=8)
    oc, oh, ow = oracle_lib.correlation_out_shape(H, W, **attrs)
    out = torch.zeros(N, oh, ow, oc + ld_extra, device=dev)
    out[..., :oc] = float('nan')            # every output is written (also the all-zero Grams outside the image)
    a = attrs
    check(_lib.lib().unflow_correlation_nhwc_fwd_pl(ptr(F.t), ptr(F.t), F.t.stride(2), _lib.planes_of(F.pl), _lib.planes_of(F.pl),
                                                    B, ptr(out), oc + ld_extra, N, C, H, W, a['kernel_size'], a['max_displacement'],
                                                    a['pad'], a['stride_1'], a['stride_2'], stream()), "correlation_pl")
    x = np.ascontiguousarray(feat.numpy().transpose(0, 3, 1, 2))
    ref = oracle_lib.correlation(x, np.ascontiguousarray(np.roll(x, -B, axis=0)), **attrs)
    got = out[..., :oc].permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    if ld_extra:
        assert out[..., oc:].abs().max().item() == 0


# (B, H, W, Cin, Cout, k, stride): shapes whose forward / data gradient split K (few output tiles, deep K)
SPLITK_CASES = [
    (8, 96, 128, 128, 256, 5, 2),    # conv3: gather kernel forward with 2 slices per tile (fused), halo data gradient
    (8, 48, 64, 256, 256, 3, 1),     # conv3_1-like: halo kernel, 4 slices over channel chunks (fused)
    (8, 48, 64, 256, 512, 3, 2),     # conv4: gather forward 4 slices (fused), 4-class data gradient
    (8, 12, 16, 512, 512, 3, 1),     # conv5_1: 16 slices -> the reduce kernel
]


def _splitk_outputs(case, dev, reps):
    """Forward + data gradient of one layer, `reps` times into fresh buffers: list of (y, dx) CPU tensors."""
    from unflow_amd.core import layers as L
    B, H, W, Cin, Cout, k, stride = case
    g = torch.Generator().manual_seed(zlib.crc32(str(case).encode()))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(k, k, Cin, Cout, generator=g) * (1.0 / np.sqrt(k * k * Cin))
    b = torch.randn(Cout, generator=g) * 0.1
    Ho, Wo = L.out_hw(H, W, stride)
    dz = torch.randn(B, Ho, Wo, Cout, generator=g)
    X, DZ = make_pt(x, dev, 3), make_pt(dz, dev, 3)
    wd, w_dir, w_tr = weight_planes(w, dev, 3)
    outs = []
    for _ in range(reps):
        Y = L.PT.alloc((B, Ho, Wo, Cout), dev, 3)
        DX = L.PT.alloc((B, H, W, Cin), dev, 3)
        L.conv_fwd(X, wd, w_tr, b.to(dev), Y, stride, True)
        L.conv_bwd_data(DZ, wd, w_dir, DX, stride, accumulate=False)
        # a second, dependent launch right behind (the hand-off must also hold with the consumer hot on the producer's heels)
        L.conv_bwd_data(DZ, wd, w_dir, DX, stride, accumulate=True)
        outs.append((Y.t.cpu(), DX.t.cpu(), Y.pl.cpu()))
    return outs


@pytest.fixture
def lib_option():
    """Set library options (csrc/options.h) for one test; restored afterwards."""
    from unflow_amd import _lib
    saved = {}

    def setter(name, value):
        if name not in saved:
            saved[name] = _lib.get_option(name)
        _lib.set_option(name, value)
    yield setter
    for k, v in saved.items():
        _lib.set_option(k, v)


@pytest.mark.parametrize("case", SPLITK_CASES)
def test_splitk_reduce_is_bit_stable(case, dev):
    """Split-K layers (partial tiles + the fixed-order reduce / epilogue pass): forward, data gradient and the output planes are
    bit-identical over 10 back-to-back launches, with a dependent accumulating launch right behind each."""
    o = _splitk_outputs(case, dev, 10)
    assert all(torch.equal(a, b) for q in o[1:] for a, b in zip(q, o[0])), "split-K result not stable run to run"


def test_library_options_roundtrip(dev):
    """unflow_set_option / unflow_get_option: every name of unflow_option_names round-trips; unknown names are refused."""
    from unflow_amd import _lib
    names = _lib.option_names()
    assert "conv_math_fp32" in names and "streamk" in names
    for n in names:
        v = _lib.get_option(n)
        _lib.set_option(n, v)
        assert _lib.get_option(n) == v
    assert _lib.lib().unflow_set_option(b"no_such_option", 1) == -7


# (B, H, W, Cin, Cout, k, stride, deconv): filter gradients at the step's split-K plans
WGRAD_CASES = [
    (8, 96, 128, 128, 256, 5, 2, False),   # conv3: 50 tiles x 15 splits
    (8, 48, 64, 476, 256, 3, 1, False),    # conv3_1: 68 tiles (last M tile 96 of 128 rows)
    (8, 192, 256, 64, 128, 5, 2, False),   # conv2: 13 tiles x 59 splits (two-level partial sums)
    (8, 48, 64, 388, 64, 4, 2, True),      # deconv2-like: N = 388 (4 real columns in the last N tile)
    (2, 40, 56, 72, 40, 3, 1, False),      # ragged everything, 128 x 64 tiles
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad_planes_vs_fp64(case, dev, lib_option):
    """Filter gradients of the plane kernels (LDS-DMA form and, option wgrad_dma = 0, the register-staged form) against an
    fp64 torch reference at the step's large shapes; repeated launches bit-identical."""
    from unflow_amd.core import layers as L
    import torch.nn.functional as F
    B, H, W, Cin, Cout, k, stride, deconv = case
    g = torch.Generator().manual_seed(zlib.crc32(str(case).encode()))
    if deconv:       # conv_transpose [B,H/2,W/2,Cin] -> [B,H,W,Cout]; dW [4,4,Cout,Cin]
        x = torch.randn(B, H // 2, W // 2, Cin, generator=g)
        dz = torch.randn(B, H, W, Cout, generator=g)
        xd, dzd = x.double().permute(0, 3, 1, 2), dz.double().permute(0, 3, 1, 2)
        wz = torch.zeros(Cin, Cout, 4, 4, dtype=torch.float64, requires_grad=True)
        y = F.conv_transpose2d(xd, wz, stride=2, padding=1)
        (y * dzd).sum().backward()
        ref = wz.grad.permute(2, 3, 1, 0).contiguous()           # [4,4,Cout,Cin]
        dw_shape = (4, 4, Cout, Cin)
    else:
        x = torch.randn(B, H, W, Cin, generator=g)
        Ho, Wo = L.out_hw(H, W, stride)
        dz = torch.randn(B, Ho, Wo, Cout, generator=g)
        pt_, pl_ = ((Ho - 1) * stride + k - H), ((Wo - 1) * stride + k - W)
        pt_, pl_ = max(pt_, 0), max(pl_, 0)
        xp = F.pad(x.double().permute(0, 3, 1, 2), (pl_ // 2, pl_ - pl_ // 2, pt_ // 2, pt_ - pt_ // 2))
        wz = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(xp, wz, stride=stride)
        (y * dz.double().permute(0, 3, 1, 2)).sum().backward()
        ref = wz.grad.permute(2, 3, 1, 0).contiguous()           # HWIO
        dw_shape = (k, k, Cin, Cout)
    X, DZ = make_pt(x, dev, 3), make_pt(dz, dev, 3)
    results = {}
    for dma in (1, 0):
        lib_option("wgrad_dma", dma)
        outs = []
        for _ in range(3):
            dw = torch.full(dw_shape, float('nan'), device=dev)
            if deconv:
                L.deconv_bwd_filter(X, DZ, dw)
            else:
                L.conv_bwd_filter(X, DZ, dw, stride)
            outs.append(dw.cpu())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        err = (outs[0].double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-5, (dma, err)
        results[dma] = outs[0]


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("case", [(8, 48, 64, 476, 256, 3, 1, False), (8, 96, 128, 128, 256, 5, 2, False), (2, 40, 56, 72, 40, 3, 1, False),
                                  (8, 48, 64, 388, 64, 4, 2, True), (4, 24, 32, 1028, 256, 4, 2, True)])
def test_wgrad_planes_f16_forms_vs_fp64(case, mode, dev, lib_option):
    """fp16 filter gradients in their three forms (f16_wgrad_dma 0: register-staged; 1: the LDS-DMA kernel with stages of three
    16-site groups — the default; 2: also its 8-wave ping-pong form) against fp64: ragged site counts (a last stage with one or two
    empty groups), image-border taps, N tails; repeated launches bit-identical."""
    from unflow_amd.core import layers as L
    import torch.nn.functional as F
    lib_option("f16_wgrad_dma", mode)
    if mode == 2:
        lib_option("wgrad_pp", 3)
    B, H, W, Cin, Cout, k, stride, deconv = case
    g = torch.Generator().manual_seed(zlib.crc32(str(case).encode()))
    if deconv:
        x = torch.randn(B, H // 2, W // 2, Cin, generator=g)
        dz = torch.randn(B, H, W, Cout, generator=g)
        xh, dzh = x.half().double(), dz.half().double()              # the operands the kernel sees: fp16-rounded
        wz = torch.zeros(Cin, Cout, 4, 4, dtype=torch.float64, requires_grad=True)
        y = F.conv_transpose2d(xh.permute(0, 3, 1, 2), wz, stride=2, padding=1)
        (y * dzh.permute(0, 3, 1, 2)).sum().backward()
        ref = wz.grad.permute(2, 3, 1, 0).contiguous()
        dw_shape = (4, 4, Cout, Cin)
    else:
        x = torch.randn(B, H, W, Cin, generator=g)
        Ho, Wo = L.out_hw(H, W, stride)
        dz = torch.randn(B, Ho, Wo, Cout, generator=g)
        xh, dzh = x.half().double(), dz.half().double()
        pt_, pl_ = max((Ho - 1) * stride + k - H, 0), max((Wo - 1) * stride + k - W, 0)
        xp = F.pad(xh.permute(0, 3, 1, 2), (pl_ // 2, pl_ - pl_ // 2, pt_ // 2, pt_ - pt_ // 2))
        wz = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(xp, wz, stride=stride)
        (y * dzh.permute(0, 3, 1, 2)).sum().backward()
        ref = wz.grad.permute(2, 3, 1, 0).contiguous()
        dw_shape = (k, k, Cin, Cout)
    X, DZ = make_pt(x, dev, 1), make_pt(dz, dev, 1)
    outs = []
    for _ in range(3):
        dw = torch.full(dw_shape, float('nan'), device=dev)
        if deconv:
            L.deconv_bwd_filter(X, DZ, dw)
        else:
            L.conv_bwd_filter(X, DZ, dw, stride)
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    err = (outs[0].double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-5, (mode, err)                                   # fp16 operands are exact inputs here: only the fp32 accumulation differs


@pytest.mark.parametrize("case", [(2, 64, 12, 16, 20, 2), (1, 256, 24, 32, 20, 2), (2, 128, 9, 21, 4, 1),
                                  # narrow-band tiling (correlation_planes.hip: corr_pl_tiles)
                                  (1, 64, 6, 131, 4, 1), (2, 64, 5, 100, 6, 1), (1, 128, 7, 101, 8, 2),
                                  # C % 256 == 0 (band operand shared by the four channel-group waves of a workgroup): 3 site tiles at
                                  # r = 4, the wide band at r = 10, odd row counts in both row classes, r = 4 in two classes, C = 512
                                  (2, 256, 13, 70, 4, 1), (1, 256, 7, 30, 10, 1), (1, 256, 9, 40, 20, 2), (1, 256, 7, 101, 8, 2),
                                  (1, 512, 6, 33, 8, 2)])
def test_correlation_planes_bwd_vs_fp32_kernel(case, dev):
    """unflow_correlation_nhwc_bwd_pl (feature operand from the bf16 planes through LDS-DMA + transposing reads, band operand
    split in registers, six terms on the bf16 matrix cores) vs the fp32-MFMA backward of the same library (itself checked
    against the scalar C oracle in tests/test_ops_gpu.py), fused g0 + g1 with the training step's pairing."""
    import os
    import subprocess
    import sys
    from unflow_amd import _lib
    from unflow_amd._lib import check, ptr, stream
    B, C, H, W, md, s2 = case
    N = 2 * B
    g = torch.Generator().manual_seed(zlib.crc32(str(case).encode()))
    x = torch.randn(N, H, W, C, generator=g)
    F = make_pt(x, dev, 3)
    import ctypes
    o3 = (ctypes.c_int * 3)()
    assert _lib.lib().unflow_correlation_out_shape(H, W, 1, md, md, 1, s2, o3) == 0
    oc, oh, ow = tuple(o3)
    dout = torch.randn(N, oh, ow, oc, generator=g).to(dev)
    g_pl = torch.zeros(N, H, W, C, device=dev)
    g_ref = torch.zeros(N, H, W, C, device=dev)
    check(_lib.lib().unflow_correlation_nhwc_bwd_pl(ptr(dout), oc, ptr(F.t), ptr(F.t), F.t.stride(2), _lib.planes_of(F.pl),
                                                    _lib.planes_of(F.pl), B, ptr(g_pl), ptr(None), C, 1, N, C, H, W, 1, md, md, 1, s2,
                                                    stream()), "correlation_bwd_pl")
    check(_lib.lib().unflow_correlation_nhwc_bwd_pl(ptr(dout), oc, ptr(F.t), ptr(F.t), F.t.stride(2), None, None, B, ptr(g_ref),
                                                    ptr(None), C, 1, N, C, H, W, 1, md, md, 1, s2, stream()), "correlation_bwd")
    assert rel_err(g_pl, g_ref) < 2e-5
    # the two gradients kept apart (grad0 / grad1 of the reference op), planes against fp32 kernels
    ga, gb = torch.full((N, H, W, C), float('nan'), device=dev), torch.full((N, H, W, C), float('nan'), device=dev)
    ra, rb = torch.zeros(N, H, W, C, device=dev), torch.zeros(N, H, W, C, device=dev)
    check(_lib.lib().unflow_correlation_nhwc_bwd_pl(ptr(dout), oc, ptr(F.t), ptr(F.t), F.t.stride(2), _lib.planes_of(F.pl),
                                                    _lib.planes_of(F.pl), B, ptr(ga), ptr(gb), C, 0, N, C, H, W, 1, md, md, 1, s2,
                                                    stream()), "correlation_bwd_pl")
    check(_lib.lib().unflow_correlation_nhwc_bwd_pl(ptr(dout), oc, ptr(F.t), ptr(F.t), F.t.stride(2), None, None, B, ptr(ra),
                                                    ptr(rb), C, 0, N, C, H, W, 1, md, md, 1, s2, stream()), "correlation_bwd")
    assert rel_err(ga, ra) < 2e-5 and rel_err(gb, rb) < 2e-5

