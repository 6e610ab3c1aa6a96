"""Channels-last conv / conv_transpose layer calls into libunflow_hip.so — the replacement for the
slim.conv2d / slim.conv2d_transpose calls of src/e2eflow/core/flownet.py:89-237 (TF 'SAME' padding,
leaky-ReLU 0.1, HWIO weights; conv_transpose weights [k,k,out,in]).

Tensors are NHWC float32 CUDA tensors or channel-slice views of them (stride(3) == 1); a view lets a
layer read from / write into a slice of a concat buffer without a copy.
"""
import torch

from .. import _lib
from .._lib import check, ptr, stream, csz
from ..ops import workspace


def nhwc(t):
    """(ptr, ld, B, H, W, C) of an NHWC tensor or channel-slice view."""
    assert t.dim() == 4 and t.dtype == torch.float32 and t.is_cuda, "NHWC float32 CUDA tensor expected"
    B, H, W, C = t.shape
    ld = t.stride(2)
    assert t.stride(3) == 1 and t.stride(1) == W * ld and t.stride(0) == H * W * ld, \
        "tensor must be a channel-slice view of a contiguous NHWC buffer"
    return ptr(t), ld, B, H, W, C


def out_hw(H, W, stride):
    return -(-H // stride), -(-W // stride)


_WS_SLOT = [1]


class ws_slot:
    """Select the split-K scratch buffer for the calls inside the block (one buffer per concurrently running stream)."""

    def __init__(self, slot):
        self.slot = slot

    def __enter__(self):
        self.prev, _WS_SLOT[0] = _WS_SLOT[0], self.slot

    def __exit__(self, *exc):
        _WS_SLOT[0] = self.prev


def _ws(device, B, H, W, Cin, Cout, k, stride):
    n = _lib.lib().unflow_conv_workspace_bytes(B, H, W, Cin, Cout, k, stride)
    t = workspace(n, device, slot=_WS_SLOT[0])
    return ptr(t), csz(t.numel() * 4)


def conv2d_fwd(x, w, bias, y, stride, leaky):
    xp, ldx, B, H, W, Cin = nhwc(x)
    yp, ldy, _, Ho, Wo, Cout = nhwc(y)
    k = w.shape[0]
    assert tuple(w.shape) == (k, k, Cin, Cout) and w.is_contiguous()
    assert (Ho, Wo) == out_hw(H, W, stride)
    wsp, wsn = _ws(x.device, B, H, W, Cin, Cout, k, stride)
    check(_lib.lib().unflow_conv2d_fwd(xp, ldx, ptr(w), ptr(bias), yp, ldy, B, H, W, Cin, Cout, k, stride,
                                       int(bool(leaky)), wsp, wsn, stream()), "conv2d_fwd")
    return y


def conv2d_bwd_data(dz, w, dx, stride, accumulate=False, act_src=None, act_lo=0, act_hi=0):
    dzp, lddz, B, Ho, Wo, Cout = nhwc(dz)
    dxp, lddx, _, H, W, Cin = nhwc(dx)
    k = w.shape[0]
    assert tuple(w.shape) == (k, k, Cin, Cout)
    ap, lda = (ptr(None), 0)
    if act_src is not None:
        ap, lda = nhwc(act_src)[:2]
    wsp, wsn = _ws(dz.device, B, H, W, Cin, Cout, k, stride)
    check(_lib.lib().unflow_conv2d_bwd_data(dzp, lddz, ptr(w), dxp, lddx, B, H, W, Cin, Cout, k, stride,
                                            int(bool(accumulate)), ap, lda, act_lo, act_hi, wsp, wsn, stream()),
          "conv2d_bwd_data")
    return dx


def conv2d_bwd_filter(x, dz, dw, dbias, stride):
    xp, ldx, B, H, W, Cin = nhwc(x)
    dzp, lddz, _, Ho, Wo, Cout = nhwc(dz)
    k = dw.shape[0]
    assert tuple(dw.shape) == (k, k, Cin, Cout) and dw.is_contiguous()
    wsp, wsn = _ws(x.device, B, H, W, Cin, Cout, k, stride)
    check(_lib.lib().unflow_conv2d_bwd_filter(xp, ldx, dzp, lddz, ptr(dw), ptr(dbias), B, H, W, Cin, Cout, k, stride,
                                              wsp, wsn, stream()), "conv2d_bwd_filter")
    return dw, dbias


def conv2d_transpose_fwd(x, w, bias, y, leaky):
    xp, ldx, B, H, W, Cin = nhwc(x)
    yp, ldy, _, Ho, Wo, Cout = nhwc(y)
    assert tuple(w.shape) == (4, 4, Cout, Cin) and (Ho, Wo) == (2 * H, 2 * W)
    wsp, wsn = _ws(x.device, B, 2 * H, 2 * W, Cin, Cout, 4, 2)
    check(_lib.lib().unflow_conv2d_transpose_fwd(xp, ldx, ptr(w), ptr(bias), yp, ldy, B, H, W, Cin, Cout,
                                                 int(bool(leaky)), wsp, wsn, stream()), "conv2d_transpose_fwd")
    return y


def conv2d_transpose_bwd_data(dz, w, dx, accumulate=False, act_src=None, act_lo=0, act_hi=0):
    dzp, lddz, B, Ho, Wo, Cout = nhwc(dz)
    dxp, lddx, _, H, W, Cin = nhwc(dx)
    assert tuple(w.shape) == (4, 4, Cout, Cin) and (Ho, Wo) == (2 * H, 2 * W)
    ap, lda = (ptr(None), 0)
    if act_src is not None:
        ap, lda = nhwc(act_src)[:2]
    wsp, wsn = _ws(dz.device, B, Ho, Wo, Cin, Cout, 4, 2)
    check(_lib.lib().unflow_conv2d_transpose_bwd_data(dzp, lddz, ptr(w), dxp, lddx, B, H, W, Cin, Cout,
                                                      int(bool(accumulate)), ap, lda, act_lo, act_hi, wsp, wsn,
                                                      stream()), "conv2d_transpose_bwd_data")
    return dx


def conv2d_transpose_bwd_filter(x, dz, dw, dbias):
    xp, ldx, B, H, W, Cin = nhwc(x)
    dzp, lddz, _, Ho, Wo, Cout = nhwc(dz)
    assert tuple(dw.shape) == (4, 4, Cout, Cin)
    wsp, wsn = _ws(x.device, B, Ho, Wo, Cin, Cout, 4, 2)
    check(_lib.lib().unflow_conv2d_transpose_bwd_filter(xp, ldx, dzp, lddz, ptr(dw), ptr(dbias), B, H, W, Cin, Cout,
                                                        wsp, wsn, stream()), "conv2d_transpose_bwd_filter")
    return dw, dbias


def leaky_bwd_inplace(dy, y):
    dp, ldd, B, H, W, C = nhwc(dy)
    yp, ldy = nhwc(y)[:2]
    check(_lib.lib().unflow_leaky_bwd_inplace(dp, ldd, yp, ldy, _lib.cl(B * H * W), C, stream()), "leaky_bwd")
    return dy


# ---------------------------------------------------------------------------------------------------------------
# Operand planes (csrc/conv_planes.hip): the same layer calls with the tensors' 16-bit planes riding along.
# ---------------------------------------------------------------------------------------------------------------
def round8(c):
    return (c + 7) // 8 * 8


class PT:
    """An NHWC fp32 tensor (or channel-slice view) together with its operand planes: `pl` is an int16 view
    [P, N, H, W, Cp] of the same pixels (P = 3: bf16 hi/mid/lo, P = 1: fp16; Cp >= round8(C), pad channels zero), or None.
    `scale`: the planes hold scale * value (fp16 planes of gradient tensors: a power of two that keeps small gradients in
    fp16's normal range; 0 / 1 otherwise)."""
    __slots__ = ('t', 'pl', 'scale')

    def __init__(self, t, pl=None, scale=0.0):
        self.t, self.pl, self.scale = t, pl, scale

    def sl(self, lo, hi):
        """Channel slice [lo, hi): the planes view keeps the channels up to round8 of the slice width (zero pad or, for
        an interior slice whose width is a multiple of 8, exactly the slice)."""
        if self.pl is None:
            return PT(self.t[..., lo:hi])
        return PT(self.t[..., lo:hi], self.pl[..., lo:min(self.pl.shape[-1], lo + round8(hi - lo))], self.scale)

    def planes(self):
        return _lib.planes_of(self.pl, self.scale)

    @staticmethod
    def alloc(shape, device, n_planes, scale=0.0):
        """Zero tensor + zero planes (n_planes 0: no planes)."""
        import torch as _t
        t = _t.zeros(*shape, dtype=_t.float32, device=device)
        # plane rows are padded to 64 bytes (32 channels): every 64-byte K-tile row of the gather kernels and every 256-byte
        # site row of the filter-gradient kernel is then ONE aligned L1 access (TCP_TOTAL_CACHE_ACCESSES per buffer load:
        # 18 / 30 with 8-channel padding, 16 is the minimum)
        pad = 32
        cp = (shape[-1] + pad - 1) // pad * pad if shape[-1] > 8 else round8(shape[-1])
        pl = _t.zeros(n_planes, *shape[:-1], cp, dtype=_t.int16, device=device) if n_planes else None
        return PT(t, pl, scale if n_planes == 1 else 0.0)


def _pt(x):
    return x if isinstance(x, PT) else PT(x)


def _ws_pl(device, B, H, W, Cin, Cout, k, stride, npl):
    n = _lib.lib().unflow_conv_pl_workspace_bytes(B, H, W, Cin, Cout, k, stride, npl)
    t = workspace(n, device, slot=_WS_SLOT[0])
    return ptr(t), csz(t.numel() * 4)


def _npl(*pts):
    for q in pts:
        if q.pl is not None:
            return q.pl.shape[0]
    return 0


def planes_from_f32(x, out_pl, C=None):
    """Fill the planes view `out_pl` [P,N,H,W,>=round8(C)] from the fp32 NHWC tensor / slice x (pad channels zeroed)."""
    xp, ldx, B, H, W, Cx = nhwc(x)
    C = Cx if C is None else C
    fill = min(round8(C), out_pl.shape[-1])         # a slice may end the buffer row before the next multiple of 8
    check(_lib.lib().unflow_planes_from_f32(xp, ldx, _lib.cl(B * H * W), C, fill, _lib.planes_of(out_pl), stream()),
          "planes_from_f32")


def conv_fwd(x, w, w_pl, bias, y, stride, leaky, planes_only=False):
    """conv2d_fwd on PTs; w_pl: the layer's TRANSPOSED weight planes [P, k*k, Cout, round8(Cin)] (or None).
    planes_only: y lives as operand planes alone (its fp32 tensor is not written)."""
    x, y = _pt(x), _pt(y)
    xp, ldx, B, H, W, Cin = nhwc(x.t)
    yp, ldy, _, Ho, Wo, Cout = nhwc(y.t)
    if planes_only:
        assert y.pl is not None
        yp = ptr(None)
    k = w.shape[0]
    assert tuple(w.shape) == (k, k, Cin, Cout) and w.is_contiguous() and (Ho, Wo) == out_hw(H, W, stride)
    wsp, wsn = _ws_pl(x.t.device, B, H, W, Cin, Cout, k, stride, _npl(x, y))
    check(_lib.lib().unflow_conv2d_fwd_pl(xp, ldx, x.planes(), ptr(w), _lib.planes_of(w_pl), ptr(bias), yp, ldy,
                                          y.planes(), B, H, W, Cin, Cout, k, stride, int(bool(leaky)), wsp, wsn,
                                          stream()), "conv2d_fwd_pl")


def _act_args(act_src, act_planes):
    """(fp32 pointer, ld, planes) of the activation whose sign gives the leaky-ReLU derivative: act_planes=True takes it
    from the first operand plane of act_src (a tensor without an fp32 copy), else from its fp32 values."""
    if act_src is None:
        return ptr(None), 0, None
    a = _pt(act_src)
    if act_planes:
        assert a.pl is not None
        return ptr(None), 0, a.planes()
    ap, lda = nhwc(a.t)[:2]
    return ap, lda, None


def conv_bwd_data(dz, w, w_pl, dx, stride, accumulate=False, act_src=None, act_lo=0, act_hi=0, act_planes=False):
    """conv2d_bwd_data on PTs; w_pl: the DIRECT weight planes [P, k*k, Cin, round8(Cout)].  dx's planes receive the
    channels [act_lo, act_hi) (final after this call)."""
    dz, dx = _pt(dz), _pt(dx)
    dzp, lddz, B, Ho, Wo, Cout = nhwc(dz.t)
    dxp, lddx, _, H, W, Cin = nhwc(dx.t)
    k = w.shape[0]
    assert tuple(w.shape) == (k, k, Cin, Cout)
    ap, lda, apl = _act_args(act_src, act_planes)
    wsp, wsn = _ws_pl(dz.t.device, B, H, W, Cin, Cout, k, stride, _npl(dz, dx))
    check(_lib.lib().unflow_conv2d_bwd_data_pl(dzp, lddz, dz.planes(), ptr(w), _lib.planes_of(w_pl), dxp, lddx,
                                               dx.planes(), act_lo, act_hi, B, H, W, Cin, Cout, k, stride,
                                               int(bool(accumulate)), ap, lda, apl, act_lo, act_hi, wsp, wsn, stream()),
          "conv2d_bwd_data_pl")


def conv_bwd_filter(x, dz, dw, stride, x_fp32=True):
    """x_fp32 = False: x lives as operand planes alone (its fp32 tensor is stale) — the library then gets no fp32 pointer and
    fails loudly if it cannot take the planes path instead of computing a gradient from stale values."""
    x, dz = _pt(x), _pt(dz)
    xp, ldx, B, H, W, Cin = nhwc(x.t)
    if not x_fp32:
        assert x.pl is not None
        xp = ptr(None)
    dzp, lddz, _, Ho, Wo, Cout = nhwc(dz.t)
    k = dw.shape[0]
    assert tuple(dw.shape) == (k, k, Cin, Cout) and dw.is_contiguous()
    wsp, wsn = _ws_pl(x.t.device, B, H, W, Cin, Cout, k, stride, _npl(x, dz))
    check(_lib.lib().unflow_conv2d_bwd_filter_pl(xp, ldx, x.planes(), dzp, lddz, dz.planes(), ptr(dw), B, H,
                                                 W, Cin, Cout, k, stride, wsp, wsn, stream()), "conv2d_bwd_filter_pl")


def deconv_fwd(x, w, w_pl, bias, y, leaky, planes_only=False):
    """conv2d_transpose_fwd on PTs; w_pl: the DIRECT weight planes [P, 16, Cout, round8(Cin)]."""
    x, y = _pt(x), _pt(y)
    xp, ldx, B, H, W, Cin = nhwc(x.t)
    yp, ldy, _, Ho, Wo, Cout = nhwc(y.t)
    if planes_only:
        assert y.pl is not None
        yp = ptr(None)
    assert tuple(w.shape) == (4, 4, Cout, Cin) and (Ho, Wo) == (2 * H, 2 * W)
    wsp, wsn = _ws_pl(x.t.device, B, 2 * H, 2 * W, Cin, Cout, 4, 2, _npl(x, y))
    check(_lib.lib().unflow_conv2d_transpose_fwd_pl(xp, ldx, x.planes(), ptr(w), _lib.planes_of(w_pl), ptr(bias), yp,
                                                    ldy, y.planes(), B, H, W, Cin, Cout, int(bool(leaky)), wsp, wsn,
                                                    stream()), "conv2d_transpose_fwd_pl")


def deconv_bwd_data(dz, w, w_pl, dx, accumulate=False, act_src=None, act_lo=0, act_hi=0, act_planes=False):
    """conv2d_transpose_bwd_data on PTs; w_pl: the TRANSPOSED weight planes [P, 16, Cin, round8(Cout)]."""
    dz, dx = _pt(dz), _pt(dx)
    dzp, lddz, B, Ho, Wo, Cout = nhwc(dz.t)
    dxp, lddx, _, H, W, Cin = nhwc(dx.t)
    assert tuple(w.shape) == (4, 4, Cout, Cin) and (Ho, Wo) == (2 * H, 2 * W)
    ap, lda, apl = _act_args(act_src, act_planes)
    wsp, wsn = _ws_pl(dz.t.device, B, Ho, Wo, Cin, Cout, 4, 2, _npl(dz, dx))
    check(_lib.lib().unflow_conv2d_transpose_bwd_data_pl(dzp, lddz, dz.planes(), ptr(w), _lib.planes_of(w_pl), dxp,
                                                         lddx, dx.planes(), act_lo, act_hi, B, H, W, Cin, Cout,
                                                         int(bool(accumulate)), ap, lda, apl, act_lo, act_hi, wsp, wsn,
                                                         stream()),
          "conv2d_transpose_bwd_data_pl")


def deconv_bwd_filter(x, dz, dw, x_fp32=True):
    x, dz = _pt(x), _pt(dz)
    xp, ldx, B, H, W, Cin = nhwc(x.t)
    if not x_fp32:
        assert x.pl is not None
        xp = ptr(None)
    dzp, lddz, _, Ho, Wo, Cout = nhwc(dz.t)
    assert tuple(dw.shape) == (4, 4, Cout, Cin)
    wsp, wsn = _ws_pl(x.t.device, B, Ho, Wo, Cin, Cout, 4, 2, _npl(x, dz))
    check(_lib.lib().unflow_conv2d_transpose_bwd_filter_pl(xp, ldx, x.planes(), dzp, lddz, dz.planes(),
                                                           ptr(dw), B, H, W, Cin, Cout, wsp, wsn, stream()),
          "conv2d_transpose_bwd_filter_pl")


def flow_wgrad_batched(jobs):
    """Filter gradients of all Cout = 2 layers of a decoder in one batch (csrc/conv_igemm.hip: flow_wgrad_batched_kernel).
    jobs: [(kind, x, dz, dw)] with kind 'conv' (flowN: x [B,H,W,Cin], dz [B,H,W,2], dw [3,3,Cin,2]) or 'deconv' (flowN_upM:
    x [B,H,W,2], dz [B,2H,2W,2], dw [4,4,2,2]); x / dz may be PTs.  Falls back to the per-layer calls when a level has no
    strip form."""
    import ctypes
    n = len(jobs)
    if n == 0:
        return
    xs = [_pt(j[1]).t for j in jobs]
    dzs = [_pt(j[2]).t for j in jobs]
    kinds = (ctypes.c_int * n)(*[0 if j[0] == 'conv' else 1 for j in jobs])
    dims = [nhwc(x) for x in xs]
    Bs = (ctypes.c_int * n)(*[d[2] for d in dims])
    Hs = (ctypes.c_int * n)(*[d[3] for d in dims])
    Ws = (ctypes.c_int * n)(*[d[4] for d in dims])
    Cs = (ctypes.c_int * n)(*[d[5] for d in dims])
    lib = _lib.lib()
    lib.unflow_flow_wgrad_batched_workspace_bytes.restype = ctypes.c_size_t
    need = lib.unflow_flow_wgrad_batched_workspace_bytes(n, kinds, Bs, Hs, Ws, Cs)
    if need == 0 or n > 16:
        for kind, x, dz, dw in jobs:
            if kind == 'conv':
                conv_bwd_filter(x, dz, dw, 1)
            else:
                deconv_bwd_filter(x, dz, dw)
        return
    for (kind, x, dz, dw), xt, dzt in zip(jobs, xs, dzs):
        assert dw.is_contiguous() and dzt.shape[-1] == 2
        assert tuple(dw.shape) == ((3, 3, xt.shape[-1], 2) if kind == 'conv' else (4, 4, 2, 2))
    ws = workspace(need, xs[0].device, slot=_WS_SLOT[0])
    check(lib.unflow_flow_wgrad_batched(n, kinds, (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs]),
                                        (ctypes.c_int * n)(*[x.stride(2) for x in xs]),
                                        (ctypes.c_void_p * n)(*[d.data_ptr() for d in dzs]),
                                        (ctypes.c_int * n)(*[d.stride(2) for d in dzs]),
                                        (ctypes.c_void_p * n)(*[j[3].data_ptr() for j in jobs]), Bs, Hs, Ws, Cs, ptr(ws),
                                        csz(ws.numel() * 4), stream()), "flow_wgrad_batched")
