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
