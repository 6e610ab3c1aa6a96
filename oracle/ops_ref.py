"""ORACLE — test infrastructure only (see oracle/ops_ref.c header).

numpy-facing wrappers around oracle/_build/liboracle.so, the plain-C restatement
of the reference's four custom ops (ops/*_op.cu.cc) and of
src/e2eflow/core/image_warp.py.  Layouts are the reference's: correlation NCHW,
everything else NHWC, flow channel 0 = x/u.

Nothing under unflow_amd/ may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile ops_ref.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "ops_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_F)


class OracleError(ValueError):
    pass


_MSG = {-2: "Invalid correlation settings", -3: "kernel_size must be odd",
        -4: "Input height and width must be divisible by scale"}


def _check(st):
    if st != 0:
        raise OracleError(_MSG.get(st, "status %d" % st))


def correlation_out_shape(H, W, kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2):
    out = (ctypes.c_int * 3)()
    _check(lib().ref_correlation_out_shape(H, W, kernel_size, max_displacement, pad, stride_1, stride_2, out))
    return tuple(out)


def correlation(in0, in1, kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2):
    """ops.correlation (src/e2eflow/ops.py:69-70); NCHW in, NCHW out."""
    in0, p0 = _f(in0)
    in1, p1 = _f(in1)
    if in0.shape != in1.shape:
        raise OracleError("Input shapes have to be the same")
    B, C, H, W = in0.shape
    oc, oh, ow = correlation_out_shape(H, W, kernel_size, max_displacement, pad, stride_1, stride_2)
    out = np.empty((B, oc, oh, ow), np.float32)
    _check(lib().ref_correlation_fwd(p0, p1, out.ctypes.data_as(_F), B, C, H, W, kernel_size,
                                     max_displacement, pad, stride_1, stride_2))
    return out


def correlation_grad(dout, in0, in1, kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2):
    """_CorrelationGrad (src/e2eflow/ops.py:94-104) -> (grad0, grad1), NCHW."""
    dout, pd = _f(dout)
    in0, p0 = _f(in0)
    in1, p1 = _f(in1)
    B, C, H, W = in0.shape
    g0 = np.empty_like(in0)
    g1 = np.empty_like(in0)
    _check(lib().ref_correlation_bwd(pd, p0, p1, g0.ctypes.data_as(_F), g1.ctypes.data_as(_F), B, C, H, W,
                                     kernel_size, max_displacement, pad, stride_1, stride_2))
    return g0, g1


def backward_warp(images, flows):
    images, pi = _f(images)
    flows, pf = _f(flows)
    B, H, W, C = images.shape
    out = np.empty_like(images)
    lib().ref_backward_warp_fwd(pi, pf, out.ctypes.data_as(_F), B, H, W, C)
    return out


def backward_warp_indices(flows):
    flows, pf = _f(flows)
    B, H, W, _ = flows.shape
    out = np.empty((B, H, W, 2), np.int32)
    lib().ref_backward_warp_indices(pf, out.ctypes.data_as(_I), B, H, W)
    return out


def backward_warp_grad(dout, images, flows):
    dout, pd = _f(dout)
    images, pi = _f(images)
    flows, pf = _f(flows)
    B, H, W, C = images.shape
    out = np.empty_like(flows)
    lib().ref_backward_warp_bwd(pd, pi, pf, out.ctypes.data_as(_F), B, H, W, C)
    return out


def forward_warp(flows):
    flows, pf = _f(flows)
    B, H, W, _ = flows.shape
    out = np.empty((B, H, W, 1), np.float32)
    lib().ref_forward_warp_fwd(pf, out.ctypes.data_as(_F), B, H, W)
    return out


def forward_warp_ranges(flows):
    flows, pf = _f(flows)
    B, H, W, _ = flows.shape
    out = np.empty((B, H, W, 4), np.int32)
    lib().ref_forward_warp_ranges(pf, out.ctypes.data_as(_I), B, H, W)
    return out


def forward_warp_grad(dout, flows):
    dout, pd = _f(dout)
    flows, pf = _f(flows)
    B, H, W, _ = flows.shape
    out = np.empty_like(flows)
    lib().ref_forward_warp_bwd(pd, pf, out.ctypes.data_as(_F), B, H, W)
    return out


def downsample(images, scale=2):
    images, pi = _f(images)
    B, H, W, C = images.shape
    if scale <= 0 or H % scale or W % scale:
        raise OracleError(_MSG[-4])
    out = np.empty((B, H // scale, W // scale, C), np.float32)
    _check(lib().ref_downsample(pi, out.ctypes.data_as(_F), B, H, W, C, scale))
    return out


def image_warp(im, flow, return_indices=False):
    im, pi = _f(im)
    flow, pf = _f(flow)
    B, H, W, C = im.shape
    out = np.empty_like(im)
    idx = np.empty((B, H, W, 4), np.int32) if return_indices else None
    lib().ref_image_warp_fwd(pi, pf, out.ctypes.data_as(_F),
                             idx.ctypes.data_as(_I) if return_indices else None, B, H, W, C)
    return (out, idx) if return_indices else out


def image_warp_grad(dwarp, im, flow, need_im_grad=True):
    dwarp, pd = _f(dwarp)
    im, pi = _f(im)
    flow, pf = _f(flow)
    B, H, W, C = im.shape
    d_im = np.empty_like(im) if need_im_grad else None
    d_flow = np.empty_like(flow)
    lib().ref_image_warp_bwd(pd, pi, pf, d_im.ctypes.data_as(_F) if need_im_grad else None,
                             d_flow.ctypes.data_as(_F), B, H, W, C)
    return d_im, d_flow
