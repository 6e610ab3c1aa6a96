"""Operator boundary — mirror of the reference's src/e2eflow/ops.py.

Same four callables, positional order, kwarg names and defaults (ops.py:69-75):
    correlation(first, second, **kwargs)   NCHW in / NCHW out  (output 0 of the reference op)
    backward_warp(images, flows)           NHWC, flow channel 0 = x
    downsample(images, scale=2)            NHWC, not differentiable (ops.py:107)
    forward_warp(flows)                    NHWC -> [B,H,W,1]
and the same three gradients (ops.py:80-104), wired through torch.autograd instead of
tf.RegisterGradient.  Tensors are PyTorch-ROCm CUDA tensors used purely as device buffers; every
op is a call into libunflow_hip.so (include/unflow_hip.h).  No CPU path exists.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr, stream

OP_NAMES = ['backward_warp', 'downsample', 'correlation', 'forward_warp']  # ops.py:11

_CORR_DEFAULTS = dict(kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2)  # correlation_op.cc:136-140

_ws = {}
_ws_retired = []      # outgrown buffers stay allocated: a captured hipGraph may still hold their address


def workspace(nbytes, device, slot=0):
    """Grow-only per-device scratch buffer (the library never allocates).  A buffer that is outgrown is RETIRED, not freed:
    launches captured into a hipGraph keep replaying with the address they were captured with (a second, larger engine in
    the same process — Trainer.eval's beside a small-crop training engine — must not pull the scratch from under them)."""
    key = (device.index, slot)
    t = _ws.get(key)
    if t is None or t.numel() * 4 < nbytes:
        if t is not None:
            _ws_retired.append(t)
        t = torch.empty((int(nbytes) + 3) // 4 + 64, dtype=torch.float32, device=device)
        _ws[key] = t
    return t


def _dev(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise TypeError("%s must be a CUDA (ROCm) tensor: the ops have GPU kernels only, "
                        "like the reference (REGISTER_KERNEL_BUILDER(... DEVICE_GPU))" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (the reference ops are float-only)" % name)
    return t.contiguous()


def _corr_attrs(kwargs):
    a = dict(_CORR_DEFAULTS)
    for k in kwargs:
        if k not in a:
            raise TypeError("correlation() got an unexpected keyword argument '%s'" % k)
    a.update(kwargs)
    return a


def correlation_out_shape(H, W, **kwargs):
    a = _corr_attrs(kwargs)
    out = (ctypes.c_int * 3)()
    check(_lib.lib().unflow_correlation_out_shape(H, W, a['kernel_size'], a['max_displacement'], a['pad'],
                                                  a['stride_1'], a['stride_2'], out), "correlation")
    return tuple(out)


def _corr_args(a):
    return (a['kernel_size'], a['max_displacement'], a['pad'], a['stride_1'], a['stride_2'])


def _correlation_fwd(first, second, a):
    if first.shape != second.shape:
        raise _lib.UnflowError(-5, "correlation")   # correlation_op.cc:47-48
    B, C, H, W = first.shape
    oc, oh, ow = correlation_out_shape(H, W, **a)
    out = torch.empty((B, oc, oh, ow), dtype=torch.float32, device=first.device)
    L = _lib.lib()
    nbytes = L.unflow_correlation_workspace_bytes(B, C, H, W, *_corr_args(a))
    ws = workspace(nbytes, first.device)
    check(L.unflow_correlation_fwd(ptr(first), ptr(second), ptr(out), B, C, H, W, *_corr_args(a), ptr(ws),
                                   _lib.csz(ws.numel() * 4), stream()), "correlation")
    return out


def correlation_grad(in_grad, first, second, **kwargs):
    """_correlation_module.correlation_grad (ops.py:96-103) minus the padded_0/1 inputs, which the
    replacement recomputes on the fly."""
    a = _corr_attrs(kwargs)
    in_grad, first, second = _dev(in_grad, 'in_grad'), _dev(first, 'first'), _dev(second, 'second')
    B, C, H, W = first.shape
    g0 = torch.empty_like(first)
    g1 = torch.empty_like(first)
    L = _lib.lib()
    nbytes = L.unflow_correlation_workspace_bytes(B, C, H, W, *_corr_args(a))
    ws = workspace(nbytes, first.device)
    check(L.unflow_correlation_bwd(ptr(in_grad), ptr(first), ptr(second), ptr(g0), ptr(g1), B, C, H, W,
                                   *_corr_args(a), ptr(ws), _lib.csz(ws.numel() * 4), stream()), "correlation_grad")
    return g0, g1


class _Correlation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, first, second, attrs):
        ctx.save_for_backward(first, second)
        ctx.attrs = attrs
        return _correlation_fwd(first, second, attrs)

    @staticmethod
    def backward(ctx, g):
        first, second = ctx.saved_tensors
        g0, g1 = correlation_grad(g, first, second, **ctx.attrs)
        return g0, g1, None


def correlation(first, second, **kwargs):
    """ops.correlation (ops.py:69-70)."""
    return _Correlation.apply(_dev(first, 'first'), _dev(second, 'second'), _corr_attrs(kwargs))


def backward_warp_grad(grad, images, flows):
    grad, images, flows = _dev(grad, 'grad'), _dev(images, 'images'), _dev(flows, 'flows')
    B, H, W, C = images.shape
    out = torch.empty_like(flows)
    check(_lib.lib().unflow_backward_warp_bwd(ptr(grad), ptr(images), ptr(flows), ptr(out), B, H, W, C, stream()),
          "backward_warp_grad")
    return out


class _BackwardWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, flows):
        B, H, W, C = images.shape
        if flows.shape != (B, H, W, 2):
            raise ValueError("flows must be [B,H,W,2] matching images")
        ctx.save_for_backward(images, flows)
        out = torch.empty_like(images)
        check(_lib.lib().unflow_backward_warp_fwd(ptr(images), ptr(flows), ptr(out), B, H, W, C, stream()),
              "backward_warp")
        return out

    @staticmethod
    def backward(ctx, g):
        images, flows = ctx.saved_tensors
        return None, backward_warp_grad(g, images, flows)   # ops.py:80-84: [None, grad0]


def backward_warp(images, flows):
    if images.numel() == 0:      # empty batch: nothing to launch (the reference's kernel launch would cover 0 elements)
        return torch.empty_like(images)
    return _BackwardWarp.apply(_dev(images, 'images'), _dev(flows, 'flows'))


def backward_warp_indices(flows):
    flows = _dev(flows, 'flows')
    B, H, W, _ = flows.shape
    out = torch.empty((B, H, W, 2), dtype=torch.int32, device=flows.device)
    check(_lib.lib().unflow_backward_warp_indices(ptr(flows), ptr(out), B, H, W, stream()))
    return out


def forward_warp_grad(grad, flows):
    grad, flows = _dev(grad, 'grad'), _dev(flows, 'flows')
    B, H, W, _ = flows.shape
    out = torch.empty_like(flows)
    check(_lib.lib().unflow_forward_warp_bwd(ptr(grad), ptr(flows), ptr(out), B, H, W, stream()), "forward_warp_grad")
    return out


class _ForwardWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flows, deterministic):
        B, H, W, c = flows.shape
        if c != 2:
            raise ValueError("flows must have 2 channels")
        ctx.save_for_backward(flows)
        out = torch.empty((B, H, W, 1), dtype=torch.float32, device=flows.device)
        # full workspace: far-flung sources are binned by target tile instead of scattered with global atomics (either mode)
        ws = workspace(_lib.lib().unflow_forward_warp_workspace_bytes(B, H, W, int(bool(deterministic))), flows.device)
        check(_lib.lib().unflow_forward_warp_fwd(ptr(flows), ptr(out), B, H, W, int(bool(deterministic)), ptr(ws),
                                                 _lib.csz(0 if ws is None else ws.numel() * 4), stream()),
              "forward_warp")
        return out

    @staticmethod
    def backward(ctx, g):
        (flows,) = ctx.saved_tensors
        return forward_warp_grad(g, flows), None


def forward_warp(flows, deterministic=True):
    """ops.forward_warp (ops.py:75).  deterministic=False reproduces the reference's float-atomic scatter."""
    if flows.numel() == 0:
        return flows.new_empty(tuple(flows.shape[:3]) + (1,))
    return _ForwardWarp.apply(_dev(flows, 'flows'), deterministic)


def forward_warp_ranges(flows):
    flows = _dev(flows, 'flows')
    B, H, W, _ = flows.shape
    out = torch.empty((B, H, W, 4), dtype=torch.int32, device=flows.device)
    check(_lib.lib().unflow_forward_warp_ranges(ptr(flows), ptr(out), B, H, W, stream()))
    return out


def downsample(images, scale=2):
    """ops.downsample (ops.py:74); NotDifferentiable (ops.py:107)."""
    images = _dev(images.detach(), 'images')
    B, H, W, C = images.shape
    if scale <= 0 or H % scale or W % scale:
        raise _lib.UnflowError(-4, "downsample")
    if images.numel() == 0:
        return images.new_empty((B, H // scale, W // scale, C))
    out = torch.empty((B, H // scale, W // scale, C), dtype=torch.float32, device=images.device)
    check(_lib.lib().unflow_downsample_fwd(ptr(images), ptr(out), B, H, W, C, int(scale), stream()), "downsample")
    return out
