"""ORACLE — test infrastructure only.  Nothing under unflow_amd/ may import this.

torch-CPU restatement of the reference's Python/TensorFlow-1 graph for the
training step: FlowNetC/S (src/e2eflow/core/flownet.py), image_warp
(core/image_warp.py), the proxy losses (core/losses.py), the step assembly
(core/unsupervised.py), EPE (core/flow_util.py:98-123) and TF-form Adam
(core/train.py:151-152).  Gradients come from torch autograd over this
restatement; the custom ops go through oracle/ops_ref.c.

PARITY UNPINNED for everything whose arithmetic lives in (absent) TensorFlow:
conv / conv_transpose SAME geometry, resize_bilinear, rgb_to_grayscale, pow,
Adam — and for the augmentation (core/augment.py, core/spatial_transformer.py:
tf.linspace, tf.matmul, tf.pow), which the reference does not test at all.  They follow TF1's documented semantics and are
cross-checked fp32-vs-fp64 in tests; the reference's own tests only pin
image_warp values, the 1st-order stencil/masks, create_outgoing_mask and
gradient_loss~0 (tests/golden/ref_kats.json).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import ops_ref

FLOW_SCALE = 5.0          # flownet.py:11
DISOCC_THRESH = 0.8       # losses.py:9
CHANNEL_MEAN = [104.920005, 110.1753, 114.785955]  # core/input.py:45
LOSSES = ['occ', 'sym', 'fb', 'grad', 'ternary', 'photo', 'smooth_1st', 'smooth_2nd']  # unsupervised.py:15


# ----------------------------------------------------------------------------
# TF layers
# ----------------------------------------------------------------------------
def same_pads(in_size, k, s):
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return total // 2, total - total // 2


LEAKY_HOOK = None   # tests may install a callable here (tests/parity_util.py: branch-aligned derivative at the kink)


def leaky_relu(x):
    # flownet.py:84-86  tf.maximum(0.1 * x, x)
    if LEAKY_HOOK is not None:
        return LEAKY_HOOK(x)
    return torch.maximum(0.1 * x, x)


def conv2d(x, w, b, stride=1, act=True):
    """slim.conv2d NCHW, SAME.  w: HWIO [k,k,Cin,Cout]."""
    k = w.shape[0]
    pt, pb = same_pads(x.shape[2], k, stride)
    pl, pr = same_pads(x.shape[3], k, stride)
    y = F.conv2d(F.pad(x, (pl, pr, pt, pb)), w.permute(3, 2, 0, 1), b, stride=stride)
    return leaky_relu(y) if act else y


def conv2d_transpose(x, w, b, act=True):
    """slim.conv2d_transpose k=4 stride=2 SAME.  w: [k,k,Cout,Cin] (TF layout)."""
    assert w.shape[0] == 4
    y = F.conv_transpose2d(x, w.permute(3, 2, 0, 1), b, stride=2, padding=1)
    return leaky_relu(y) if act else y


class _Correlation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, attrs):
        ctx.save_for_backward(a, b)
        ctx.attrs = attrs
        return torch.from_numpy(ops_ref.correlation(a.detach().numpy(), b.detach().numpy(), **attrs))

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g0, g1 = ops_ref.correlation_grad(g.contiguous().numpy(), a.detach().numpy(), b.detach().numpy(), **ctx.attrs)
        return torch.from_numpy(g0), torch.from_numpy(g1), None


def correlation(a, b, **attrs):
    if a.dtype == torch.float64:   # fp64 shadow path: dense torch restatement (k=1,s1=1 only)
        return correlation_dense(a, b, **attrs)
    return _Correlation.apply(a, b, attrs)


def correlation_dense(a, b, kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2):
    """Vectorised torch restatement (any dtype) for kernel_size=1, stride_1=1, pad==max_displacement —
    used to cross-check ops_ref.c and as the fp64 shadow."""
    assert kernel_size == 1 and stride_1 == 1 and pad == max_displacement
    B, C, H, W = a.shape
    r = max_displacement // stride_2
    bp = F.pad(b, (pad, pad, pad, pad))
    outs = []
    for p in range(-r, r + 1):
        for o in range(-r, r + 1):
            dy, dx = pad + p * stride_2, pad + o * stride_2
            outs.append((a * bp[:, :, dy:dy + H, dx:dx + W]).sum(1) / C)
    return torch.stack(outs, 1)


# ----------------------------------------------------------------------------
# parameters (TF variable layouts)
# ----------------------------------------------------------------------------
def _vs_init(shape, fan_in, gen):
    # layers.variance_scaling_initializer(): factor 2, FAN_IN, truncated normal, stddev sqrt(1.3*2/fan_in)
    std = math.sqrt(1.3 * 2.0 / fan_in)
    t = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std, generator=gen)
    return t


def channel_mult(name):
    """flownet.py:22-23: upper-case nets are full width, lower-case ones 3/8 width."""
    return 1 if name in ('C', 'S') else 3 / 8


def flownet_layer_specs(spec='C', in_channels=6, full_res=False):
    """(name, kind, k, cin, cout, stride, act) in the reference's variable order; spec in 'C', 'S' (full width) or
    'c', 's' (3/8 width, flownet.py:22-23); full_res adds the 'full_res/' variables of flownet.py:133-153."""
    m = channel_mult(spec)
    c = lambda x: int(x * m)                                                # noqa: E731
    L = []
    if spec in ('C', 'c'):
        L += [('flownet_c_features/conv1', 'conv', 7, 3, c(64), 2, True),
              ('flownet_c_features/conv2', 'conv', 5, c(64), c(128), 2, True),
              ('flownet_c_features/conv3', 'conv', 5, c(128), c(256), 2, True)]
        pre = 'flownet_c/'
        L += [(pre + 'conv_redir', 'conv', 1, c(256), c(32), 1, True),
              (pre + 'conv3_1', 'conv', 3, c(32) + 441, c(256), 1, True)]
    else:
        pre = 'flownet_s/'
        L += [(pre + 'conv1', 'conv', 7, in_channels, c(64), 2, True),
              (pre + 'conv2', 'conv', 5, c(64), c(128), 2, True),
              (pre + 'conv3', 'conv', 5, c(128), c(256), 2, True),
              (pre + 'conv3_1', 'conv', 3, c(256), c(256), 1, True)]
    cat5, cat4, cat3, cat2 = c(512) + c(512) + 2, c(512) + c(256) + 2, c(256) + c(128) + 2, c(128) + c(64) + 2
    L += [(pre + 'conv4', 'conv', 3, c(256), c(512), 2, True), (pre + 'conv4_1', 'conv', 3, c(512), c(512), 1, True),
          (pre + 'conv5', 'conv', 3, c(512), c(512), 2, True), (pre + 'conv5_1', 'conv', 3, c(512), c(512), 1, True),
          (pre + 'conv6', 'conv', 3, c(512), c(1024), 2, True), (pre + 'conv6_1', 'conv', 3, c(1024), c(1024), 1, True),
          (pre + 'flow6', 'conv', 3, c(1024), 2, 1, False),
          (pre + 'deconv5', 'deconv', 4, c(1024), c(512), 2, True), (pre + 'flow6_up5', 'deconv', 4, 2, 2, 2, False),
          (pre + 'flow5', 'conv', 3, cat5, 2, 1, False),
          (pre + 'deconv4', 'deconv', 4, cat5, c(256), 2, True), (pre + 'flow5_up4', 'deconv', 4, 2, 2, 2, False),
          (pre + 'flow4', 'conv', 3, cat4, 2, 1, False),
          (pre + 'deconv3', 'deconv', 4, cat4, c(128), 2, True), (pre + 'flow4_up3', 'deconv', 4, 2, 2, 2, False),
          (pre + 'flow3', 'conv', 3, cat3, 2, 1, False),
          (pre + 'deconv2', 'deconv', 4, cat3, c(64), 2, True), (pre + 'flow3_up2', 'deconv', 4, 2, 2, 2, False),
          (pre + 'flow2', 'conv', 3, cat2, 2, 1, False)]
    if full_res:
        fr = pre + 'full_res/'
        cat1 = c(64) + c(32) + 2
        cat0 = in_channels + c(16) + 2
        L += [(fr + 'deconv1', 'deconv', 4, cat2, c(32), 2, True), (fr + 'flow2_up1', 'deconv', 4, 2, 2, 2, False),
              (fr + 'flow1', 'conv', 3, cat1, 2, 1, False),
              (fr + 'deconv0', 'deconv', 4, cat1, c(16), 2, True), (fr + 'flow1_up0', 'deconv', 4, 2, 2, 2, False),
              (fr + 'flow0', 'conv', 3, cat0, 2, 1, False)]
    return L


def init_params(spec='C', seed=0, in_channels=6):
    gen = torch.Generator().manual_seed(seed)
    P = OrderedDict()
    for name, kind, k, cin, cout, stride, act in flownet_layer_specs(spec, in_channels):
        if kind == 'conv':
            P[name + '/weights'] = _vs_init((k, k, cin, cout), k * k * cin, gen)
        else:  # conv2d_transpose weights [k,k,out,in]; slim's fan_in for this shape = k*k*out
            P[name + '/weights'] = _vs_init((k, k, cout, cin), k * k * cout, gen)
        P[name + '/biases'] = torch.zeros(cout)
    return P


# ----------------------------------------------------------------------------
# networks (flownet.py)
# ----------------------------------------------------------------------------
def _cv(P, pre, name, x, stride=1, act=True):
    return conv2d(x, P[pre + name + '/weights'], P[pre + name + '/biases'], stride, act)


def _dc(P, pre, name, x, act=True):
    return conv2d_transpose(x, P[pre + name + '/weights'], P[pre + name + '/biases'], act)


def flownet_upconv(P, pre, conv6_1, conv5_1, conv4_1, conv3_1, conv2, conv1=None, inputs=None, full_res=False):
    """_flownet_upconv, flownet.py:89-155."""
    flow6 = _cv(P, pre, 'flow6', conv6_1, act=False)
    deconv5 = _dc(P, pre, 'deconv5', conv6_1)
    flow6_up5 = _dc(P, pre, 'flow6_up5', flow6, act=False)
    concat5 = torch.cat([conv5_1, deconv5, flow6_up5], 1)
    flow5 = _cv(P, pre, 'flow5', concat5, act=False)
    deconv4 = _dc(P, pre, 'deconv4', concat5)
    flow5_up4 = _dc(P, pre, 'flow5_up4', flow5, act=False)
    concat4 = torch.cat([conv4_1, deconv4, flow5_up4], 1)
    flow4 = _cv(P, pre, 'flow4', concat4, act=False)
    deconv3 = _dc(P, pre, 'deconv3', concat4)
    flow4_up3 = _dc(P, pre, 'flow4_up3', flow4, act=False)
    concat3 = torch.cat([conv3_1, deconv3, flow4_up3], 1)
    flow3 = _cv(P, pre, 'flow3', concat3, act=False)
    deconv2 = _dc(P, pre, 'deconv2', concat3)
    flow3_up2 = _dc(P, pre, 'flow3_up2', flow3, act=False)
    concat2 = torch.cat([conv2, deconv2, flow3_up2], 1)
    flow2 = _cv(P, pre, 'flow2', concat2, act=False)
    flows = [flow2, flow3, flow4, flow5, flow6]
    if full_res:                                                            # flownet.py:133-153
        fr = pre + 'full_res/'
        deconv1 = _dc(P, fr, 'deconv1', concat2)
        flow2_up1 = _dc(P, fr, 'flow2_up1', flow2, act=False)
        concat1 = torch.cat([conv1, deconv1, flow2_up1], 1)
        flow1 = _cv(P, fr, 'flow1', concat1, act=False)
        deconv0 = _dc(P, fr, 'deconv0', concat1)
        flow1_up0 = _dc(P, fr, 'flow1_up0', flow1, act=False)
        concat0 = torch.cat([inputs, deconv0, flow1_up0], 1)
        flow0 = _cv(P, fr, 'flow0', concat0, act=False)
        flows = [flow0, flow1] + flows
    return flows


def flownet_c_features(P, im_nhwc):
    """flownet.py:195-206."""
    pre = 'flownet_c_features/'
    x = im_nhwc.permute(0, 3, 1, 2)
    conv1 = _cv(P, pre, 'conv1', x, 2)
    conv2 = _cv(P, pre, 'conv2', conv1, 2)
    conv3 = _cv(P, pre, 'conv3', conv2, 2)
    return conv1, conv2, conv3


def flownet_c(P, conv3_a, conv3_b, conv2_a, return_internals=False):
    """flownet.py:209-237; returns NHWC flows [flow2..flow6]."""
    pre = 'flownet_c/'
    corr = correlation(conv3_a, conv3_b, pad=20, kernel_size=1, max_displacement=20, stride_1=1, stride_2=2)
    conv_redir = _cv(P, pre, 'conv_redir', conv3_a, 1)
    conv3_1 = _cv(P, pre, 'conv3_1', torch.cat([conv_redir, corr], 1), 1)
    conv4 = _cv(P, pre, 'conv4', conv3_1, 2)
    conv4_1 = _cv(P, pre, 'conv4_1', conv4, 1)
    conv5 = _cv(P, pre, 'conv5', conv4_1, 2)
    conv5_1 = _cv(P, pre, 'conv5_1', conv5, 1)
    conv6 = _cv(P, pre, 'conv6', conv5_1, 2)
    conv6_1 = _cv(P, pre, 'conv6_1', conv6, 1)
    res = flownet_upconv(P, pre, conv6_1, conv5_1, conv4_1, conv3_1, conv2_a)
    flows = [t.permute(0, 2, 3, 1) for t in res]
    if return_internals:
        return flows, dict(corr=corr, conv_redir=conv_redir, conv3_1=conv3_1, conv4=conv4, conv4_1=conv4_1,
                           conv5=conv5, conv5_1=conv5_1, conv6=conv6, conv6_1=conv6_1)
    return flows


def flownet_s(P, inputs_nhwc, pre='flownet_s/', full_res=False):
    """flownet.py:166-192 (the channel multiplier is implied by the shapes of P)."""
    x = inputs_nhwc.permute(0, 3, 1, 2)
    conv1 = _cv(P, pre, 'conv1', x, 2)
    conv2 = _cv(P, pre, 'conv2', conv1, 2)
    conv3 = _cv(P, pre, 'conv3', conv2, 2)
    conv3_1 = _cv(P, pre, 'conv3_1', conv3, 1)
    conv4 = _cv(P, pre, 'conv4', conv3_1, 2)
    conv4_1 = _cv(P, pre, 'conv4_1', conv4, 1)
    conv5 = _cv(P, pre, 'conv5', conv4_1, 2)
    conv5_1 = _cv(P, pre, 'conv5_1', conv5, 1)
    conv6 = _cv(P, pre, 'conv6', conv5_1, 2)
    conv6_1 = _cv(P, pre, 'conv6_1', conv6, 1)
    res = flownet_upconv(P, pre, conv6_1, conv5_1, conv4_1, conv3_1, conv2, conv1, x, full_res=full_res)
    return [t.permute(0, 2, 3, 1) for t in res]


def flownet(P, im1, im2, flownet_spec='C', backward_flow=False, train_all=False, full_resolution=False):
    """flownet.py:14-81.  'C', 'S', 'c', 's' and stacked specs ('CS', 'CSS', 'SS' ...): every later net is
    a FlowNetS on [im1, im2, flow*20 upsampled, warp(im2, flow), |warp - im1|] of the previous net's finest flow,
    with stop_gradient on flow/warp/diff unless train_all (:46-57).  Variable scopes: first net 'flownet_c*/' or
    'flownet_s/', net i >= 1 'stack_<i>_flownet/flownet_s/' (:72-77)."""
    H, W = im1.shape[1:3]
    flows_fw, flows_bw = [], []
    for i, name in enumerate(flownet_spec):
        assert name in ('C', 'S', 'c', 's')
        scope = '' if i == 0 else 'stack_%d_flownet/' % i
        full_res = full_resolution and i == len(flownet_spec) - 1           # flownet.py:24
        if name in ('C', 'c'):
            assert not full_res, "flownet_c passes neither conv1 nor inputs to _flownet_upconv (flownet.py:231-233)"
            assert i == 0, 'FlowNetS must be used for refinement networks'
            _, conv2_a, conv3_a = flownet_c_features(P, im1)
            _, conv2_b, conv3_b = flownet_c_features(P, im2)
            flows_fw.append(flownet_c(P, conv3_a, conv3_b, conv2_a))
            if backward_flow:
                flows_bw.append(flownet_c(P, conv3_b, conv3_a, conv2_b))
        else:
            def _s(a, b, flow):
                if flow is not None:
                    flow = resize_bilinear_tf1(flow, H, W) * 4 * FLOW_SCALE
                    warp = image_warp(b, flow)
                    diff = torch.abs(warp - a)
                    if not train_all:
                        flow, warp, diff = flow.detach(), warp.detach(), diff.detach()
                    inputs = torch.cat([a, b, flow, warp, diff], 3)
                else:
                    inputs = torch.cat([a, b], 3)
                return flownet_s(P, inputs, pre=scope + 'flownet_s/', full_res=full_res)
            stacked = len(flows_fw) > 0
            flows_fw.append(_s(im1, im2, flows_fw[-1][0] if stacked else None))
            if backward_flow:
                flows_bw.append(_s(im2, im1, flows_bw[-1][0] if stacked else None))
    if backward_flow:
        return flows_fw, flows_bw
    return flows_fw


def init_params_spec(flownet_spec='C', seed=0, full_res=False):
    """Variables of a (possibly stacked) spec with the reference's scope names."""
    gen = torch.Generator().manual_seed(seed)
    P = OrderedDict()
    for i, name in enumerate(flownet_spec):
        scope = '' if i == 0 else 'stack_%d_flownet/' % i
        fr = full_res and i == len(flownet_spec) - 1
        for lname, kind, k, cin, cout, stride, act in flownet_layer_specs(name, 14 if i > 0 else 6, fr):
            lname = scope + lname
            if kind == 'conv':
                P[lname + '/weights'] = _vs_init((k, k, cin, cout), k * k * cin, gen)
            else:
                P[lname + '/weights'] = _vs_init((k, k, cout, cin), k * k * cout, gen)
            P[lname + '/biases'] = torch.zeros(cout)
    return P


# ----------------------------------------------------------------------------
# image_warp (image_warp.py:4-76), differentiable in torch
# ----------------------------------------------------------------------------
def image_warp(im, flow):
    B, H, W, C = im.shape
    im_flat = im.reshape(-1, C)
    flow_flat = flow.reshape(-1, 2)
    ffloor = torch.floor(flow_flat)
    fl = ffloor.detach().to(torch.int64)
    bw = flow_flat - ffloor.detach()   # floor has zero gradient in TF
    pos_x = torch.arange(W).repeat(H * B)
    pos_y = torch.arange(H).unsqueeze(1).repeat(1, W).reshape(-1).repeat(B)
    xw, yw = bw[:, 0], bw[:, 1]
    wa = ((1 - xw) * (1 - yw)).unsqueeze(1)
    wb = ((1 - xw) * yw).unsqueeze(1)
    wc = (xw * (1 - yw)).unsqueeze(1)
    wd = (xw * yw).unsqueeze(1)
    x0 = pos_x + fl[:, 0]
    y0 = pos_y + fl[:, 1]
    x1 = (x0 + 1).clamp(0, W - 1)
    y1 = (y0 + 1).clamp(0, H - 1)
    x0 = x0.clamp(0, W - 1)
    y0 = y0.clamp(0, H - 1)
    base = (torch.arange(B) * (W * H)).unsqueeze(1).repeat(1, W * H).reshape(-1)
    Ia = im_flat[base + y0 * W + x0]
    Ib = im_flat[base + y1 * W + x0]
    Ic = im_flat[base + y0 * W + x1]
    Id = im_flat[base + y1 * W + x1]
    return (((wa * Ia + wb * Ib) + wc * Ic) + wd * Id).reshape(B, H, W, C)


class _ForwardWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow):
        ctx.save_for_backward(flow)
        return torch.from_numpy(ops_ref.forward_warp(flow.detach().numpy()))

    @staticmethod
    def backward(ctx, g):
        (flow,) = ctx.saved_tensors
        return torch.from_numpy(ops_ref.forward_warp_grad(g.contiguous().numpy(), flow.detach().numpy()))


def forward_warp(flow):
    return _ForwardWarp.apply(flow.float()).to(flow.dtype)


def downsample(t, num):
    """core/util.py:21-26 -> ops.downsample (box mean); pure-torch equivalent, any dtype."""
    B, H, W, C = t.shape
    assert H % num == 0 and W % num == 0
    return t.reshape(B, H // num, num, W // num, num, C).sum(4).sum(2) / float(num * num)


def resize_bilinear_tf1(x, out_h, out_w):
    """tf.image.resize_bilinear, TF1 legacy, align_corners=False (no half-pixel centres). NHWC."""
    B, H, W, C = x.shape

    def axis(in_s, out_s):
        src = torch.arange(out_s, dtype=x.dtype) * (in_s / out_s)
        lo = torch.floor(src)
        hi = torch.clamp(lo + 1, max=in_s - 1)
        return lo.long(), hi.long(), (src - lo)

    ylo, yhi, yl = axis(H, out_h)
    xlo, xhi, xl = axis(W, out_w)
    top = x[:, ylo]
    bot = x[:, yhi]
    xl_ = xl.view(1, 1, -1, 1)
    t = top[:, :, xlo] + (top[:, :, xhi] - top[:, :, xlo]) * xl_
    b = bot[:, :, xlo] + (bot[:, :, xhi] - bot[:, :, xlo]) * xl_
    return t + (b - t) * yl.view(1, -1, 1, 1)


# ----------------------------------------------------------------------------
# losses (losses.py)
# ----------------------------------------------------------------------------
def length_sq(x):
    return (x * x).sum(3, keepdim=True)


def create_mask(tensor, paddings):
    """losses.py:325-335: ones interior, zero `paddings` = [[top,bottom],[left,right]]."""
    B, H, W, _ = tensor.shape
    inner = torch.ones(H - paddings[0][0] - paddings[0][1], W - paddings[1][0] - paddings[1][1], dtype=tensor.dtype)
    m = F.pad(inner, (paddings[1][0], paddings[1][1], paddings[0][0], paddings[0][1]))
    return m.view(1, H, W, 1).repeat(B, 1, 1, 1)


def create_border_mask(tensor, border_ratio=0.1):
    """losses.py:338-344."""
    _, H, W, _ = tensor.shape
    sz = int(math.ceil(np.float32(min(H, W)) * np.float32(border_ratio)))
    return create_mask(tensor, [[sz, sz], [sz, sz]])


def create_outgoing_mask(flow):
    """losses.py:347-366."""
    B, H, W, _ = flow.shape
    gx = torch.arange(W, dtype=flow.dtype).view(1, 1, W)
    gy = torch.arange(H, dtype=flow.dtype).view(1, H, 1)
    px = gx + flow[..., 0]
    py = gy + flow[..., 1]
    inside = (px <= W - 1) & (px >= 0) & (py <= H - 1) & (py >= 0)
    return inside.to(flow.dtype).unsqueeze(3)


def charbonnier_loss(x, mask=None, alpha=0.45, beta=1.0, epsilon=0.001, truncate=None):
    """losses.py:298-322; normaliser = number of elements of x (not the mask sum)."""
    norm = float(x.numel())
    err = torch.pow((x * beta) ** 2 + epsilon ** 2, alpha)
    if mask is not None:
        err = mask * err
    if truncate is not None:
        err = torch.clamp(err, max=truncate)
    return err.sum() / norm


def _stencil(x_nhwc_1ch, filters):
    """tf.nn.conv2d(x, w, SAME) with 3x3 filters (list of 3x3 lists) on a 1-channel NHWC tensor."""
    w = torch.tensor(filters, dtype=x_nhwc_1ch.dtype).unsqueeze(1)  # [n,1,3,3]
    y = F.conv2d(x_nhwc_1ch.permute(0, 3, 1, 2), w, padding=1)
    return y.permute(0, 2, 3, 1)


def rgb_to_grayscale(im):
    w = torch.tensor([0.2989, 0.5870, 0.1140], dtype=im.dtype)
    return (im * w).sum(3, keepdim=True)


def ternary_loss(im1, im2_warped, mask, max_distance=1):
    """losses.py:90-122 (census / ternary)."""
    ps = 2 * max_distance + 1
    P = ps * ps

    def transform(image):
        inten = rgb_to_grayscale(image) * 255
        w = torch.eye(P, dtype=image.dtype).reshape(ps, ps, 1, P).permute(3, 2, 0, 1)
        patches = F.conv2d(inten.permute(0, 3, 1, 2), w, padding=max_distance).permute(0, 2, 3, 1)
        t = patches - inten
        return t / torch.sqrt(0.81 + t * t)

    t1, t2 = transform(im1), transform(im2_warped)
    d = (t1 - t2) ** 2
    dist = (d / (0.1 + d)).sum(3, keepdim=True)
    tmask = create_mask(mask, [[max_distance, max_distance], [max_distance, max_distance]])
    return charbonnier_loss(dist, mask * tmask)


def photometric_loss(im_diff, mask):
    return charbonnier_loss(im_diff, mask, beta=255)


def _smoothness_deltas(flow):
    """losses.py:206-222."""
    mask_x = create_mask(flow, [[0, 0], [0, 1]])
    mask_y = create_mask(flow, [[0, 1], [0, 0]])
    mask = torch.cat([mask_x, mask_y], 3)
    fx = [[0, 0, 0], [0, 1, -1], [0, 0, 0]]
    fy = [[0, 0, 0], [0, 1, 0], [0, -1, 0]]
    du = _stencil(flow[..., 0:1], [fx, fy])
    dv = _stencil(flow[..., 1:2], [fx, fy])
    return du, dv, mask


def smoothness_loss(flow):
    du, dv, mask = _smoothness_deltas(flow)
    return charbonnier_loss(du, mask) + charbonnier_loss(dv, mask)


def _second_order_deltas(flow):
    """losses.py:258-287."""
    mask_x = create_mask(flow, [[0, 0], [1, 1]])
    mask_y = create_mask(flow, [[1, 1], [0, 0]])
    mask_d = create_mask(flow, [[1, 1], [1, 1]])
    mask = torch.cat([mask_x, mask_y, mask_d, mask_d], 3)
    fs = [[[0, 0, 0], [1, -2, 1], [0, 0, 0]], [[0, 1, 0], [0, -2, 0], [0, 1, 0]],
          [[1, 0, 0], [0, -2, 0], [0, 0, 1]], [[0, 0, 1], [0, -2, 0], [1, 0, 0]]]
    return _stencil(flow[..., 0:1], fs), _stencil(flow[..., 1:2], fs), mask


def second_order_loss(flow):
    du, dv, mask = _second_order_deltas(flow)
    return charbonnier_loss(du, mask) + charbonnier_loss(dv, mask)


def gradient_loss(im1, im2_warped, mask):
    """losses.py:225-247 (Sobel gradient constancy)."""
    mask_x = create_mask(im1, [[0, 0], [1, 1]])
    mask_y = create_mask(im1, [[1, 1], [0, 0]])
    gmask = torch.cat([mask_x, mask_y], 3).repeat(1, 1, 1, 3)
    sx = [[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]]
    sy = [[-1, -2, -1], [0, 0, 0], [1, 2, 1]]

    def grads(im):
        return torch.cat([_stencil(im[..., c:c + 1], [sx, sy]) for c in range(3)], 3)

    return charbonnier_loss(grads(im1) - grads(im2_warped), mask * gmask)


def compute_losses(im1, im2, flow_fw, flow_bw, border_mask=None, mask_occlusion='', data_max_distance=1,
                   need=None):
    """losses.py:16-87.  `need`: optional set of loss names to evaluate (TF prunes the rest)."""
    need = set(LOSSES) if need is None else set(need)
    losses = {}
    im2_warped = image_warp(im2, flow_fw)
    im1_warped = image_warp(im1, flow_bw)
    im_diff_fw = im1 - im2_warped
    im_diff_bw = im2 - im1_warped
    if border_mask is None:
        mask_fw = create_outgoing_mask(flow_fw)
        mask_bw = create_outgoing_mask(flow_bw)
    else:
        mask_fw = border_mask
        mask_bw = border_mask
    flow_bw_warped = image_warp(flow_bw, flow_fw)
    flow_fw_warped = image_warp(flow_fw, flow_bw)
    flow_diff_fw = flow_fw + flow_bw_warped
    flow_diff_bw = flow_bw + flow_fw_warped
    mag_sq_fw = length_sq(flow_fw) + length_sq(flow_bw_warped)
    mag_sq_bw = length_sq(flow_bw) + length_sq(flow_fw_warped)
    occ_thresh_fw = 0.01 * mag_sq_fw + 0.5
    occ_thresh_bw = 0.01 * mag_sq_bw + 0.5
    fb_occ_fw = (length_sq(flow_diff_fw) > occ_thresh_fw).to(im1.dtype)
    fb_occ_bw = (length_sq(flow_diff_bw) > occ_thresh_bw).to(im1.dtype)
    if mask_occlusion == 'disocc' or 'sym' in need:      # losses.py:28-29 (TF prunes it when nothing consumes it)
        disocc_fw = (forward_warp(flow_fw) < DISOCC_THRESH).to(im1.dtype)
        disocc_bw = (forward_warp(flow_bw) < DISOCC_THRESH).to(im1.dtype)
    if mask_occlusion == 'fb':
        mask_fw = mask_fw * (1 - fb_occ_fw)
        mask_bw = mask_bw * (1 - fb_occ_bw)
    elif mask_occlusion == 'disocc':
        mask_fw = mask_fw * (1 - disocc_bw)
        mask_bw = mask_bw * (1 - disocc_fw)
    occ_fw = 1 - mask_fw
    occ_bw = 1 - mask_bw
    if 'sym' in need:
        losses['sym'] = charbonnier_loss(occ_fw - disocc_bw) + charbonnier_loss(occ_bw - disocc_fw)
    if 'occ' in need:
        losses['occ'] = charbonnier_loss(occ_fw) + charbonnier_loss(occ_bw)
    if 'photo' in need:
        losses['photo'] = photometric_loss(im_diff_fw, mask_fw) + photometric_loss(im_diff_bw, mask_bw)
    if 'grad' in need:
        losses['grad'] = gradient_loss(im1, im2_warped, mask_fw) + gradient_loss(im2, im1_warped, mask_bw)
    if 'smooth_1st' in need:
        losses['smooth_1st'] = smoothness_loss(flow_fw) + smoothness_loss(flow_bw)
    if 'smooth_2nd' in need:
        losses['smooth_2nd'] = second_order_loss(flow_fw) + second_order_loss(flow_bw)
    if 'fb' in need:
        losses['fb'] = charbonnier_loss(flow_diff_fw, mask_fw) + charbonnier_loss(flow_diff_bw, mask_bw)
    if 'ternary' in need:
        losses['ternary'] = (ternary_loss(im1, im2_warped, mask_fw, max_distance=data_max_distance) +
                             ternary_loss(im2, im1_warped, mask_bw, max_distance=data_max_distance))
    return losses


DEFAULT_PARAMS = dict(flownet='C', pyramid_loss=True, border_mask=True, ternary_weight=1.0, smooth_2nd_weight=3.0)


def regularization_loss(P, scale=0.0004):
    """slim.l2_regularizer(0.0004) on every '/weights' (flownet.py:176,200,218): scale * sum(w^2)/2."""
    return sum(scale * 0.5 * (v * v).sum() for n, v in P.items() if n.endswith('/weights'))


def pyramid_loss_from_flows(im1, im2, flows_fw, flows_bw, params, border_mask=None):
    """unsupervised.py:85-147, given un-normalised images in [0,255] — or, with `border_mask` given
    (the augmented per-sample mask of unsupervised.py:39-49), the geometrically augmented images in [0,1]."""
    if border_mask is None:
        im1 = im1 / 255.0
        im2 = im2 / 255.0
        border_mask = create_border_mask(im1, 0.1)
    layer_weights = [12.7, 4.35, 3.9, 3.4, 1.1]
    layer_patch_distances = [3, 2, 2, 1, 1]
    if params.get('full_res'):                                              # unsupervised.py:89-97
        layer_weights = [12.7, 5.5, 5.0, 4.35, 3.9, 3.4, 1.1]
        layer_patch_distances = [3, 3] + layer_patch_distances
        im1_s, im2_s, mask_s = im1, im2, border_mask
        final_flow_scale = FLOW_SCALE * 4
    else:
        im1_s, im2_s, mask_s = downsample(im1, 4), downsample(im2, 4), downsample(border_mask, 4)
        final_flow_scale = FLOW_SCALE
    combined = 0.0
    terms = {k: 0.0 for k in LOSSES}
    need = {l for l in LOSSES if params.get(l + '_weight')}
    levels = list(enumerate(zip(flows_fw, flows_bw))) if params.get('pyramid_loss') else [(0, (flows_fw[0], flows_bw[0]))]
    for i, (ffw, fbw) in levels:
        flow_scale = final_flow_scale / (2 ** i)
        losses = compute_losses(im1_s, im2_s, ffw * flow_scale, fbw * flow_scale,
                                border_mask=mask_s if params.get('border_mask') else None,
                                mask_occlusion=params.get('mask_occlusion', ''),
                                data_max_distance=layer_patch_distances[i], need=need)
        layer_loss = 0.0
        for l in LOSSES:
            if params.get(l + '_weight'):
                layer_loss = layer_loss + params[l + '_weight'] * losses[l]
                terms[l] = terms[l] + layer_weights[i] * losses[l]
        combined = combined + layer_weights[i] * layer_loss
        if i + 1 < len(levels):  # the reference also downsamples after the last level; that result is unused
            im1_s, im2_s, mask_s = downsample(im1_s, 2), downsample(im2_s, 2), downsample(mask_s, 2)
    return combined, terms


# ----------------------------------------------------------------------------
# augmentation (core/augment.py, core/spatial_transformer.py) — PARITY UNPINNED: the reference has no test for it
# ----------------------------------------------------------------------------
def affine_theta(tx, ty, rot_deg, scale, flip=None):
    """augment.py:17-48: theta = [[cos,-sin,tx],[sin,cos,ty]] @ diag(scale_x, scale, 1), scale_x = scale*flip.
    All arguments are [B] tensors (the random draws); flip is +-1 (or None: no horizontal flipping)."""
    rad = (rot_deg * np.pi) / 180.0
    sx = scale if flip is None else scale * flip
    c, s_ = torch.cos(rad), torch.sin(rad)
    row0 = torch.stack([c * sx, -s_ * scale, tx], 1)
    row1 = torch.stack([s_ * sx, c * scale, ty], 1)
    return torch.stack([row0, row1], 1)          # [B,2,3]


def stn_transformer(U, theta, out_size=None):
    """spatial_transformer.py:19-175.  U [B,H,W,C], theta [B,2,3]; bilinear with indices clipped BEFORE the weights
    are formed (:84-87,113-120), so border samples are not convex combinations."""
    B, H, W, C = U.shape
    Ho, Wo = (H, W) if out_size is None else out_size
    dt = U.dtype
    lin = lambda n: torch.tensor(-1.0, dtype=dt) + torch.arange(n, dtype=dt) * ((torch.tensor(1.0, dtype=dt) -
                                                                               torch.tensor(-1.0, dtype=dt)) / (n - 1))
    x_t = lin(Wo).view(1, 1, Wo).expand(1, Ho, Wo).reshape(1, -1)      # _meshgrid :122-140
    y_t = lin(Ho).view(1, Ho, 1).expand(1, Ho, Wo).reshape(1, -1)
    t = theta.to(dt)
    x_s = t[:, 0, 0:1] * x_t + t[:, 0, 1:2] * y_t + t[:, 0, 2:3]       # theta @ (x_t, y_t, 1) :160-163
    y_s = t[:, 1, 0:1] * x_t + t[:, 1, 1:2] * y_t + t[:, 1, 2:3]
    x = (x_s + 1.0) * float(W) / 2.0                                   # :75-76
    y = (y_s + 1.0) * float(H) / 2.0
    x0 = torch.floor(x).long()
    y0 = torch.floor(y).long()
    x1, y1 = x0 + 1, y0 + 1
    x0, x1 = x0.clamp(0, W - 1), x1.clamp(0, W - 1)
    y0, y1 = y0.clamp(0, H - 1), y1.clamp(0, H - 1)
    flat = U.reshape(B, H * W, C)
    g = lambda yy, xx: torch.gather(flat, 1, (yy * W + xx).unsqueeze(-1).expand(-1, -1, C))
    Ia, Ib, Ic, Id = g(y0, x0), g(y1, x0), g(y0, x1), g(y1, x1)        # :98-105
    x0f, x1f, y0f, y1f = x0.to(dt), x1.to(dt), y0.to(dt), y1.to(dt)
    wa = ((x1f - x) * (y1f - y)).unsqueeze(-1)
    wb = ((x1f - x) * (y - y0f)).unsqueeze(-1)
    wc = ((x - x0f) * (y1f - y)).unsqueeze(-1)
    wd = ((x - x0f) * (y - y0f)).unsqueeze(-1)
    out = wa * Ia + wb * Ib + wc * Ic + wd * Id
    return out.reshape(B, Ho, Wo, C)


def random_affine_apply(tensors, theta):
    """augment.py:50-55 with the random draws replaced by a given theta; stop_gradient -> detach."""
    return [stn_transformer(t_, theta).detach() for t_ in tensors]


def random_photometric_apply(ims, contrast, gamma, colour, noise, brightness):
    """augment.py:78-108 with the draws given: contrast/gamma/noise/brightness [B], colour [B,3].  noise and
    brightness are one scalar per SAMPLE (shape [num_batch,1], :83-91), not per pixel."""
    out = []
    v = lambda t_: t_.view(-1, 1, 1, 1)
    for im in ims:
        r = (im * (v(contrast) + 1.0) + v(brightness)) * colour.view(-1, 1, 1, 3)
        r = torch.clamp(r, 0.0, 1.0)
        r = torch.pow(r, v(1.0 / gamma))
        out.append((r + v(noise)).detach())
    return out


def augment_apply(im1, im2, aug):
    """unsupervised.py:31-60 with augment=True and the random draws supplied in `aug` (dict with theta_global,
    theta_local [B,2,3], contrast, gamma, noise, brightness [B], colour [B,3]).  im1/im2 in [0,255].
    Returns (im1_geo, im2_geo, border_mask [B,H,W,1], im1_photo, im2_photo), images in [0,1]."""
    im1 = im1 / 255.0
    im2 = im2 / 255.0
    border = create_border_mask(im1, 0.1).expand(im1.shape[0], -1, -1, -1)
    im1_geo, im2_geo, mask_g = random_affine_apply([im1, im2, border], aug['theta_global'])
    im2_geo, mask_l = random_affine_apply([im2_geo, border], aug['theta_local'])
    mask = mask_l * mask_g
    p1, p2 = random_photometric_apply([im1_geo, im2_geo], aug['contrast'], aug['gamma'], aug['colour'],
                                      aug['noise'], aug['brightness'])
    return im1_geo, im2_geo, mask, p1, p2


def unsupervised_loss(P, im1, im2, params=None, return_flow=False, augment=None):
    """unsupervised.py:27-164.  im1/im2: NHWC in [0,255].  augment: None (augment=False) or the dict of
    augment_apply."""
    params = dict(DEFAULT_PARAMS) if params is None else params
    mean = torch.tensor(CHANNEL_MEAN, dtype=im1.dtype) / 255.0
    border_mask = None
    if augment is None:
        a = im1 / 255.0 - mean
        b = im2 / 255.0 - mean
    else:
        im1, im2, border_mask, p1, p2 = augment_apply(im1, im2, augment)   # losses see the geo images (:63-65)
        a, b = p1 - mean, p2 - mean                                         # the network the photo ones (:67-68)
    flows_fw, flows_bw = flownet(P, a, b, flownet_spec=params.get('flownet', 'S'), backward_flow=True,
                                 train_all=bool(params.get('train_all')), full_resolution=bool(params.get('full_res')))
    flows_fw, flows_bw = flows_fw[-1], flows_bw[-1]
    combined, terms = pyramid_loss_from_flows(im1, im2, flows_fw, flows_bw, params, border_mask=border_mask)
    final_loss = combined + regularization_loss(P)
    if not return_flow:
        return final_loss
    H, W = im1.shape[1:3]
    if params.get('full_res'):                                              # unsupervised.py:95-97
        return final_loss, flows_fw[0] * FLOW_SCALE * 4, flows_bw[0] * FLOW_SCALE * 4, terms
    ffw = resize_bilinear_tf1(flows_fw[0], H, W) * FLOW_SCALE * 4
    fbw = resize_bilinear_tf1(flows_bw[0], H, W) * FLOW_SCALE * 4
    return final_loss, ffw, fbw, terms


# ----------------------------------------------------------------------------
# metrics / optimiser
# ----------------------------------------------------------------------------
def flow_error_avg(f1, f2, mask):
    """flow_util.py:98-103,122-123 (EPE)."""
    d = torch.sqrt(((f1 - f2) ** 2).sum(3, keepdim=True)) * mask
    return d.sum() / mask.sum()


def adam_step_tf(P, G, M, V, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer update (train.py:151-152): eps is NOT bias-corrected."""
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    for k in P:
        M[k] = beta1 * M[k] + (1 - beta1) * G[k]
        V[k] = beta2 * V[k] + (1 - beta2) * G[k] * G[k]
        P[k] = P[k] - lr_t * M[k] / (torch.sqrt(V[k]) + eps)


def learning_rate_at(params, decay_iters):
    """train.py:225-244 (non-manual branch)."""
    decay_interval = params['decay_interval']
    decay_after = params.get('decay_after', 0)
    if decay_iters >= decay_after:
        decay = (decay_iters // decay_interval) - decay_after / decay_interval
        return params['learning_rate'] / (2 ** decay)
    return params['learning_rate']
