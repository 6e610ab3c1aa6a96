"""Static-plan training-step engine for FlowNetC / FlowNetS / stacked C->S->S on one MI355X.

Replaces, for the hot path, what TensorFlow's executor does for the reference graph built by
  core/flownet.py:14-237  (network, shared weights for both images and both flow directions)
  core/unsupervised.py:27-164 (normalisation, image/mask pyramid, per-level losses, L2 term)
  core/train.py:147-185 (Adam, single-GPU minimize)
with an explicit launch list over libunflow_hip.so:

  * one "directed batch" of N = 2B samples: samples [0,B) are (im1 -> im2), [B,2B) are (im2 -> im1), so the
    feature towers, both flownet_c passes and both loss directions of the reference run as ONE
    pass over N samples with the reference's weight sharing; the filter gradients then sum over both
    directions exactly like TF's gradient aggregation over reused variables;
  * channels-last activations; every concat of flownet.py is a channel slice of a preallocated buffer
    (producers write into it, consumers read with a channel stride) — no concat, no transposes;
  * channel counts padded to multiples of 4 at the END of the concat (473->476, 1026->1028, 770->772,
    386->388, 194->196, RGB 3->4): the padded activations are zero and never written, the padded weight
    rows are zero and receive exactly zero gradient, so the maths is unchanged;
  * backward is an explicit reverse launch list; gradient accumulation into multi-consumer tensors
    and the leaky-ReLU derivative are fused into the dgrad epilogues (accumulate / act range);
  * parameters, gradients and Adam moments live in four flat fp32 buffers (weights first, then
    biases) so L2 + Adam is one fused kernel and the data-parallel all-reduce is over one buffer.
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, stream, cf, cl
from ..ops import workspace
from . import layers as L

FLOW_SCALE = 5.0                                   # flownet.py:11
CHANNEL_MEAN = [104.920005, 110.1753, 114.785955]  # core/input.py:45
LAYER_WEIGHTS = [12.7, 4.35, 3.9, 3.4, 1.1]        # unsupervised.py:87
LAYER_PATCH_DISTANCES = [3, 2, 2, 1, 1]            # unsupervised.py:88
L2_SCALE = 0.0004                                  # flownet.py:176
LOSSES = ['occ', 'sym', 'fb', 'grad', 'ternary', 'photo', 'smooth_1st', 'smooth_2nd']  # unsupervised.py:15

DEFAULT_PARAMS = dict(flownet='C', pyramid_loss=True, border_mask=True, ternary_weight=1.0, smooth_2nd_weight=3.0)


def pad4(c):
    return (c + 3) // 4 * 4


class Layer:
    __slots__ = ('name', 'kind', 'k', 'cin', 'cout', 'stride', 'act', 'cin_p', 'w', 'b', 'dw', 'db')

    def __init__(self, name, kind, k, cin, cout, stride, act):
        self.name, self.kind, self.k, self.cin, self.cout, self.stride, self.act = name, kind, k, cin, cout, stride, act
        self.cin_p = cin if (kind == 'deconv' and cin == 2) else pad4(cin)

    def wshape(self):
        if self.kind == 'conv':
            return (self.k, self.k, self.cin_p, self.cout)   # HWIO
        return (self.k, self.k, self.cout, self.cin_p)       # conv2d_transpose: [k,k,out,in]


def _decoder_layers(c, skip2=128):
    """_flownet_upconv variables (flownet.py:89-131) under scope prefix c."""
    return [
        Layer(c + 'flow6', 'conv', 3, 1024, 2, 1, False),
        Layer(c + 'deconv5', 'deconv', 4, 1024, 512, 2, True), Layer(c + 'flow6_up5', 'deconv', 4, 2, 2, 2, False),
        Layer(c + 'flow5', 'conv', 3, 1026, 2, 1, False),
        Layer(c + 'deconv4', 'deconv', 4, 1026, 256, 2, True), Layer(c + 'flow5_up4', 'deconv', 4, 2, 2, 2, False),
        Layer(c + 'flow4', 'conv', 3, 770, 2, 1, False),
        Layer(c + 'deconv3', 'deconv', 4, 770, 128, 2, True), Layer(c + 'flow4_up3', 'deconv', 4, 2, 2, 2, False),
        Layer(c + 'flow3', 'conv', 3, 386, 2, 1, False),
        Layer(c + 'deconv2', 'deconv', 4, 386, 64, 2, True), Layer(c + 'flow3_up2', 'deconv', 4, 2, 2, 2, False),
        Layer(c + 'flow2', 'conv', 3, skip2 + 64 + 2, 2, 1, False),
    ]


def _contracting_layers(c):
    return [
        Layer(c + 'conv4', 'conv', 3, 256, 512, 2, True), Layer(c + 'conv4_1', 'conv', 3, 512, 512, 1, True),
        Layer(c + 'conv5', 'conv', 3, 512, 512, 2, True), Layer(c + 'conv5_1', 'conv', 3, 512, 512, 1, True),
        Layer(c + 'conv6', 'conv', 3, 512, 1024, 2, True), Layer(c + 'conv6_1', 'conv', 3, 1024, 1024, 1, True),
    ]


def flownet_c_layers(scope=''):
    """Variables of flownet_c_features + flownet_c in the reference's creation order (flownet.py:195-237,89-131)."""
    f, c = scope + 'flownet_c_features/', scope + 'flownet_c/'
    return [
        Layer(f + 'conv1', 'conv', 7, 3, 64, 2, True), Layer(f + 'conv2', 'conv', 5, 64, 128, 2, True),
        Layer(f + 'conv3', 'conv', 5, 128, 256, 2, True),
        Layer(c + 'conv_redir', 'conv', 1, 256, 32, 1, True), Layer(c + 'conv3_1', 'conv', 3, 473, 256, 1, True),
    ] + _contracting_layers(c) + _decoder_layers(c)


def flownet_s_layers(scope='', in_channels=6):
    """Variables of flownet_s (flownet.py:166-192): 6 input channels, or 14 for a refinement stage (:56-57)."""
    c = scope + 'flownet_s/'
    return [
        Layer(c + 'conv1', 'conv', 7, in_channels, 64, 2, True), Layer(c + 'conv2', 'conv', 5, 64, 128, 2, True),
        Layer(c + 'conv3', 'conv', 5, 128, 256, 2, True), Layer(c + 'conv3_1', 'conv', 3, 256, 256, 1, True),
    ] + _contracting_layers(c) + _decoder_layers(c)


class _Stage:
    """One network of the (possibly stacked) spec: its layers, activations, gradients and launch lists."""

    def __init__(self, eng, kind, index):
        self.eng, self.kind, self.index = eng, kind, index
        scope = '' if index == 0 else 'stack_%d_flownet/' % index   # flownet.py:72-77
        self.in_ch = 3 if kind == 'C' else (6 if index == 0 else 14)
        self.layers = flownet_c_layers(scope) if kind == 'C' else flownet_s_layers(scope, self.in_ch)
        self.by_name = {l.name.split('/')[-1]: l for l in self.layers}
        self.trainable = True

    def alloc(self):
        e = self.eng
        N, H, W, dev = e.N, e.H, e.W, e.dev
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        a = self.act = {}
        if self.kind == 'S':
            a['x0s'] = z(N, H, W, pad4(self.in_ch))
        a['c1'] = z(N, H // 2, W // 2, 64)
        a['cat2'] = z(N, H // 4, W // 4, 196)
        a['c3'] = z(N, H // 8, W // 8, 256)
        if self.kind == 'C':
            a['catc'] = z(N, H // 8, W // 8, 476)
        a['cat3'] = z(N, H // 8, W // 8, 388)
        a['c4'] = z(N, H // 16, W // 16, 512)
        a['cat4'] = z(N, H // 16, W // 16, 772)
        a['c5'] = z(N, H // 32, W // 32, 512)
        a['cat5'] = z(N, H // 32, W // 32, 1028)
        a['c6'] = z(N, H // 64, W // 64, 1024)
        a['c6_1'] = z(N, H // 64, W // 64, 1024)
        for lvl, d in zip((2, 3, 4, 5, 6), (4, 8, 16, 32, 64)):
            a['flow%d' % lvl] = z(N, H // d, W // d, 2)
        self.grad = {k: torch.zeros_like(v) for k, v in a.items() if k != 'x0s'} if self.trainable else {}
        if self.kind == 'S' and self.index > 0 and e.train_all:
            self.grad['x0s'] = torch.zeros_like(a['x0s'])      # d loss / d stage input -> the previous network

    def _sl(self, name, lo, hi):
        return self.act[name][..., lo:hi]

    def _gsl(self, name, lo, hi):
        return self.grad[name][..., lo:hi]

    def _conv(self, lname, x, y):
        l = self.by_name[lname]
        if l.kind == 'conv':
            L.conv2d_fwd(x, l.w, l.b, y, l.stride, l.act)
        else:
            L.conv2d_transpose_fwd(x, l.w, l.b, y, l.act)

    # -------------------------------------------------------------- forward
    def forward(self, prev_flow2=None):
        e, a, s = self.eng, self.act, self._sl
        B, N = e.B, e.N
        if self.kind == 'C':
            self._conv('conv1', e.x0, a['c1'])
            self._conv('conv2', a['c1'], s('cat2', 0, 128))
            self._conv('conv3', s('cat2', 0, 128), a['c3'])
            h8, w8 = e.H // 8, e.W // 8
            corr_out = s('catc', 32, 473)
            check(_lib.lib().unflow_correlation_nhwc_fwd(ptr(a['c3']), ptr(a['c3']), 256, B, ptr(corr_out), 476, N, 256,
                                                         h8, w8, 1, 20, 20, 1, 2, stream()), "correlation")
            self._conv('conv_redir', a['c3'], s('catc', 0, 32))
            self._conv('conv3_1', a['catc'], s('cat3', 0, 256))
        else:
            x0s = a['x0s']
            pf = prev_flow2
            check(_lib.lib().unflow_stack_input(ptr(e.x0), ptr(pf), ptr(x0s), x0s.shape[3], B, N, e.H, e.W,
                                                0 if pf is None else pf.shape[1], 0 if pf is None else pf.shape[2],
                                                cf(4 * FLOW_SCALE), stream()), "stack_input")
            self._conv('conv1', x0s, a['c1'])
            self._conv('conv2', a['c1'], s('cat2', 0, 128))
            self._conv('conv3', s('cat2', 0, 128), a['c3'])
            self._conv('conv3_1', a['c3'], s('cat3', 0, 256))
        self._conv('conv4', s('cat3', 0, 256), a['c4'])
        self._conv('conv4_1', a['c4'], s('cat4', 0, 512))
        self._conv('conv5', s('cat4', 0, 512), a['c5'])
        self._conv('conv5_1', a['c5'], s('cat5', 0, 512))
        self._conv('conv6', s('cat5', 0, 512), a['c6'])
        self._conv('conv6_1', a['c6'], a['c6_1'])
        # refinement decoder (_flownet_upconv, flownet.py:89-131)
        self._conv('flow6', a['c6_1'], a['flow6'])
        self._conv('deconv5', a['c6_1'], s('cat5', 512, 1024))
        self._conv('flow6_up5', a['flow6'], s('cat5', 1024, 1026))
        self._conv('flow5', a['cat5'], a['flow5'])
        self._conv('deconv4', a['cat5'], s('cat4', 512, 768))
        self._conv('flow5_up4', a['flow5'], s('cat4', 768, 770))
        self._conv('flow4', a['cat4'], a['flow4'])
        self._conv('deconv3', a['cat4'], s('cat3', 256, 384))
        self._conv('flow4_up3', a['flow4'], s('cat3', 384, 386))
        self._conv('flow3', a['cat3'], a['flow3'])
        self._conv('deconv2', a['cat3'], s('cat2', 128, 192))
        self._conv('flow3_up2', a['flow3'], s('cat2', 192, 194))
        self._conv('flow2', a['cat2'], a['flow2'])

    # -------------------------------------------------------------- backward
    def _bwd(self, lname, x, dz, dx=None, accumulate=False, act_src=None, act_lo=0, act_hi=0):
        """Filter gradient of layer `lname` from dz (d pre-activation), then (optionally) the data gradient."""
        l = self.by_name[lname]
        e = self.eng
        if e._bias_plan is None:
            e._bias_jobs.append((dz, l))         # bias gradients: one batched column-sum launch at the end
        # The filter gradient only needs dz (final at this point) and the stored activation: it runs on the side
        # stream, concurrently with the data-gradient chain on the main stream (its tail waves and the latency-bound
        # flow-head kernels fill the CUs the other kernel leaves idle).  Own split-K scratch (slot 3).
        # ('small': only the latency-bound layers, the Cout = 2 flow heads and the 2->2 deconvs)
        side = e.side if (e.side_all or l.cout <= 4) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), L.ws_slot(3):
                self._wgrad(l, x, dz)
        else:
            self._wgrad(l, x, dz)
        if dx is not None:
            if l.kind == 'conv':
                L.conv2d_bwd_data(dz, l.w, dx, l.stride, accumulate, act_src, act_lo, act_hi)
            else:
                L.conv2d_transpose_bwd_data(dz, l.w, dx, accumulate, act_src, act_lo, act_hi)

    @staticmethod
    def _wgrad(l, x, dz):
        if l.kind == 'conv':
            L.conv2d_bwd_filter(x, dz, l.dw, None, l.stride)
        else:
            L.conv2d_transpose_bwd_filter(x, dz, l.dw, None)

    def backward(self, part=None):
        """part None: everything; 0: decoder + conv6_1..conv4 (94 % of the parameters — their gradients are final
        afterwards, so their all-reduce can start); 1: conv3_1 .. conv1."""
        if part in (None, 0):
            self._backward_deep()
        if part in (None, 1):
            self._backward_shallow()
        if self.eng.side is not None:     # join: every filter gradient of this part is final after this point
            torch.cuda.current_stream().wait_stream(self.eng.side)

    def _backward_deep(self):
        e, a, g, s, gs = self.eng, self.act, self.grad, self._sl, self._gsl
        # level 2 (flow2 head reads concat2 = [conv2 | deconv2 | flow3_up2])
        self._bwd('flow2', a['cat2'], g['flow2'], g['cat2'], False, a['cat2'], 128, 192)
        self._bwd('flow3_up2', a['flow3'], gs('cat2', 192, 194), g['flow3'], True)
        self._bwd('flow3', a['cat3'], g['flow3'], g['cat3'], False)
        self._bwd('deconv2', a['cat3'], gs('cat2', 128, 192), g['cat3'], True, a['cat3'], 256, 384)
        self._bwd('flow4_up3', a['flow4'], gs('cat3', 384, 386), g['flow4'], True)
        self._bwd('flow4', a['cat4'], g['flow4'], g['cat4'], False)
        self._bwd('deconv3', a['cat4'], gs('cat3', 256, 384), g['cat4'], True, a['cat4'], 512, 768)
        self._bwd('flow5_up4', a['flow5'], gs('cat4', 768, 770), g['flow5'], True)
        self._bwd('flow5', a['cat5'], g['flow5'], g['cat5'], False)
        self._bwd('deconv4', a['cat5'], gs('cat4', 512, 768), g['cat5'], True, a['cat5'], 512, 1024)
        self._bwd('flow6_up5', a['flow6'], gs('cat5', 1024, 1026), g['flow6'], True)
        self._bwd('flow6', a['c6_1'], g['flow6'], g['c6_1'], False)
        self._bwd('deconv5', a['c6_1'], gs('cat5', 512, 1024), g['c6_1'], True, a['c6_1'], 0, 1024)
        # contracting part
        self._bwd('conv6_1', a['c6'], g['c6_1'], g['c6'], False, a['c6'], 0, 1024)
        self._bwd('conv6', s('cat5', 0, 512), g['c6'], gs('cat5', 0, 512), True, s('cat5', 0, 512), 0, 512)
        self._bwd('conv5_1', a['c5'], gs('cat5', 0, 512), g['c5'], False, a['c5'], 0, 512)
        self._bwd('conv5', s('cat4', 0, 512), g['c5'], gs('cat4', 0, 512), True, s('cat4', 0, 512), 0, 512)
        self._bwd('conv4_1', a['c4'], gs('cat4', 0, 512), g['c4'], False, a['c4'], 0, 512)
        self._bwd('conv4', s('cat3', 0, 256), g['c4'], gs('cat3', 0, 256), True, s('cat3', 0, 256), 0, 256)

    def _backward_shallow(self):
        e, a, g, s, gs = self.eng, self.act, self.grad, self._sl, self._gsl
        B, N = e.B, e.N
        if self.kind == 'C':
            self._bwd('conv3_1', a['catc'], gs('cat3', 0, 256), g['catc'], False, a['catc'], 0, 32)
            # correlation: gradient wrt the shared feature tensor (both roles of every sample), then conv_redir adds
            h8, w8 = e.H // 8, e.W // 8
            check(_lib.lib().unflow_correlation_nhwc_bwd(ptr(gs('catc', 32, 473)), 476, ptr(a['c3']), ptr(a['c3']), 256,
                                                         B, ptr(g['c3']), ptr(None), 256, 1, N, 256, h8, w8, 1, 20, 20,
                                                         1, 2, stream()), "correlation_grad")
            self._bwd('conv_redir', a['c3'], gs('catc', 0, 32), g['c3'], True, a['c3'], 0, 256)
            x_in = e.x0
        else:
            self._bwd('conv3_1', a['c3'], gs('cat3', 0, 256), g['c3'], False, a['c3'], 0, 256)
            x_in = a['x0s']
        self._bwd('conv3', s('cat2', 0, 128), g['c3'], gs('cat2', 0, 128), True, s('cat2', 0, 128), 0, 128)
        self._bwd('conv2', a['c1'], gs('cat2', 0, 128), g['c1'], False, a['c1'], 0, 64)
        # inputs are data (or behind stop_gradient, flownet.py:51-54) unless train_all reaches a refinement stage
        self._bwd('conv1', x_in, g['c1'], g.get('x0s'))


class FlowNetEngine:
    """Bidirectional forward / loss / backward / Adam of a FlowNet spec on one GPU, fixed (B, H, W).
    params['flownet']: 'C', 'S' or a stack 'CS', 'CSS', 'SS' ... (flownet.py:14-81).  In a stack only the last network
    is trained unless params['train_all'] (flownet.py:51-54, train.py:29-37); frozen stages run forward only and —
    exactly as in the reference, whose regulariser and optimizer span all variables — still receive the L2 gradient in
    the Adam update.  With train_all the gradient also flows back through every inter-stage input (upsampled flow, warp,
    |warp - first|) into the earlier networks."""

    def __init__(self, batch, height, width, params=None, device=None, seed=0):
        assert height % 64 == 0 and width % 64 == 0, "FlowNet needs H, W divisible by 64"
        self.params = dict(DEFAULT_PARAMS) if params is None else dict(params)
        if self.params.get('mask_occlusion', '') not in ('', 'fb', 'disocc'):   # unsupervised.py:125-126
            raise ValueError("mask_occlusion must be one of 'fb', 'disocc', ''")
        spec = self.params.get('flownet', 'C')
        if not spec or any(ch not in 'CS' for ch in spec) or 'C' in spec[1:]:
            raise ValueError("flownet spec must be 'C' or 'S' followed by 'S' refinement nets (full-size nets only)")
        self.train_all = bool(self.params.get('train_all')) and len(spec) > 1     # train.py:29-37, flownet.py:51-54
        if self.params.get('full_res'):
            raise NotImplementedError("full_res decoder is not implemented")
        self.spec = spec
        self.B, self.H, self.W = batch, height, width
        self.N = 2 * batch
        self.dev = torch.device('cuda:0') if device is None else device
        self.stages = [_Stage(self, k, i) for i, k in enumerate(spec)]
        for st in self.stages[:-1]:
            st.trainable = self.train_all
        self.layers = [l for st in self.stages for l in st.layers]
        self.by_name = self.stages[-1].by_name
        self._alloc_params()
        self._alloc_activations()
        self._build_masks()
        self.step_count = 0
        self._bias_jobs, self._bias_plan = [], None
        self.defer_l2 = False      # True: forward_loss leaves the L2 term to adam_step (train_step / bench)
        self.fused_pyramid = os.environ.get('UNFLOW_FUSED_PYRAMID', '1') != '0'   # default loss terms: 4 launches for all levels
        self._pyr_cache = None
        # optional second HIP stream for filter gradients (see _Stage._bwd), off by default: both variants measured
        # slower inside the captured graph on MI355X — 'all' 397 vs 405 pairs/s, 'small' (only the latency-bound flow
        # heads / 2->2 deconvs) 438.6 vs 445.2
        mode = os.environ.get('UNFLOW_SIDE_STREAM', '0')      # '0' (default) | 'small' | '1' / 'all'
        self.side = torch.cuda.Stream(self.dev) if mode != '0' else None
        self.side_all = mode in ('1', 'all')
        if seed is not None:
            self.init_params(seed)

    # ------------------------------------------------------------------ parameters
    def _alloc_params(self):
        nw = sum(int(torch.Size(l.wshape()).numel()) for l in self.layers)
        nb = sum(l.cout for l in self.layers)
        self.n_weights, self.n_params = nw, nw + nb
        z = lambda: torch.zeros(self.n_params, dtype=torch.float32, device=self.dev)
        self.P, self.G, self.M, self.V = z(), z(), z(), z()
        off = 0
        for l in self.layers:
            n = int(torch.Size(l.wshape()).numel())
            l.w = self.P[off:off + n].view(l.wshape())
            l.dw = self.G[off:off + n].view(l.wshape())
            off += n
        for l in self.layers:
            l.b = self.P[off:off + l.cout]
            l.db = self.G[off:off + l.cout]
            off += l.cout

    def init_params(self, seed=0):
        """layers.variance_scaling_initializer() (flownet.py:177): truncated normal, stddev sqrt(1.3*2/fan_in);
        zero biases.  Drawn on the host in the reference's variable order so the oracle can share them."""
        from_tf = OrderedDict()
        gen = torch.Generator().manual_seed(seed)
        for l in self.layers:
            shape = (l.k, l.k, l.cin, l.cout) if l.kind == 'conv' else (l.k, l.k, l.cout, l.cin)
            fan_in = l.k * l.k * (l.cin if l.kind == 'conv' else l.cout)
            std = math.sqrt(1.3 * 2.0 / fan_in)
            t = torch.empty(shape, dtype=torch.float32)
            torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std, generator=gen)
            from_tf[l.name + '/weights'] = t
            from_tf[l.name + '/biases'] = torch.zeros(l.cout)
        self.load_tf_params(from_tf)
        return from_tf

    def load_tf_params(self, tf_params):
        """tf_params: {'<scope>/<layer>/weights': HWIO (conv) or [k,k,out,in] (deconv), '.../biases'} on any device."""
        self.P.zero_()
        for l in self.layers:
            w = tf_params[l.name + '/weights'].to(self.dev, torch.float32)
            if l.kind == 'conv':
                l.w[:, :, :l.cin, :] = w
            else:
                l.w[:, :, :, :l.cin] = w
            l.b.copy_(tf_params[l.name + '/biases'].to(self.dev, torch.float32))

    def _export(self, wattr, battr):
        out = OrderedDict()
        for l in self.layers:
            w = getattr(l, wattr)
            out[l.name + '/weights'] = (w[:, :, :l.cin, :] if l.kind == 'conv' else w[:, :, :, :l.cin]).detach().cpu().clone()
            out[l.name + '/biases'] = getattr(l, battr).detach().cpu().clone()
        return out

    def export_tf_params(self):
        return self._export('w', 'b')

    def export_tf_grads(self):
        """Gradients of the data loss (the L2-regulariser gradient 0.0004*w is fused into adam_step)."""
        return self._export('dw', 'db')

    # ------------------------------------------------------------------ buffers
    def _alloc_activations(self):
        N, H, W, dev = self.N, self.H, self.W, self.dev
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.x0 = z(N, H, W, 4)       # mean-subtracted network input (4th channel zero)
        self.im01 = z(N, H, W, 3)     # images in [0,1] for the losses
        for st in self.stages:
            st.alloc()
        last = self.stages[-1]
        # views used by the loss code, tests and tools: the trained (last) network
        self.act = dict(last.act, x0=self.x0, im01=self.im01)
        self.grad = last.grad
        # loss-side pyramid
        self.lv = []
        for i, d in enumerate((4, 8, 16, 32, 64)):
            h, w = H // d, W // d
            self.lv.append(dict(h=h, w=w, im=z(N, h, w, 3), gray1=z(N, h, w), gray2w=z(N, h, w), dist=z(N, h, w),
                                dgray=z(N, h, w), flow=last.act['flow%d' % (i + 2)], gflow=last.grad['flow%d' % (i + 2)]))
        self.loss_acc = z(1)
        self.raw = z(N, H, W, 3)
        self.final_flow = z(N, H, W, 2)
        self.mean_host = (_lib.ctypes.c_float * 3)(*CHANNEL_MEAN)
        self.epe_out = z(2)

    def _build_masks(self):
        """create_border_mask(im, 0.1) (losses.py:338-344) then downsample 4, 2, 2, 2, 2 (unsupervised.py:101,147)."""
        from .. import ops
        H, W = self.H, self.W
        sz = int(math.ceil(min(H, W) * 0.1))
        m = torch.zeros(1, H, W, 1, device=self.dev)
        m[:, sz:H - sz, sz:W - sz] = 1.0
        self.border0, self._aug, self._mask_aug = m, None, False
        ones = torch.ones(1, H, W, 1, device=self.dev)
        use_border = bool(self.params.get('border_mask'))
        cur = ops.downsample(m, 4)
        cur1 = ops.downsample(ones, 4)
        for i, lv in enumerate(self.lv):
            lv['mask'] = (cur if use_border else cur1).reshape(1, lv['h'], lv['w']).contiguous()
            lv['mask_static'], lv['n_mask'] = lv['mask'], 1
            if i + 1 < len(self.lv):
                cur = ops.downsample(cur, 2)
                cur1 = ops.downsample(cur1, 2)

    # ------------------------------------------------------------------ forward
    def set_input(self, im1, im2, augment=None):
        """im1, im2: [B,H,W,3] float32 in [0,255] (what the reference's input queue delivers).
        augment: None (augment=False) or the draws of core.augment.draw_training_augmentation — then the step sees
        what unsupervised.py:37-68 builds: geometrically augmented images for the losses, photometrically augmented
        mean-free ones for the network, and a per-sample border mask (product of the two warped masks)."""
        B, N, H, W = self.B, self.N, self.H, self.W
        lib = _lib.lib()
        self.raw[:B].copy_(im1)
        self.raw[B:].copy_(im2)
        if augment is None:
            check(lib.unflow_prepare_images(ptr(self.raw), ptr(self.x0), ptr(self.im01),
                                            self.mean_host, cl(N * H * W), stream()), "prepare_images")
            if self._mask_aug:
                for lv in self.lv:
                    lv['mask'], lv['n_mask'] = lv['mask_static'], 1
                self._mask_aug = False
            return
        from . import augment as A
        if self._aug is None:
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.dev)
            self._aug = dict(tmp=z(N, H, W, 3), mg=z(B, H, W, 1), ml=z(B, H, W, 1),
                             masks=[z(B, lv['h'], lv['w'], 1) for lv in self.lv])
        a = self._aug
        check(lib.unflow_prepare_images(ptr(self.raw), ptr(self.x0), ptr(a['tmp']), self.mean_host, cl(N * H * W),
                                        stream()), "prepare_images")
        tg, tl = augment['theta_global'], augment['theta_local']
        A.transformer(a['tmp'], tg, out=self.im01, n_samples=N)            # im1_geo, first pass of im2 (:40-44)
        A.transformer(self.im01[B:], tl, out=a['tmp'][B:], n_samples=B)    # im2 locally (:47-50)
        self.im01[B:].copy_(a['tmp'][B:])
        if self.params.get('border_mask'):
            A.transformer(self.border0, tg, out=a['mg'], n_samples=B)
            A.transformer(self.border0, tl, out=a['ml'], n_samples=B)
            a['mg'].mul_(a['ml'])                                          # border_mask_local * global (:51)
            from .. import ops
            cur = ops.downsample(a['mg'], 4)
            for i, lv in enumerate(self.lv):
                a['masks'][i].copy_(cur)
                lv['mask'], lv['n_mask'] = a['masks'][i].view(B, lv['h'], lv['w']), B
                if i + 1 < len(self.lv):
                    cur = ops.downsample(cur, 2)
            self._mask_aug = True
        A.photometric(self.im01, augment, out=self.x0, mean=CHANNEL_MEAN)  # im*_photo - channel_mean (:53-57,67-68)

    def forward_net(self):
        """flownet(im1, im2, spec, backward_flow=True) (flownet.py:14-81): every stage in order; a refinement stage
        consumes the previous stage's finest flow."""
        prev = None
        for st in self.stages:
            st.forward(prev)
            prev = st.act['flow2']

    def forward_loss(self, with_grad=True):
        """compute_losses + the pyramid assembly (losses.py:16-87, unsupervised.py:85-150) over the directed batch;
        with_grad also leaves d(loss)/d(flowN) in self.grad['flowN'].  Terms enter iff their `<name>_weight` is set
        (unsupervised.py:136-141), exactly the pruning TF does."""
        lib = _lib.lib()
        st = stream()
        N, B = self.N, self.B
        P = self.params
        wt = lambda k: float(P.get(k + '_weight') or 0.0)
        self.loss_acc.zero_()
        occl = {'': 0, None: 0, 'fb': 1, 'disocc': 2}[P.get('mask_occlusion', '')]
        use_border = bool(P.get('border_mask'))
        levels = self.lv if P.get('pyramid_loss') else self.lv[:1]
        # image pyramid: downsample(im, 4) then successive downsample(., 2) (unsupervised.py:99-100,145-146)
        check(lib.unflow_downsample_fwd(ptr(self.act['im01']), ptr(self.lv[0]['im']), N, self.H, self.W, 3, 4, st),
              "downsample")
        for i in range(1, len(levels)):
            p = self.lv[i - 1]
            check(lib.unflow_downsample_fwd(ptr(p['im']), ptr(self.lv[i]['im']), N, p['h'], p['w'], 3, 2, st),
                  "downsample")
        need_fbwarp = bool(wt('fb')) or occl == 1
        need_fwarp = bool(wt('sym')) or occl == 2
        need_mask_terms = need_fbwarp or need_fwarp or bool(wt('occ')) or not use_border
        # the default terms (config.ini [train]: ternary + second-order, border mask): the whole pyramid in four launches
        if (not need_mask_terms and wt('ternary') and wt('smooth_2nd')
                and not any(wt(k) for k in ('smooth_1st', 'photo', 'grad')) and self.fused_pyramid):
            check(lib.unflow_loss_pyramid_default(self._pyramid_levels(levels, wt), len(levels), N, B, ptr(self.loss_acc),
                                                  int(bool(with_grad)), st), "loss_pyramid")
            levels = []
        for i, lv in enumerate(levels):
            h, w = lv['h'], lv['w']
            fs = FLOW_SCALE / (2 ** i)
            lw = LAYER_WEIGHTS[i]
            flow, gflow = lv['flow'], lv['gflow']
            gf = ptr(gflow) if with_grad else ptr(None)
            n1 = B * h * w
            wrote = [False]

            def acc():      # first writer of gflow overwrites, later ones accumulate
                a = 1 if wrote[0] else 0
                wrote[0] = True
                return a
            # ---- smoothness terms (flow only)
            if wt('smooth_2nd'):
                check(lib.unflow_second_order_fwd_bwd(ptr(flow), cf(fs), ptr(self.loss_acc), gf, acc(),
                                                      cf(lw * wt('smooth_2nd')), cf(n1 * 4), N, h, w, st), "second_order")
            if wt('smooth_1st'):
                check(lib.unflow_smooth_1st_fwd_bwd(ptr(flow), cf(fs), ptr(self.loss_acc), gf, acc(),
                                                    cf(lw * wt('smooth_1st')), cf(n1 * 2), N, h, w, st), "smooth_1st")
            if with_grad and not wrote[0]:
                gflow.zero_()
                wrote[0] = True
            # ---- masks, fb / occ / sym
            mask, n_mask = lv['mask'], lv['n_mask']
            if need_mask_terms:
                self._level_extra(lv)
                warped = fwm = None
                if need_fbwarp:     # image_warp(flow_other, flow_own) (losses.py:38-39); scaling is linear, applied later
                    warped = lv['fwarped']
                    check(lib.unflow_image_warp_fwd(ptr(flow), 2, ptr(flow), cf(fs), ptr(warped), ptr(None), B, N, h, w,
                                                    2, st), "image_warp(flow)")
                if need_fwarp:      # forward_warp(flow*scale) (losses.py:28-29), deterministic accumulation
                    torch.mul(flow, fs, out=lv['fscaled'])
                    fwm = lv['fwmap']
                    ws = workspace(8 * N * h * w, self.dev, slot=2)
                    check(lib.unflow_forward_warp_fwd(ptr(lv['fscaled']), ptr(fwm), N, h, w, 1, ptr(ws),
                                                      _lib.csz(ws.numel() * 4), st), "forward_warp")
                a = acc() if (with_grad and wt('fb')) else 0
                check(lib.unflow_mask_terms(ptr(flow), ptr(warped), ptr(fwm), ptr(lv['mask'] if use_border else None),
                                            lv['n_mask'],
                                            cf(fs), occl, ptr(lv['maskN']), ptr(self.loss_acc),
                                            gf if wt('fb') else ptr(None), ptr(lv['gwarped']), a, cf(lw * wt('fb')),
                                            cf(lw * wt('occ')), cf(lw * wt('sym')), B, B, N, h, w, st), "mask_terms")
                mask, n_mask = lv['maskN'], N
                if with_grad and wt('fb'):
                    # back through image_warp(flow_other, flow_own): scatter into the partner's flow gradient (via a
                    # scratch buffer: the scatter must not race with the in-place accumulation) + own flow gradient
                    lv['dimtmp'].zero_()
                    check(lib.unflow_image_warp_bwd(ptr(lv['gwarped']), ptr(flow), 2, ptr(flow), cf(fs), ptr(lv['dimtmp']),
                                                    ptr(gflow), 1, B, N, h, w, 2, st), "image_warp_bwd(flow)")
                    gflow.add_(lv['dimtmp'])
            # ---- data terms
            if wt('ternary'):
                D = LAYER_PATCH_DISTANCES[i]
                check(lib.unflow_gray_pair(ptr(lv['im']), 3, ptr(flow), cf(fs), ptr(lv['gray1']), ptr(lv['gray2w']), B, N, h,
                                           w, st), "gray_pair")
                check(lib.unflow_ternary_fwd(ptr(lv['gray1']), ptr(lv['gray2w']), ptr(mask), n_mask, ptr(lv['dist']),
                                             ptr(self.loss_acc), cf(lw * wt('ternary')), cf(n1), D, N, h, w, st), "ternary")
                if with_grad:
                    check(lib.unflow_ternary_warp_bwd(ptr(lv['gray1']), ptr(lv['gray2w']), ptr(lv['dist']), ptr(lv['im']),
                                                      3, ptr(flow), cf(fs), ptr(gflow), 1, B, D, N, h, w, st),
                          "ternary_warp_bwd")
            if wt('photo'):
                check(lib.unflow_photometric_fwd_bwd(ptr(lv['im']), 3, ptr(flow), cf(fs), ptr(mask), n_mask,
                                                     ptr(self.loss_acc), gf, 1, cf(lw * wt('photo')), cf(n1 * 3), B, N, h,
                                                     w, st), "photometric")
            if wt('grad'):
                self._level_extra(lv)
                check(lib.unflow_image_warp_fwd(ptr(lv['im']), 3, ptr(flow), cf(fs), ptr(lv['imw']), ptr(None), B, N, h, w,
                                                3, st), "image_warp(im)")
                check(lib.unflow_gradient_loss_fwd(ptr(lv['im']), 3, ptr(lv['imw']), ptr(mask), n_mask, ptr(lv['gdiff']),
                                                   ptr(self.loss_acc), cf(lw * wt('grad')), cf(n1 * 6), N, h, w, st),
                      "gradient_loss")
                if with_grad:
                    check(lib.unflow_gradient_loss_bwd(ptr(lv['gdiff']), ptr(lv['dimw']), N, h, w, st), "gradient_loss_bwd")
                    check(lib.unflow_image_warp_bwd(ptr(lv['dimw']), ptr(lv['im']), 3, ptr(flow), cf(fs), ptr(None),
                                                    ptr(gflow), 1, B, N, h, w, 3, st), "image_warp_bwd(im)")
        if with_grad and not P.get('pyramid_loss'):
            for lv in self.lv[1:]:
                lv['gflow'].zero_()
        # regularisation term (value only; its gradient is fused into adam_step)
        if not self.defer_l2:
            check(lib.unflow_l2_loss(ptr(self.P), cl(self.n_weights), cf(L2_SCALE), ptr(self.loss_acc), st), "l2_loss")
        return self.loss_acc

    def _pyramid_levels(self, levels, wt):
        """ctypes array of unflow_pyr_level for unflow_loss_pyramid_default (rebuilt when the mask pointers change)."""
        Level = _lib.PyrLevel
        key = tuple(lv['mask'].data_ptr() for lv in levels) + (wt('ternary'), wt('smooth_2nd'))
        if self._pyr_cache is None or self._pyr_cache[0] != key:
            arr = (Level * len(levels))()
            for i, lv in enumerate(levels):
                n1 = self.B * lv['h'] * lv['w']
                lw = LAYER_WEIGHTS[i]
                a = arr[i]
                a.im, a.flow, a.gray1, a.gray2w = lv['im'].data_ptr(), lv['flow'].data_ptr(), lv['gray1'].data_ptr(), \
                    lv['gray2w'].data_ptr()
                a.mask, a.dist, a.d_flow = lv['mask'].data_ptr(), lv['dist'].data_ptr(), lv['gflow'].data_ptr()
                a.H, a.W, a.n_mask, a.max_distance = lv['h'], lv['w'], lv['n_mask'], LAYER_PATCH_DISTANCES[i]
                a.flow_scale = FLOW_SCALE / (2 ** i)
                f32 = np.float32    # the same fp32 quotient the per-level entry points form from (weight, normaliser)
                a.ternary_scale = float(f32(lw * wt('ternary')) / f32(n1))            # normaliser B*H*W*1 (losses.py:311-312)
                a.smooth_scale = float(f32(lw * wt('smooth_2nd')) / f32(n1 * 4))      # B*H*W*4 per flow channel
            self._pyr_cache = (key, arr)
        return self._pyr_cache[1]

    def _level_extra(self, lv):
        """Buffers only the non-default loss terms need (allocated on first use, before any graph capture)."""
        if 'maskN' in lv:
            return
        N, h, w = self.N, lv['h'], lv['w']
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.dev)
        lv.update(maskN=z(N, h, w), fwarped=z(N, h, w, 2), gwarped=z(N, h, w, 2), dimtmp=z(N, h, w, 2),
                  fscaled=z(N, h, w, 2), fwmap=z(N, h, w), imw=z(N, h, w, 3), gdiff=z(N, h, w, 6), dimw=z(N, h, w, 3))

    # ------------------------------------------------------------------ backward
    def backward_net(self, part=None):
        """Gradients of the trained (last) network; earlier stages are behind stop_gradient (flownet.py:51-54).
        part 0 / 1: the two halves used to overlap the data-parallel all-reduce (see grad_buckets)."""
        self.stages[-1].backward(part)
        if part in (None, 1):
            if self.train_all:
                for i in range(len(self.stages) - 1, 0, -1):
                    self._stack_backward(self.stages[i], self.stages[i - 1])
                    self.stages[i - 1].backward()
            self._bias_grads()

    def _stack_backward(self, st, prev):
        """d loss / d (flow2 of the previous network) through the stage input of `st` (train_all).  The previous
        network's coarser flows do not reach the loss directly (only flows[-1] enters it, unsupervised.py:82-83)."""
        for lvl in (3, 4, 5, 6):
            prev.grad['flow%d' % lvl].zero_()
        g2 = prev.grad['flow2']
        g2.zero_()
        pf = prev.act['flow2']
        dx = st.grad['x0s']
        check(_lib.lib().unflow_stack_input_bwd(ptr(dx), dx.shape[3], ptr(self.x0), ptr(pf), ptr(g2), self.B, self.N,
                                                self.H, self.W, pf.shape[1], pf.shape[2], cf(4 * FLOW_SCALE), stream()),
              "stack_input_bwd")

    def grad_buckets(self):
        """Flat ranges of self.G: (early, late).  `early` = the weights whose gradients are complete after
        backward_net(0) (conv4 .. flow2 of the trained network: a contiguous tail of the weight region);
        `late` = everything else (conv1 .. conv3_1 weights, all biases, and the never-written zeros of frozen stages)."""
        if self.train_all:      # earlier networks are still accumulating until the very end
            return [], [(0, self.n_params)]
        st = self.stages[-1]
        first = st.by_name['conv4']
        lo = first.dw.data_ptr() - self.G.data_ptr()
        lo //= 4
        return [(lo, self.n_weights)], [(0, lo), (self.n_weights, self.n_params)]

    def _bias_grads(self):
        """db = column sums of every layer's dz, batched (unflow_colsum_batched)."""
        import ctypes
        lib = _lib.lib()
        if self._bias_plan is None:
            lib.unflow_colsum_batched_workspace_bytes.restype = ctypes.c_size_t
            plan = []
            MAXB = 32        # descriptors per launch (MAX_COLSUM in csrc/conv_igemm.hip); train_all stacks have more layers
            for j0 in range(0, len(self._bias_jobs), MAXB):
                jobs = self._bias_jobs[j0:j0 + MAXB]
                n = len(jobs)
                xs = (ctypes.c_void_p * n)(*[dz.data_ptr() for dz, _ in jobs])
                lds = (ctypes.c_int * n)(*[dz.stride(2) for dz, _ in jobs])
                npx = (ctypes.c_long * n)(*[dz.shape[0] * dz.shape[1] * dz.shape[2] for dz, _ in jobs])
                cs = (ctypes.c_int * n)(*[l.cout for _, l in jobs])
                outs = (ctypes.c_void_p * n)(*[l.db.data_ptr() for _, l in jobs])
                nbytes = lib.unflow_colsum_batched_workspace_bytes(n, cs)
                ws = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=self.dev)
                plan.append((n, xs, lds, npx, cs, outs, ws))
            self._bias_plan = plan
        for n, xs, lds, npx, cs, outs, ws in self._bias_plan:
            check(lib.unflow_colsum_batched(n, xs, lds, npx, cs, outs, ptr(ws), _lib.csz(ws.numel() * 4), stream()),
                  "colsum_batched")

    # ------------------------------------------------------------------ optimiser
    def adam_step(self, lr, grad_scale=1.0, beta1=0.9, beta2=0.999, eps=1e-8):
        """tf.train.AdamOptimizer(beta1=0.9, beta2=0.999) update (train.py:151-152), TF formulation, with the
        slim.l2_regularizer(0.0004) gradient added for the weight tensors (biases are not regularised)."""
        self.step_count += 1
        t = self.step_count
        lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
        if self.defer_l2:   # the loss of THIS step gets its regularisation term from the pre-update parameters here
            check(_lib.lib().unflow_adam_step_regloss(ptr(self.P), ptr(self.G), ptr(self.M), ptr(self.V),
                                                      cl(self.n_params), cl(self.n_weights), cf(grad_scale), cf(L2_SCALE),
                                                      cf(lr_t), cf(beta1), cf(beta2), cf(eps), ptr(self.loss_acc),
                                                      stream()), "adam")
            return
        check(_lib.lib().unflow_adam_step(ptr(self.P), ptr(self.G), ptr(self.M), ptr(self.V), cl(self.n_params),
                                          cl(self.n_weights), cf(grad_scale), cf(L2_SCALE), cf(lr_t), cf(beta1),
                                          cf(beta2), cf(eps), stream()), "adam")

    # ------------------------------------------------------------------ composite
    def fwd_bwd(self, im1=None, im2=None):
        if im1 is not None:
            self.set_input(im1, im2)
        self.forward_net()
        loss = self.forward_loss(with_grad=True)
        self.backward_net()
        return loss

    def train_step(self, im1, im2, lr):
        """One optimisation step; returns the loss tensor [1] (complete once the Adam kernel has run: the 0.0004*sum(w^2)/2
        term is accumulated by the pass Adam makes over the parameters, not by a separate reduction)."""
        prev, self.defer_l2 = self.defer_l2, True
        try:
            loss = self.fwd_bwd(im1, im2)
            self.adam_step(lr)
        finally:
            self.defer_l2 = prev
        return loss

    def final_flows(self):
        """final_flow_fw / _bw (unsupervised.py:103-104): resize_bilinear(flow2, im_shape) * 5 * 4."""
        f2 = self.act['flow2']
        N, h, w, _ = f2.shape
        check(_lib.lib().unflow_resize_bilinear_tf1(ptr(f2), ptr(self.final_flow), N, h, w, 2, self.H, self.W,
                                                    cf(FLOW_SCALE * 4), stream()), "resize_bilinear")
        return self.final_flow[:self.B], self.final_flow[self.B:]

    def flows(self):
        """(flows_fw, flows_bw): lists [flow2..flow6], NHWC, like flownet(..., backward_flow=True)[-1]."""
        B = self.B
        fw = [self.act['flow%d' % l][:B] for l in (2, 3, 4, 5, 6)]
        bw = [self.act['flow%d' % l][B:] for l in (2, 3, 4, 5, 6)]
        return fw, bw


def flow_error_avg(flow_1, flow_2, mask=None):
    """core/flow_util.py:98-103 (EPE)."""
    f1, f2 = flow_1.contiguous(), flow_2.contiguous()
    out = torch.empty(2, dtype=torch.float32, device=f1.device)
    npix = f1.numel() // 2
    check(_lib.lib().unflow_flow_error_sums(ptr(f1), ptr(f2), ptr(None if mask is None else mask.contiguous()),
                                            ptr(out), cl(npix), stream()), "flow_error")
    return out[0] / out[1]


FlowNetCEngine = FlowNetEngine   # the FlowNetC training step is the default spec ('C')
