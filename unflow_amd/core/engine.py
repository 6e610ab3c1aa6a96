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
from . import layers as L

FLOW_SCALE = 5.0                                   # flownet.py:11
CHANNEL_MEAN = [104.920005, 110.1753, 114.785955]  # core/input.py:45
LAYER_WEIGHTS = [12.7, 4.35, 3.9, 3.4, 1.1]        # unsupervised.py:87
LAYER_PATCH_DISTANCES = [3, 2, 2, 1, 1]            # unsupervised.py:88
LAYER_WEIGHTS_FULL_RES = [12.7, 5.5, 5.0, 4.35, 3.9, 3.4, 1.1]     # unsupervised.py:90
LAYER_PATCH_DISTANCES_FULL_RES = [3, 3] + LAYER_PATCH_DISTANCES     # unsupervised.py:91
L2_SCALE = 0.0004                                  # flownet.py:176
LOSSES = ['occ', 'sym', 'fb', 'grad', 'ternary', 'photo', 'smooth_1st', 'smooth_2nd']  # unsupervised.py:15

DEFAULT_PARAMS = dict(flownet='C', pyramid_loss=True, border_mask=True, ternary_weight=1.0, smooth_2nd_weight=3.0)


def pad4(c):
    return (c + 3) // 4 * 4


def round8(c):
    return (c + 7) // 8 * 8


def conv_math_mode():
    """How the conv / conv_transpose layers multiply (UNFLOW_CONV_MATH):
       'bf16x3'  (default) operands pre-split into three bf16 planes by their producers, six bf16 MFMA terms, fp32
                 accumulate — fp32-class accuracy (csrc/conv_planes.hip);
       'fp32'    v_mfma_f32_32x32x2_f32 everywhere (csrc/conv_igemm.hip);
       'f16'     fp16 activations and weights in, fp32 accumulate (BASELINE configs[4]); NOT fp32-equivalent.
    (Round 6 removed the fourth mode, 'bf16x3_inline' — the engine without planes on the in-staging split of
    csrc/conv_igemm.hip: no user.  Those kernels remain what the plain fp32-tensor entry points of the C ABI run,
    unflow_conv2d_fwd & co., which have no planes to take.)"""
    m = os.environ.get('UNFLOW_CONV_MATH', 'bf16x3')
    if m not in ('bf16x3', 'fp32', 'f16'):
        raise ValueError("UNFLOW_CONV_MATH must be bf16x3, fp32 or f16")
    return m


RGB4_FORM = True     # two-pixel K granules for FlowNetC's first layer (csrc/conv_planes.hip rgb4_form)


class Layer:
    """One conv / conv_transpose variable pair.  in_map: [(physical lo, tf lo, n)] — where the reference's input channels
    live in the (padded, segment-aligned) physical input of the layer; cout_p >= cout is the physical output width
    (pad columns keep zero weights / bias and receive exactly zero gradient)."""
    __slots__ = ('name', 'kind', 'k', 'cin', 'cout', 'stride', 'act', 'cin_p', 'cout_p', 'in_map', 'w', 'b', 'dw', 'db', 'mw', 'vw', 'mb', 'vb',
                 'wpl_d', 'wpl_t')

    def __init__(self, name, kind, k, cin, cout, stride, act, in_map=None, cin_p=None):
        self.name, self.kind, self.k, self.cin, self.cout, self.stride, self.act = name, kind, k, cin, cout, stride, act
        self.in_map = in_map or [(0, 0, cin)]
        if cin_p is None:
            cin_p = cin if (kind == 'deconv' and cin == 2) else pad4(max(lo + n for lo, _, n in self.in_map))
        self.cin_p = cin_p
        self.cout_p = cout if cout <= 2 else pad4(cout)
        self.wpl_d = self.wpl_t = None

    def wshape(self):
        if self.kind == 'conv':
            return (self.k, self.k, self.cin_p, self.cout_p)   # HWIO
        return (self.k, self.k, self.cout_p, self.cin_p)       # conv2d_transpose: [k,k,out,in]

    def tf_wshape(self):
        return (self.k, self.k, self.cin, self.cout) if self.kind == 'conv' else (self.k, self.k, self.cout, self.cin)

    def wplane_view(self):
        """(taps, R, Cc) the weight planes are built from.  FlowNetC's first layer (7x7 stride 2 over RGB0) is read as
        [7 tap rows][28 = 7 kx * 4 channels][Cout]: the two-pixel-granule form of csrc/conv_planes.hip (rgb4_form)."""
        if self.kind == 'conv' and self.k == 7 and self.stride == 2 and self.cin_p == 4 and RGB4_FORM:
            return (7, 28, self.cout_p)
        return (self.k * self.k, self.wshape()[2], self.wshape()[3])

    def uses_planes(self):
        return self.cout > 4 and self.cin_p >= 4


def concat_layout(parts):
    """Physical channel layout of a tf.concat: every part starts at a multiple of 4 (16-byte aligned fp32 slices, 8-byte
    aligned plane slices).  parts: [(name, channels)] -> ({name: (lo, n)}, physical width)."""
    off, d = 0, {}
    for nm, n in parts:
        off = pad4(off)
        d[nm] = (off, n)
        off += n
    return d, pad4(off)


class ActView(dict):
    """name -> fp32 activation of a stage.  Tensors that only convolutions read (_Stage.planes_only) are not written in fp32
    by the step: for those the values are rebuilt from the operand planes on access (hi + mid + lo is the fp32 value exactly;
    fp16 mode: the rounded value) — for tests and tools, never on the step's path."""

    def __init__(self, stage):
        super().__init__()
        self._stage = stage

    def __getitem__(self, k):
        st = self._stage
        if k in st.planes_only:
            pl = st.A[k].pl
            C = st.A[k].t.shape[-1]
            dt = torch.bfloat16 if pl.shape[0] == 3 else torch.float16
            return pl.view(dt).float().sum(0)[..., :C].contiguous()
        return super().__getitem__(k)


class _Op:
    """One forward launch of a stage: a layer (src slice -> dst slice) or the correlation."""
    __slots__ = ('kind', 'l', 'src', 'dst')

    def __init__(self, kind, l, src, dst):
        self.kind, self.l, self.src, self.dst = kind, l, src, dst


class _Stage:
    """One network of the (possibly stacked) spec: layer table, buffers, forward launch list and the backward list derived
    from it.  kind: 'C' / 'S' (full width) or 'c' / 's' (3/8 width, flownet.py:22-23); full_res adds deconv1/0 and
    flow1/0 (flownet.py:133-153; FlowNetS only: flownet_c never passes conv1 / inputs, flownet.py:231-233)."""

    def __init__(self, eng, kind, index, full_res=False):
        self.eng, self.kind, self.index, self.full_res = eng, kind, index, full_res
        self.is_c = kind in 'Cc'
        m = 1.0 if kind in 'CS' else 3.0 / 8.0
        c = lambda x: int(x * m)                                            # noqa: E731
        self.c = c
        scope = '' if index == 0 else 'stack_%d_flownet/' % index           # flownet.py:72-77
        self.in_ch = 3 if self.is_c else (6 if index == 0 else 14)
        self.trainable = True
        if full_res and self.is_c:
            raise ValueError("full_res needs a FlowNetS as the last network: flownet_c passes neither conv1 nor the inputs "
                             "to the refinement decoder (flownet.py:231-233)")
        # ---- buffers: name -> (spatial divisor, {segment: (lo, n)}, physical width)
        B = self.bufs = OrderedDict()

        def buf(name, div, parts):
            lay, width = concat_layout(parts)
            B[name] = (div, lay, 2 if name.startswith('flow') else width)   # flow outputs: exactly [.., 2]
        if full_res:
            buf('cat0', 1, [('in', self.in_ch), ('deconv0', c(16)), ('up', 2)])
            buf('cat1', 2, [('conv1', c(64)), ('deconv1', c(32)), ('up', 2)])
        else:
            if not self.is_c:
                buf('x0s', 1, [('in', self.in_ch)])
            buf('c1', 2, [('conv1', c(64))])
        buf('cat2', 4, [('conv2', c(128)), ('deconv2', c(64)), ('up', 2)])
        buf('c3', 8, [('conv3', c(256))])
        if self.is_c:
            buf('catc', 8, [('conv_redir', c(32)), ('corr', 441)])
        buf('cat3', 8, [('conv3_1', c(256)), ('deconv3', c(128)), ('up', 2)])
        buf('c4', 16, [('conv4', c(512))])
        buf('cat4', 16, [('conv4_1', c(512)), ('deconv4', c(256)), ('up', 2)])
        buf('c5', 32, [('conv5', c(512))])
        buf('cat5', 32, [('conv5_1', c(512)), ('deconv5', c(512)), ('up', 2)])
        buf('c6', 64, [('conv6', c(1024))])
        buf('c6_1', 64, [('conv6_1', c(1024))])
        self.flow_levels = ([0, 1] if full_res else []) + [2, 3, 4, 5, 6]
        for lvl in self.flow_levels:
            buf('flow%d' % lvl, 2 ** lvl, [('flow', 2)])
        self.b1 = 'cat1' if full_res else 'c1'
        self.bin = None if self.is_c else ('cat0' if full_res else 'x0s')   # stage input buffer (FlowNetC reads eng.x0)
        # ---- layers + forward ops (the reference's variable creation order)
        self.layers, self.ops = [], []
        f = scope + ('flownet_c_features/' if self.is_c else 'flownet_s/')
        d = scope + ('flownet_c/' if self.is_c else 'flownet_s/')

        def seg(bname, sname):
            lo, n = B[bname][1][sname]
            return (bname, lo, lo + n)

        def whole(bname):
            return (bname, 0, B[bname][2])

        def layer(pre, name, kind, k, src, dst, cout, stride=1, act=True):
            sb, slo, shi = src
            if sb == 'x0':
                in_map, cin, cin_p = [(0, 0, 3)], 3, 4
            else:
                lay = B[sb][1]
                segs = sorted((lo, n) for lo, n in lay.values() if lo >= slo and lo + n <= shi)
                in_map, t = [], 0
                for lo, n in segs:
                    in_map.append((lo - slo, t, n))
                    t += n
                cin, cin_p = t, pad4(shi - slo)
                if kind == 'deconv' and cin == 2:
                    cin_p = 2
            l = Layer(pre + name, kind, k, cin, cout, stride, act, in_map, cin_p)
            self.layers.append(l)
            self.ops.append(_Op('layer', l, src, dst))
            return l

        x_in = ('x0', 0, 4) if self.is_c else (self.bin, 0, pad4(self.in_ch))
        layer(f, 'conv1', 'conv', 7, x_in, seg(self.b1, 'conv1'), c(64), 2)
        layer(f, 'conv2', 'conv', 5, seg(self.b1, 'conv1'), seg('cat2', 'conv2'), c(128), 2)
        layer(f, 'conv3', 'conv', 5, seg('cat2', 'conv2'), seg('c3', 'conv3'), c(256), 2)
        if self.is_c:
            # (conv_redir before the correlation: the backward list is the reverse, and the correlation's gradient kernel
            # writes d c3 while conv_redir's data gradient accumulates into it and applies conv3's leaky derivative)
            layer(d, 'conv_redir', 'conv', 1, seg('c3', 'conv3'), seg('catc', 'conv_redir'), c(32), 1)
            self.ops.append(_Op('corr', None, seg('c3', 'conv3'), seg('catc', 'corr')))
            layer(d, 'conv3_1', 'conv', 3, whole('catc'), seg('cat3', 'conv3_1'), c(256), 1)
        else:
            layer(d, 'conv3_1', 'conv', 3, seg('c3', 'conv3'), seg('cat3', 'conv3_1'), c(256), 1)
        layer(d, 'conv4', 'conv', 3, seg('cat3', 'conv3_1'), seg('c4', 'conv4'), c(512), 2)
        layer(d, 'conv4_1', 'conv', 3, seg('c4', 'conv4'), seg('cat4', 'conv4_1'), c(512), 1)
        layer(d, 'conv5', 'conv', 3, seg('cat4', 'conv4_1'), seg('c5', 'conv5'), c(512), 2)
        layer(d, 'conv5_1', 'conv', 3, seg('c5', 'conv5'), seg('cat5', 'conv5_1'), c(512), 1)
        layer(d, 'conv6', 'conv', 3, seg('cat5', 'conv5_1'), seg('c6', 'conv6'), c(1024), 2)
        layer(d, 'conv6_1', 'conv', 3, seg('c6', 'conv6'), seg('c6_1', 'conv6_1'), c(1024), 1)
        # refinement decoder (_flownet_upconv, flownet.py:89-153)
        prev = 'c6_1'
        for lvl, cat, dch in ((6, 'cat5', c(512)), (5, 'cat4', c(256)), (4, 'cat3', c(128)), (3, 'cat2', c(64))):
            up = lvl - 1
            layer(d, 'flow%d' % lvl, 'conv', 3, whole(prev), seg('flow%d' % lvl, 'flow'), 2, 1, False)
            layer(d, 'deconv%d' % up, 'deconv', 4, whole(prev), seg(cat, 'deconv%d' % up), dch, 2, True)
            layer(d, 'flow%d_up%d' % (lvl, up), 'deconv', 4, seg('flow%d' % lvl, 'flow'), seg(cat, 'up'), 2, 2, False)
            prev = cat
        layer(d, 'flow2', 'conv', 3, whole('cat2'), seg('flow2', 'flow'), 2, 1, False)
        if full_res:
            fr = d + 'full_res/'
            layer(fr, 'deconv1', 'deconv', 4, whole('cat2'), seg('cat1', 'deconv1'), c(32), 2, True)
            layer(fr, 'flow2_up1', 'deconv', 4, seg('flow2', 'flow'), seg('cat1', 'up'), 2, 2, False)
            layer(fr, 'flow1', 'conv', 3, whole('cat1'), seg('flow1', 'flow'), 2, 1, False)
            layer(fr, 'deconv0', 'deconv', 4, whole('cat1'), seg('cat0', 'deconv0'), c(16), 2, True)
            layer(fr, 'flow1_up0', 'deconv', 4, seg('flow1', 'flow'), seg('cat0', 'up'), 2, 2, False)
            layer(fr, 'flow0', 'conv', 3, whole('cat0'), seg('flow0', 'flow'), 2, 1, False)
        self.by_name = {l.name.split('/')[-1]: l for l in self.layers}
        self._plan_backward()

    # -------------------------------------------------------------- backward plan
    def _plan_backward(self):
        """Reverse of the forward list.  For every (consumer, input buffer) decide: accumulate or overwrite (the first data
        gradient written into a buffer covers all of it), and the channel range whose leaky-ReLU derivative this call
        applies — the segments of activated producers for which this consumer is the LAST one to add its share
        (flownet.py:84-86: the epilogue multiplies the finished sum)."""
        producers = {}                                      # (buf, lo, hi) -> producing layer
        for op in self.ops:
            if op.kind == 'layer':
                producers[op.dst] = op.l
        rev = list(reversed(self.ops))
        # consumers of every buffer in backward processing order
        order = {}
        for i, op in enumerate(rev):
            if op.src[0] != 'x0':
                order.setdefault(op.src[0], []).append((i, op))
        self.bwd = []
        for i, op in enumerate(rev):
            sb, slo, shi = op.src
            act_lo = act_hi = 0
            first = False
            if sb != 'x0':
                cons = order[sb]
                first = cons[0][0] == i
                if first:
                    assert slo == 0 and shi == self.bufs[sb][2], "first data gradient into %s must cover it" % sb
                rng = []
                for (pb, plo, phi), pl in producers.items():
                    if pb != sb or not pl.act or plo < slo or phi > shi:
                        continue
                    last = max(j for j, o in cons if o.src[1] <= plo and o.src[2] >= phi)
                    if last == i:
                        rng.append((plo - slo, phi - slo))
                if rng:
                    rng.sort()
                    for (a0, a1), (b0, b1) in zip(rng, rng[1:]):
                        assert a1 == b0, "activation ranges of one data gradient must be contiguous"
                    act_lo, act_hi = rng[0][0], rng[-1][1]
            self.bwd.append((op, first, act_lo, act_hi))
        names = [op.l.name.split('/')[-1] if op.kind == 'layer' else 'corr' for op, _, _, _ in self.bwd]
        self.bwd_names = names
        self.split_at = names.index('conv4') + 1            # part 0 = [0, split_at): decoder + conv6_1 .. conv4
        self.part_bounds = [0, self.split_at, len(self.bwd)]

    def set_parts(self, last_layers):
        """Cut the backward list after each of the named layers (in backward order), e.g. ('conv6', 'conv4'):
        part 0 = decoder + conv6_1 + conv6, part 1 = conv5_1 .. conv4, part 2 = the rest."""
        cuts = [self.bwd_names.index(n) + 1 for n in last_layers]
        assert cuts == sorted(cuts) and (not cuts or cuts[-1] < len(self.bwd))
        self.part_bounds = [0] + cuts + [len(self.bwd)]

    def part_weight_range(self, part):
        """Flat range of eng.G holding the filter gradients that are complete after backward(part) (the layers of a part
        are a contiguous run of the forward order, hence of the flat weight region)."""
        ls = [op.l for op, _, _, _ in self.bwd[self.part_bounds[part]:self.part_bounds[part + 1]] if op.kind == 'layer']
        G = self.eng.G
        lo = min(l.dw.data_ptr() for l in ls) - G.data_ptr()
        hi = max(l.dw.data_ptr() + l.dw.numel() * 4 for l in ls) - G.data_ptr()
        return lo // 4, hi // 4

    # -------------------------------------------------------------- buffers
    def alloc(self):
        e = self.eng
        N, H, W, dev = e.N, e.H, e.W, e.dev
        npl = e.n_planes
        self.A, self.Gd = {}, {}
        for name, (div, lay, width) in self.bufs.items():
            is_flow = name.startswith('flow')
            self.A[name] = L.PT.alloc((N, H // div, W // div, width), dev, 0 if is_flow else npl)
        # d loss / d stage input -> the previous network (train_all only; otherwise behind stop_gradient, flownet.py:51-54)
        self.need_in_grad = (not self.is_c) and self.index > 0 and e.train_all
        if self.need_in_grad and self.full_res:
            raise NotImplementedError("train_all through a full_res refinement stage")
        if self.trainable:
            for name, (div, lay, width) in self.bufs.items():
                if name == 'x0s' and not self.need_in_grad:
                    continue
                is_flow = name.startswith('flow')
                self.Gd[name] = L.PT.alloc((N, H // div, W // div, width), dev, 0 if is_flow else npl, scale=e.grad_plane_scale)
        # single-producer conv outputs that only convolutions consume (no flow head, no concat, no correlation fallback): they
        # live as operand planes alone — no fp32 write in the forward pass, and the data gradient that needs the sign of
        # the activation (leaky-ReLU derivative) takes it from the leading plane
        self.planes_only = set()
        if npl:
            self.planes_only = {n for n in ('c1', 'c4', 'c5', 'c6') + (() if self.is_c else ('c3',)) if n in self.A}
        self.act = ActView(self)
        self.act.update({k: v.t for k, v in self.A.items()})
        self.grad = {k: v.t for k, v in self.Gd.items()}
        if self.full_res:      # the stage input is the first segment of concat0
            self.act['x0s'] = self.act['cat0'][..., :pad4(self.in_ch)]

    def pt(self, spec, grad=False):
        """PT view of (buffer, lo, hi); 'x0' is the engine's network input."""
        b, lo, hi = spec
        if b == 'x0':
            return self.eng.X0
        src = (self.Gd if grad else self.A)[b]
        return src if (lo == 0 and hi == src.t.shape[-1]) else src.sl(lo, hi)

    # -------------------------------------------------------------- forward
    def forward(self, prev_flow2=None):
        e = self.eng
        B, N = e.B, e.N
        if not self.is_c:
            x0s = self.act['x0s']
            pf = prev_flow2
            check(_lib.lib().unflow_stack_input(ptr(e.x0), ptr(pf), ptr(x0s), x0s.stride(2), B, N, e.H, e.W,
                                                0 if pf is None else pf.shape[1], 0 if pf is None else pf.shape[2],
                                                cf(4 * FLOW_SCALE), e.stream()), "stack_input")
            inp = self.A[self.bin]
            if inp.pl is not None:
                w_in = pad4(self.in_ch)
                L.planes_from_f32(x0s, inp.pl[..., :round8(w_in)], C=w_in)
        for op in self.ops:
            if op.kind == 'corr':
                c3, out = self.pt(op.src), self.pt(op.dst)
                C = op.src[2] - op.src[1]
                h8, w8 = e.H // 8, e.W // 8
                c3pl = _lib.planes_of(self._corr_planes(c3, refresh=True))   # bf16 x 3 planes: the matrix-core path
                check(_lib.lib().unflow_correlation_nhwc_fwd_pl(ptr(c3.t), ptr(c3.t), c3.t.stride(2), c3pl, c3pl, B,
                                                                ptr(out.t), out.t.stride(2), N, C, h8, w8, 1, 20, 20, 1, 2,
                                                                e.stream()), "correlation")
                if out.pl is not None:       # operand planes of the cost volume for conv3_1 (pad channels zeroed)
                    L.planes_from_f32(out.t, out.pl)
                continue
            l = op.l
            x, y = self.pt(op.src), self.pt(op.dst)
            if l.cout_p != l.cout:
                y = L.PT(y.t.as_strided(y.t.shape[:3] + (l.cout_p,), y.t.stride(), y.t.storage_offset()),
                         None if y.pl is None else y.pl)
            po = op.dst[0] in self.planes_only
            if l.kind == 'conv':
                L.conv_fwd(x, l.w, l.wpl_t, l.b, y, l.stride, l.act, planes_only=po)
            else:
                L.deconv_fwd(x, l.w, l.wpl_d, l.b, y, l.act, planes_only=po)

    # -------------------------------------------------------------- backward
    def _corr_planes(self, c3, refresh=False):
        """The bf16 x 3 operand planes of the correlation's feature maps.  bf16x3 mode: the planes conv3 wrote.  fp16 mode: the
        correlation stays fp32-equivalent (its kernels take bf16 x 3 planes only), so the fp32 features are split once per
        forward pass into a side buffer — the planes kernels instead of the fp32-MFMA ones (B = 8: 284 + 547 us)."""
        e = self.eng
        if e.n_planes == 3:
            return c3.pl
        if e.n_planes != 1:
            return None
        if getattr(self, '_c3_b3', None) is None:
            n, h, w, c = c3.t.shape
            self._c3_b3 = torch.zeros(3, n, h, w, (c + 7) // 8 * 8, dtype=torch.int16, device=c3.t.device)
        if refresh:
            L.planes_from_f32(c3.t, self._c3_b3)
        return self._c3_b3

    def backward(self, part=None, before_join=None):
        """part None: everything; k: the k-th slice of the backward list (default cuts: 0 = decoder + conv6_1..conv4 — 94 %
        of the parameters, whose gradients are final afterwards so that their all-reduce can start — 1 = conv3_1 ..
        conv1; set_parts() changes the cuts)."""
        e = self.eng
        lo, hi = (0, len(self.bwd)) if part is None else (self.part_bounds[part], self.part_bounds[part + 1])
        B, N = e.B, e.N
        # Filter gradients leave the critical path: a layer's dW depends only on its (final) dz and its input, and nothing
        # reads it before the optimizer, so groups of them run on a second stream beside the data-gradient chain (own
        # split-K scratch, ws_slot 3); the part joins before it returns.  The small deep layers, whose kernels cannot fill
        # the chip alone, are where this pays (+2.9 % on the benchmarked step; +0.5 % more with the Cout = 2 layers — flow
        # heads, 2 -> 2 flow upsamplers — in the groups too, wgrad_inline_tiny = False).  History: those layers used to stay
        # on the main stream because the 2 -> 2 filter gradient of a replayed two-branch graph returned one wrong sum in
        # ~10 % of the replays whenever a kernel of the other branch shared its SIMD; root cause: a compiler-formed
        # v_pk_mul_f32 whose destination pair overlaps an op_sel-selected source (unflow_amd/build.py: the library is now
        # built with -fno-slp-vectorize; tools/debug/wgstream_flake.py reproduces it with the old flags: 0 wrong sums in 120
        # replays now, with every filter gradient on the second stream).  tests/test_engine_gpu.py asserts bit-identity of the
        # schedules over repeated replays.
        side = e.wgrad_stream
        main = torch.cuda.current_stream(e.dev)
        pending = []
        counter = [100 * (part or 0)]

        def flush():
            if not pending:
                return
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for fn, a in pending:
                    with L.ws_slot(3 + (counter[0] if e.wgrad_unique_ws else 0)):
                        fn(*a)
                    counter[0] += 1
            pending.clear()
            if e.wgrad_sync_each:
                main.wait_stream(side)

        # the Cout = 2 layers (flow heads, 2 -> 2 flow upsamplers): their filter gradients leave as ONE batch as soon as the
        # last of them has been visited (all their gradients are final then)
        flow_jobs = []
        n_flow = sum(1 for op, _, _, _ in self.bwd[lo:hi] if op.kind == 'layer' and op.l.cout <= 2) if e.batch_flow_wgrad else 0

        for op, first, act_lo, act_hi in self.bwd[lo:hi]:
            if op.kind == 'corr':
                c3, g3, gout = self.pt(op.src), self.pt(op.src, True), self.pt(op.dst, True)
                C = op.src[2] - op.src[1]
                h8, w8 = e.H // 8, e.W // 8
                assert first
                c3pl = _lib.planes_of(self._corr_planes(c3))                 # bf16 x 3 planes: feature operand by LDS-DMA
                check(_lib.lib().unflow_correlation_nhwc_bwd_pl(ptr(gout.t), gout.t.stride(2), ptr(c3.t), ptr(c3.t),
                                                                c3.t.stride(2), c3pl, c3pl, B, ptr(g3.t), ptr(None),
                                                                g3.t.stride(2), 1, N, C, h8, w8, 1, 20, 20, 1, 2, e.stream()),
                      "correlation_grad")
                continue
            l = op.l
            x, dz = self.pt(op.src), self.pt(op.dst, True)
            if l.cout_p != l.cout:
                dz = L.PT(dz.t.as_strided(dz.t.shape[:3] + (l.cout_p,), dz.t.stride(), dz.t.storage_offset()), dz.pl, dz.scale)
            if e._bias_plan is None:
                e._bias_jobs.append((dz.t, l))       # bias gradients: one batched column-sum launch at the end
            xv = op.src[0] not in self.planes_only      # a planes-only source has no valid fp32 copy: the filter gradient must take the planes
            wg = (L.conv_bwd_filter, (x, dz, l.dw, l.stride, xv)) if l.kind == 'conv' else (L.deconv_bwd_filter, (x, dz, l.dw, xv))
            if n_flow and l.cout <= 2:
                flow_jobs.append((l.kind, x, dz, l.dw))
                wg = (L.flow_wgrad_batched, (flow_jobs,)) if len(flow_jobs) == n_flow else None
            if wg is None:
                pass
            elif side is None or (e.wgrad_inline_tiny and l.cout <= 2):
                wg[0](*wg[1])
            else:
                pending.append(wg)
                if len(pending) >= e.wgrad_group:
                    flush()
            sb = op.src[0]
            is_input = sb == 'x0' or (sb == self.bin and op.src[1] == 0 and op.src[2] == pad4(self.in_ch)
                                      and op.l.name.endswith('conv1'))
            if sb not in self.Gd or (is_input and not self.need_in_grad):
                continue                              # inputs are data (or behind stop_gradient, flownet.py:51-54)
            dx = self.pt(op.src, True)
            # the loss wrote d flowN first (flow buffers): everything after it accumulates
            accumulate = (not first) or sb.startswith('flow')
            act_src = self.pt(op.src) if act_hi > act_lo else None
            apl = sb in self.planes_only
            if l.kind == 'conv':
                L.conv_bwd_data(dz, l.w, l.wpl_d, dx, l.stride, accumulate, act_src, act_lo, act_hi, act_planes=apl)
            else:
                L.deconv_bwd_data(dz, l.w, l.wpl_t, dx, accumulate, act_src, act_lo, act_hi, act_planes=apl)
        if side is not None:
            flush()
        if before_join is not None:
            before_join()            # HBM-bound work for the main stream while the filter gradients finish on the second
        if side is not None:
            main.wait_stream(side)


class FlowNetEngine:
    """Bidirectional forward / loss / backward / Adam of a FlowNet spec on one GPU, fixed (B, H, W).
    params['flownet']: a string over 'C', 'S' (full width) and 'c', 's' (3/8 width), a correlation net only first
    (flownet.py:14-81), e.g. 'C', 'S', 'CS', 'CSS', 'css'.  In a stack only the last network is trained unless
    params['train_all'] (flownet.py:51-54, train.py:29-37); frozen stages run forward only and — exactly as in the
    reference, whose regulariser and optimizer span all variables — still receive the L2 gradient in the Adam update.
    With train_all the gradient also flows back through every inter-stage input (upsampled flow, warp, |warp - first|)
    into the earlier networks.  params['full_res'] adds the full-resolution decoder levels to the LAST network
    (flownet.py:21,133-153; a FlowNetS) and the loss pyramid then has 7 levels (unsupervised.py:89-96)."""

    def __init__(self, batch, height, width, params=None, device=None, seed=0, layout_only=False):
        """layout_only: build the layer table and the flat parameter / gradient buffers (on `device`, which may then be the
        CPU) but no activations — for tools and the data-parallel tests that only need the flat layout and its buckets."""
        assert height % 64 == 0 and width % 64 == 0, "FlowNet needs H, W divisible by 64"
        self.params = dict(DEFAULT_PARAMS) if params is None else dict(params)
        if self.params.get('mask_occlusion', '') not in ('', 'fb', 'disocc'):   # unsupervised.py:125-126
            raise ValueError("mask_occlusion must be one of 'fb', 'disocc', ''")
        spec = self.params.get('flownet', 'C')
        if not spec or any(ch not in 'CScs' for ch in spec) or any(ch in 'Cc' for ch in spec[1:]):
            raise ValueError("flownet spec: 'C'/'c' or 'S'/'s' first, then 'S'/'s' refinement nets (flownet.py:20-28)")
        self.train_all = bool(self.params.get('train_all')) and len(spec) > 1     # train.py:29-37, flownet.py:51-54
        self.full_res = bool(self.params.get('full_res'))
        self.spec = spec
        self.B, self.H, self.W = batch, height, width
        self.N = 2 * batch
        self.dev = torch.device('cuda:0') if device is None else torch.device(device)
        self.math = conv_math_mode()
        # filter gradients on a second stream, in groups of wgrad_group layers (UNFLOW_WGRAD_GROUP=0: inline)
        self.wgrad_group = int(os.environ.get('UNFLOW_WGRAD_GROUP', '6'))
        self.wgrad_stream = None
        self.planes_external = False      # True: the captured forward does not re-split the weights (StepRunner does, per bucket)
        self.wgrad_sync_each = False      # debug: join after every group (no concurrency, still two graph branches)
        self.wgrad_unique_ws = False      # debug: one scratch buffer per deferred filter gradient
        # True: the Cout = 2 layers' filter gradients (flow heads, 2 -> 2 upsamplers: small latency-bound kernels) stay on the main
        # stream; False (default): they join the groups on the second stream like every other filter gradient
        self.wgrad_inline_tiny = False
        self.batch_flow_wgrad = True      # all Cout = 2 filter gradients of a decoder in one batched launch pair
        self.n_planes = {'bf16x3': 3, 'f16': 1}.get(self.math, 0)
        # UNFLOW_FUSED_ADAM=1: L2 + Adam and the re-split of the updated weights in one pass (csrc/conv_planes.hip
        # adam_planes_kernel; round 6).  Built, bit-identical, and NOT faster: 267.7 / 269.5 us against
        # 264.5 / 269.2 us for adam_kernel + weight_planes_kernel at FlowNetC's 39.2 M parameters (tools/debug/adam_fused_time.py;
        # the step 671.5 / 673.5 against 672.6 / 672.0 pairs/s): the 157 MB of P the second launch re-reads come from the
        # Infinity Cache, and the fused kernel's 64 x 64 tiles stream the four flat buffers in 256-byte row pieces (5.8 TB/s)
        # where the flat kernel runs at 6.5.  Default off.
        self.fused_adam = os.environ.get('UNFLOW_FUSED_ADAM', '0') == '1'
        # fp16 mode: the fp16 planes of GRADIENT tensors hold 2^12 x the gradient (producers scale, consumers divide their
        # fp32 sums): activation gradients of the deep layers are 1e-7 .. 1e-4, below fp16's normal range (6e-5); bf16 x 3
        # planes have fp32's exponent range and are never scaled
        self.grad_plane_scale = 4096.0 if self.math == 'f16' else 0.0
        if layout_only:
            self.n_planes = 0
        self.fused_adam = self.fused_adam and self.n_planes > 0
        import contextlib
        with (torch.cuda.device(self.dev) if self.dev.type == 'cuda' else contextlib.nullcontext()):
            self.stages = [_Stage(self, k, i, self.full_res and i == len(spec) - 1) for i, k in enumerate(spec)]
            for st in self.stages[:-1]:
                st.trainable = self.train_all
            self.layers = [l for st in self.stages for l in st.layers]
            self.by_name = self.stages[-1].by_name
            self._alloc_params()
            if not layout_only:
                self._alloc_activations()
                if self.wgrad_group > 0 and self.dev.type == 'cuda':
                    self.wgrad_stream = torch.cuda.Stream(self.dev)
                self._build_masks()
        self.step_count = 0
        self._bias_jobs, self._bias_plan = [], None
        self.defer_l2 = False      # True: forward_loss leaves the L2 term to adam_step (train_step / bench)
        self.fused_pyramid = True         # default loss terms: 4 launches for all levels
        self._pyr_cache = None
        self._wplanes_version = None
        if seed is not None and not layout_only:
            self.init_params(seed)

    def stream(self):
        return stream(self.dev)

    # ------------------------------------------------------------------ parameters
    def _alloc_params(self):
        nw = sum(int(torch.Size(l.wshape()).numel()) for l in self.layers)
        nb = sum(l.cout_p for l in self.layers)
        self.n_weights, self.n_params = nw, nw + nb
        z = lambda: torch.zeros(self.n_params, dtype=torch.float32, device=self.dev)
        self.P, self.G, self.M, self.V = z(), z(), z(), z()
        off = 0
        for l in self.layers:
            n = int(torch.Size(l.wshape()).numel())
            l.w = self.P[off:off + n].view(l.wshape())
            l.dw = self.G[off:off + n].view(l.wshape())
            l.mw, l.vw = self.M[off:off + n].view(l.wshape()), self.V[off:off + n].view(l.wshape())   # Adam slots
            off += n
        for l in self.layers:
            l.b = self.P[off:off + l.cout_p]
            l.db = self.G[off:off + l.cout_p]
            l.mb, l.vb = self.M[off:off + l.cout_p], self.V[off:off + l.cout_p]
            off += l.cout_p
        # operand planes of the weights (csrc/conv_planes.hip): direct [P][tap][R][round8(Cc)], transposed
        # [P][tap][Cc][round8(R)] of W[tap][R][Cc]; one flat int16 buffer, refreshed by one batched launch
        self._wp_table = None
        if self.n_planes:
            P = self.n_planes
            users = [l for l in self.layers if l.uses_planes()]
            tot = 0
            for l in users:
                taps, R, Cc = l.wplane_view()
                tot += P * (taps * R * round8(Cc) + taps * Cc * round8(R))
            self.WP = torch.zeros(tot, dtype=torch.int16, device=self.dev)
            off = 0
            for l in users:
                taps, R, Cc = l.wplane_view()
                n = P * taps * R * round8(Cc)
                l.wpl_d = self.WP[off:off + n].view(P, taps, R, round8(Cc))
                off += n
                n = P * taps * Cc * round8(R)
                l.wpl_t = self.WP[off:off + n].view(P, taps, Cc, round8(R))
                off += n
            self._wp_users = users
            self._wp_table = self._wp_table_of(users)
            self._wp_range_tables = {}
        self._adam_tables = {}

    @staticmethod
    def _wp_table_of(users):
        import ctypes
        n = len(users)
        return (n,
                (ctypes.c_void_p * n)(*[l.w.data_ptr() for l in users]),
                (ctypes.c_int * n)(*[l.wplane_view()[0] for l in users]),
                (ctypes.c_int * n)(*[l.wplane_view()[1] for l in users]),
                (ctypes.c_int * n)(*[l.wplane_view()[2] for l in users]),
                (ctypes.c_void_p * n)(*[l.wpl_d.data_ptr() for l in users]),
                (ctypes.c_void_p * n)(*[l.wpl_t.data_ptr() for l in users]))

    def refresh_weight_planes(self, force=False):
        """Re-split the parameters into their operand planes.  Needed whenever P changed: adam_step marks it; in-place
        torch writes to P are seen through the tensor version counter; inside a stream capture it always runs (the
        captured step is replayed after every optimizer update)."""
        if self._wp_table is None:
            return
        capturing = torch.cuda.is_current_stream_capturing()
        if self.planes_external and capturing and not force:
            return        # the step runner re-splits each bucket right after its optimizer update, outside the graphs
        if not (force or capturing or self._wplanes_version != self.P._version):
            return
        n, w, taps, R, Cc, d, t = self._wp_table
        check(_lib.lib().unflow_weight_planes_batched(n, w, taps, R, Cc, d, t, self.n_planes, self.stream()), "weight_planes")
        self._wplanes_version = self.P._version

    def refresh_weight_planes_ranges(self, ranges):
        """Re-split the weight tensors that lie in the flat parameter ranges [(lo, hi), ...] on the current stream (the
        bucketed form of refresh_weight_planes: train.py StepRunner calls it after each bucket's optimizer update)."""
        if self._wp_table is None:
            return
        key = tuple(ranges)
        tab = self._wp_range_tables.get(key)
        if tab is None:
            base = self.P.data_ptr()
            users = [l for l in self._wp_users
                     if any(lo <= (l.w.data_ptr() - base) // 4 < hi for lo, hi in ranges)]
            tab = self._wp_table_of(users) if users else (0,)
            self._wp_range_tables[key] = tab
        if tab[0] == 0:
            return
        n, w, taps, R, Cc, d, t = tab
        check(_lib.lib().unflow_weight_planes_batched(n, w, taps, R, Cc, d, t, self.n_planes, self.stream()), "weight_planes")

    def init_params(self, seed=0):
        """layers.variance_scaling_initializer() (flownet.py:177): truncated normal, stddev sqrt(1.3*2/fan_in);
        zero biases.  Drawn on the host in the reference's variable order so the oracle can share them."""
        from_tf = OrderedDict()
        gen = torch.Generator().manual_seed(seed)
        for l in self.layers:
            shape = l.tf_wshape()
            fan_in = l.k * l.k * (l.cin if l.kind == 'conv' else l.cout)
            std = math.sqrt(1.3 * 2.0 / fan_in)
            t = torch.empty(shape, dtype=torch.float32)
            torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std, generator=gen)
            from_tf[l.name + '/weights'] = t
            from_tf[l.name + '/biases'] = torch.zeros(l.cout)
        self.load_tf_params(from_tf)
        return from_tf

    def load_tf_params(self, tf_params):
        """tf_params: {'<scope>/<layer>/weights': HWIO (conv) or [k,k,out,in] (deconv), '.../biases'} on any device,
        keyed by the reference's variable names (what a TF checkpoint holds)."""
        self.P.zero_()
        for l in self.layers:
            w = tf_params[l.name + '/weights'].to(self.dev, torch.float32)
            assert tuple(w.shape) == l.tf_wshape(), (l.name, tuple(w.shape), l.tf_wshape())
            for plo, tlo, n in l.in_map:
                if l.kind == 'conv':
                    l.w[:, :, plo:plo + n, :l.cout] = w[:, :, tlo:tlo + n, :]
                else:
                    l.w[:, :, :l.cout, plo:plo + n] = w[:, :, :, tlo:tlo + n]
            l.b[:l.cout].copy_(tf_params[l.name + '/biases'].to(self.dev, torch.float32))
        self._wplanes_version = None

    def _export(self, wattr, battr):
        out = OrderedDict()
        for l in self.layers:
            w = getattr(l, wattr)
            parts = []
            for plo, tlo, n in l.in_map:
                parts.append(w[:, :, plo:plo + n, :l.cout] if l.kind == 'conv' else w[:, :, :l.cout, plo:plo + n])
            out[l.name + '/weights'] = torch.cat(parts, 2 if l.kind == 'conv' else 3).detach().cpu().clone()
            out[l.name + '/biases'] = getattr(l, battr)[:l.cout].detach().cpu().clone()
        return out

    def export_tf_params(self):
        return self._export('w', 'b')

    def export_tf_adam_slots(self):
        """Adam's first / second moments under the names tf.train.AdamOptimizer gives its slot variables:
        '<variable>/Adam' and '<variable>/Adam_1' (what the reference's Saver writes next to the weights: slim's
        get_variables_to_restore(include=[scope]) matches them by prefix, train.py:33-38)."""
        out = OrderedDict()
        for suffix, (wa, ba) in (('/Adam', ('mw', 'mb')), ('/Adam_1', ('vw', 'vb'))):
            for k, v in self._export(wa, ba).items():
                out[k + suffix] = v
        return out

    def load_tf_adam_slots(self, tf_slots):
        """Inverse of export_tf_adam_slots; variables without slots in `tf_slots` keep zero moments."""
        self.M.zero_()
        self.V.zero_()
        for suffix, (wa, ba) in (('/Adam', ('mw', 'mb')), ('/Adam_1', ('vw', 'vb'))):
            for l in self.layers:
                kw, kb = l.name + '/weights' + suffix, l.name + '/biases' + suffix
                if kw in tf_slots:
                    w = tf_slots[kw].to(self.dev, torch.float32)
                    dst = getattr(l, wa)
                    for plo, tlo, n in l.in_map:
                        if l.kind == 'conv':
                            dst[:, :, plo:plo + n, :l.cout] = w[:, :, tlo:tlo + n, :]
                        else:
                            dst[:, :, :l.cout, plo:plo + n] = w[:, :, :, tlo:tlo + n]
                if kb in tf_slots:
                    getattr(l, ba)[:l.cout].copy_(tf_slots[kb].to(self.dev, torch.float32))

    def export_tf_grads(self):
        """Gradients of the data loss (the L2-regulariser gradient 0.0004*w is fused into adam_step)."""
        return self._export('dw', 'db')

    # ------------------------------------------------------------------ buffers
    def _alloc_activations(self):
        N, H, W, dev = self.N, self.H, self.W, self.dev
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        # mean-subtracted network input (4th channel zero) + its operand planes.  Row length 4 when W is even: a 16-byte
        # K granule of conv1 is then two pixels (rgb4_form of csrc/conv_planes.hip), else 8 (one zero-padded pixel)
        self.X0 = L.PT(z(N, H, W, 4), torch.zeros(self.n_planes, N, H, W, 4 if (W % 2 == 0 and RGB4_FORM) else 8, dtype=torch.int16, device=dev)
                       if self.n_planes else None)
        self.x0 = self.X0.t
        self.im01 = z(N, H, W, 3)     # images in [0,1] for the losses
        for st in self.stages:
            st.alloc()
        last = self.stages[-1]
        # views used by the loss code, tests and tools: the trained (last) network
        self.act = ActView(last)
        self.act.update(last.act)
        self.act.update(x0=self.x0, im01=self.im01)
        self.grad = last.grad
        # loss-side pyramid: one level per flow output of the last network (unsupervised.py:85-104)
        self.layer_weights = LAYER_WEIGHTS_FULL_RES if self.full_res else LAYER_WEIGHTS
        self.patch_distances = LAYER_PATCH_DISTANCES_FULL_RES if self.full_res else LAYER_PATCH_DISTANCES
        self.final_flow_scale = FLOW_SCALE * 4 if self.full_res else FLOW_SCALE
        self.lv = []
        for i, lvl in enumerate(last.flow_levels):
            d = 2 ** lvl
            h, w = H // d, W // d
            im = self.im01 if d == 1 else z(N, h, w, 3)
            self.lv.append(dict(h=h, w=w, im=im, gray1=z(N, h, w), gray2w=z(N, h, w), dist=z(N, h, w), dgray=z(N, h, w),
                                flow=last.act['flow%d' % lvl], gflow=last.grad['flow%d' % lvl],
                                fs=self.final_flow_scale / (2 ** i), lw=self.layer_weights[i], pd=self.patch_distances[i]))
        self.loss_acc = z(1)
        self.final_flow = z(N, H, W, 2)
        self.mean_host = (_lib.ctypes.c_float * 3)(*CHANNEL_MEAN)
        self.epe_out = z(2)

    def _build_masks(self):
        """create_border_mask(im, 0.1) (losses.py:338-344) then downsample 4, 2, 2, 2, 2 (unsupervised.py:101,147); with
        full_res the mask itself is level 0 (unsupervised.py:94).  Every level owns ONE [B,h,w] buffer for the whole life
        of the engine (so a captured hipGraph never holds a stale mask pointer): without augmentation it holds B copies of
        the static mask, with augmentation the per-sample warped masks."""
        from .. import ops
        H, W, B = self.H, self.W, self.B
        sz = int(math.ceil(min(H, W) * 0.1))
        m = torch.zeros(1, H, W, 1, device=self.dev)
        m[:, sz:H - sz, sz:W - sz] = 1.0
        self.border0, self._aug, self._mask_aug = m, None, False
        cur = m if bool(self.params.get('border_mask')) else torch.ones(1, H, W, 1, device=self.dev)
        for i, lv in enumerate(self.lv):
            if not (i == 0 and self.full_res):
                cur = ops.downsample(cur, 4 if (i == 0) else 2)
            lv['mask_static'] = cur.reshape(1, lv['h'], lv['w']).contiguous()
            lv['mask'] = lv['mask_static'].expand(B, -1, -1).contiguous()
            lv['n_mask'] = B

    def _image_and_mask_pyramid_scales(self):
        """Downsampling factor from the previous level for every loss level."""
        return [1 if (i == 0 and self.full_res) else (4 if i == 0 else 2) for i in range(len(self.lv))]

    # ------------------------------------------------------------------ forward
    def set_input(self, im1, im2, augment=None):
        """im1, im2: [B,H,W,3] float32 in [0,255] (what the reference's input queue delivers).
        augment: None (augment=False) or the draws of core.augment.draw_training_augmentation — then the step sees
        what unsupervised.py:37-68 builds: geometrically augmented images for the losses, photometrically augmented
        mean-free ones for the network, and a per-sample border mask (product of the two warped masks)."""
        B, N, H, W = self.B, self.N, self.H, self.W
        lib = _lib.lib()
        st = self.stream()
        # the kernels below take raw device pointers: bring whatever the input pipeline delivers (numpy batches of
        # core/input.py, host tensors, strided views) to contiguous fp32 on this engine's device — a no-op for a conforming tensor
        im1 = torch.as_tensor(im1).to(device=self.dev, dtype=torch.float32).contiguous()
        im2 = torch.as_tensor(im2).to(device=self.dev, dtype=torch.float32).contiguous()
        if tuple(im1.shape) != (B, H, W, 3) or tuple(im2.shape) != (B, H, W, 3):
            raise ValueError("set_input: expected two [%d,%d,%d,3] batches, got %s and %s"
                             % (B, H, W, tuple(im1.shape), tuple(im2.shape)))
        if augment is None:
            # one launch: both frames -> mean-free network input (+ its operand planes for conv1 of a FlowNetC) and the [0,1]
            # images of the losses
            pl = self.X0.pl if (self.X0.pl is not None and self.stages[0].is_c) else None
            check(lib.unflow_prepare_image_pair(ptr(im1), ptr(im2), cl(B * H * W), ptr(self.x0), ptr(self.im01),
                                                self.mean_host, _lib.planes_of(pl), st), "prepare_image_pair")
            if self._mask_aug:
                for lv in self.lv:
                    for b in range(B):
                        check(lib.unflow_copy(ptr(lv['mask'][b]), ptr(lv['mask_static']), _lib.csz(lv['h'] * lv['w'] * 4), st), "copy")
                self._mask_aug = False
            return
        from . import augment as A
        if self._aug is None:
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.dev)
            self._aug = dict(tmp=z(N, H, W, 3), mg=z(B, H, W, 1), ml=z(B, H, W, 1))
        a = self._aug
        check(lib.unflow_prepare_image_pair(ptr(im1), ptr(im2), cl(B * H * W), ptr(self.x0), ptr(a['tmp']), self.mean_host,
                                            None, st), "prepare_image_pair")
        tg, tl = augment['theta_global'], augment['theta_local']
        A.transformer(a['tmp'], tg, out=self.im01, n_samples=N)            # im1_geo, first pass of im2 (:40-44)
        A.transformer(self.im01[B:], tl, out=a['tmp'][B:], n_samples=B)    # im2 locally (:47-50)
        check(lib.unflow_copy(ptr(self.im01[B:]), ptr(a['tmp'][B:]), _lib.csz(B * H * W * 3 * 4), st), "copy")
        if self.params.get('border_mask'):
            A.transformer(self.border0, tg, out=a['mg'], n_samples=B)
            A.transformer(self.border0, tl, out=a['ml'], n_samples=B)
            check(lib.unflow_mul_inplace(ptr(a['mg']), ptr(a['ml']), cl(B * H * W), st), "mul")    # border_mask_local * global (:51)
            from .. import ops
            cur = a['mg']
            for lv, sc in zip(self.lv, self._image_and_mask_pyramid_scales()):
                if sc != 1:
                    cur = ops.downsample(cur, sc)
                check(lib.unflow_copy(ptr(lv['mask']), ptr(cur), _lib.csz(B * lv['h'] * lv['w'] * 4), st), "copy")
            self._mask_aug = True
        A.photometric(self.im01, augment, out=self.x0, mean=CHANNEL_MEAN)  # im*_photo - channel_mean (:53-57,67-68)
        self._input_planes()

    def _input_planes(self):
        """Operand planes of the network input for conv1 of a FlowNetC (FlowNetS stages build theirs after stack_input)."""
        if self.X0.pl is not None and self.stages[0].is_c:
            L.planes_from_f32(self.x0, self.X0.pl, C=4)

    def forward_net(self):
        """flownet(im1, im2, spec, backward_flow=True) (flownet.py:14-81): every stage in order; a refinement stage
        consumes the previous stage's finest flow."""
        self.refresh_weight_planes()
        prev = None
        with torch.cuda.device(self.dev):
            for st in self.stages:
                st.forward(prev)
                prev = st.act['flow2']

    def forward_loss(self, with_grad=True):
        """compute_losses + the pyramid assembly (losses.py:16-87, unsupervised.py:85-150) over the directed batch;
        with_grad also leaves d(loss)/d(flowN) in self.grad['flowN'].  Terms enter iff their `<name>_weight` is set
        (unsupervised.py:136-141), exactly the pruning TF does."""
        lib = _lib.lib()
        st = self.stream()
        N, B = self.N, self.B
        P = self.params
        wt = lambda k: float(P.get(k + '_weight') or 0.0)
        check(lib.unflow_zero(ptr(self.loss_acc), _lib.csz(4), st), "zero")
        occl = {'': 0, None: 0, 'fb': 1, 'disocc': 2}[P.get('mask_occlusion', '')]
        use_border = bool(P.get('border_mask'))
        levels = self.lv if P.get('pyramid_loss') else self.lv[:1]
        # image pyramid: downsample(im, 4) then successive downsample(., 2) (unsupervised.py:99-100,145-146); with
        # full_res level 0 is the image itself (unsupervised.py:92-93)
        scales = self._image_and_mask_pyramid_scales()
        if len(levels) == 5 and scales[:5] == [4, 2, 2, 2, 2] and len(self.lv) == 5:
            # the default pyramid in one launch (each level bit-identical to the chained downsample calls)
            import ctypes
            ptrs = (ctypes.c_void_p * 5)(*[lv['im'].data_ptr() for lv in self.lv])
            check(lib.unflow_image_pyramid5(ptr(self.act['im01']), ptrs, N, self.H, self.W, st), "image_pyramid5")
        else:
            prev_im, ph, pw = self.act['im01'], self.H, self.W
            for lv, sc in list(zip(self.lv, scales))[:len(levels)]:
                if sc != 1:
                    check(lib.unflow_downsample_fwd(ptr(prev_im), ptr(lv['im']), N, ph, pw, 3, sc, st), "downsample")
                prev_im, ph, pw = lv['im'], lv['h'], lv['w']
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
            fs, lw = lv['fs'], lv['lw']
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
                check(lib.unflow_zero(ptr(gflow), _lib.csz(gflow.numel() * 4), st), "zero")
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
                    check(lib.unflow_scale(ptr(flow), cf(fs), ptr(lv['fscaled']), cl(flow.numel()), st), "scale")
                    fwm = lv['fwmap']
                    ws = lv['fwws']      # this engine's own scratch (see _level_extra), never the shared grow-only one
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
                    check(lib.unflow_zero(ptr(lv['dimtmp']), _lib.csz(lv['dimtmp'].numel() * 4), st), "zero")
                    check(lib.unflow_image_warp_bwd(ptr(lv['gwarped']), ptr(flow), 2, ptr(flow), cf(fs), ptr(lv['dimtmp']),
                                                    ptr(gflow), 1, B, N, h, w, 2, st), "image_warp_bwd(flow)")
                    check(lib.unflow_add_inplace(ptr(gflow), ptr(lv['dimtmp']), cl(gflow.numel()), st), "add")
            # ---- data terms
            if wt('ternary'):
                D = lv['pd']
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
                check(lib.unflow_zero(ptr(lv['gflow']), _lib.csz(lv['gflow'].numel() * 4), st), "zero")
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
                lw = lv['lw']
                a = arr[i]
                a.im, a.flow, a.gray1, a.gray2w = lv['im'].data_ptr(), lv['flow'].data_ptr(), lv['gray1'].data_ptr(), \
                    lv['gray2w'].data_ptr()
                a.mask, a.dist, a.d_flow = lv['mask'].data_ptr(), lv['dist'].data_ptr(), lv['gflow'].data_ptr()
                a.H, a.W, a.n_mask, a.max_distance = lv['h'], lv['w'], lv['n_mask'], lv['pd']
                a.flow_scale = lv['fs']
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
        # forward_warp's scratch is owned by the engine (sized here, before any capture): the module-global grow-only
        # workspace of ops.py is shared by every engine of the process — a second, larger engine (Trainer.eval's 1 x 384 x
        # 1280 one beside a small-crop training engine) would re-allocate it under this engine's captured graphs
        lib = _lib.lib()
        lib.unflow_forward_warp_workspace_bytes.restype = _lib.ctypes.c_size_t
        lv['fwws'] = torch.empty(int(lib.unflow_forward_warp_workspace_bytes(N, h, w, 1)) // 4 + 64, dtype=torch.float32, device=self.dev)

    # ------------------------------------------------------------------ backward
    def backward_net(self, part=None):
        """Gradients of the trained (last) network; earlier stages are behind stop_gradient (flownet.py:51-54).
        part 0 / 1: the two halves used to overlap the data-parallel all-reduce (see grad_buckets)."""
        last_part = len(self.stages[-1].part_bounds) - 2
        with torch.cuda.device(self.dev):
            # the batched bias gradients (column sums of every dz: HBM-bound) go out on the main stream once the last data
            # gradient is queued, beside the tail of the filter gradients on the second stream
            final = part in (None, last_part)
            multi = self.train_all and len(self.stages) > 1
            self.stages[-1].backward(part, self._bias_grads if final and not multi else None)
            if final and multi:
                for i in range(len(self.stages) - 1, 0, -1):
                    self._stack_backward(self.stages[i], self.stages[i - 1])
                    self.stages[i - 1].backward(None, self._bias_grads if i == 1 else None)

    def _stack_backward(self, st, prev):
        """d loss / d (flow2 of the previous network) through the stage input of `st` (train_all).  The previous
        network's coarser flows do not reach the loss directly (only flows[-1] enters it, unsupervised.py:82-83)."""
        for lvl in prev.flow_levels:
            gz = prev.grad['flow%d' % lvl]
            check(_lib.lib().unflow_zero(ptr(gz), _lib.csz(gz.numel() * 4), self.stream()), "zero")
        g2 = prev.grad['flow2']
        pf = prev.act['flow2']
        dx = st.grad['x0s']
        check(_lib.lib().unflow_stack_input_bwd(ptr(dx), dx.stride(2), ptr(self.x0), ptr(pf), ptr(g2), self.B, self.N,
                                                self.H, self.W, pf.shape[1], pf.shape[2], cf(4 * FLOW_SCALE),
                                                self.stream()), "stack_input_bwd")

    def set_backward_parts(self, last_layers=('conv4',)):
        """Cut the trained network's backward pass after the named layers (see _Stage.set_parts); returns the part count."""
        self.stages[-1].set_parts(tuple(last_layers))
        return len(self.stages[-1].part_bounds) - 1

    def part_buckets(self):
        """One list of flat ranges of self.G per backward part: the gradients that are FINAL once backward_net(part) has
        run, so that their all-reduce (and their Adam update) may start while the later parts still compute.  The last
        part also carries everything else: the biases (their column sums are the last launch) and the parameters of
        frozen / earlier stages (train_all: earlier networks accumulate until the very end, so one bucket)."""
        st = self.stages[-1]
        nparts = len(st.part_bounds) - 1
        if self.train_all:
            return [[] for _ in range(nparts - 1)] + [[(0, self.n_params)]]
        out, covered = [], []
        for k in range(nparts):
            lo, hi = st.part_weight_range(k)
            out.append([(lo, hi)])
            covered.append((lo, hi))
        covered += self.frozen_ranges()
        covered.sort()
        rest, pos = [], 0
        for lo, hi in covered:
            if lo > pos:
                rest.append((pos, lo))
            pos = max(pos, hi)
        if pos < self.n_params:
            rest.append((pos, self.n_params))
        out[-1] = out[-1] + rest
        return out

    def frozen_ranges(self):
        """Flat weight ranges of the networks behind stop_gradient (flownet.py:51-54): their data gradient is identically
        zero on every rank — nothing to exchange (train.py:388-422 skips None gradients) — but the optimizer still
        applies the L2 term to them."""
        if self.train_all:
            return []
        out = []
        for stg in self.stages[:-1]:
            lo = min(l.dw.data_ptr() for l in stg.layers) - self.G.data_ptr()
            hi = max(l.dw.data_ptr() + l.dw.numel() * 4 for l in stg.layers) - self.G.data_ptr()
            out.append((lo // 4, hi // 4))
        return out

    def grad_buckets(self):
        """(early, late) flat ranges for the default two-part cut (kept for callers of the two-part API)."""
        pb = self.part_buckets()
        return [r for part in pb[:-1] for r in part], pb[-1]

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
            check(lib.unflow_colsum_batched(n, xs, lds, npx, cs, outs, ptr(ws), _lib.csz(ws.numel() * 4), self.stream()),
                  "colsum_batched")

    # ------------------------------------------------------------------ optimiser
    def adam_begin(self, lr, beta1=0.9, beta2=0.999):
        """Start optimizer step t: returns the TF-form step size lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) for
        adam_range() calls (bucketed update: each flat range as soon as its reduced gradient has landed)."""
        self.step_count += 1
        t = self.step_count
        self._wplanes_version = None
        return lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)

    def adam_range(self, lo, hi, lr_t, grad_scale=1.0, beta1=0.9, beta2=0.999, eps=1e-8):
        """The fused L2 + Adam update of flat range [lo, hi) on the CURRENT stream (weights first in the flat layout: the
        regularised prefix of the range is what lies below n_weights)."""
        if hi <= lo:
            return
        nreg = max(0, min(hi, self.n_weights) - lo)
        sl = lambda t: ptr(t[lo:hi])                                        # noqa: E731
        if self.defer_l2:
            check(_lib.lib().unflow_adam_step_regloss(sl(self.P), sl(self.G), sl(self.M), sl(self.V), cl(hi - lo), cl(nreg),
                                                      cf(grad_scale), cf(L2_SCALE), cf(lr_t), cf(beta1), cf(beta2), cf(eps),
                                                      ptr(self.loss_acc), self.stream()), "adam")
        else:
            check(_lib.lib().unflow_adam_step(sl(self.P), sl(self.G), sl(self.M), sl(self.V), cl(hi - lo), cl(nreg),
                                              cf(grad_scale), cf(L2_SCALE), cf(lr_t), cf(beta1), cf(beta2), cf(eps),
                                              self.stream()), "adam")

    def _adam_table(self, ranges):
        """ctypes table of unflow_adam_planes_batched for the flat ranges [(lo, hi), ...]: every weight tensor that lies in a
        range (with its plane copies, or without for the layers that have none) + the bias part of each range as one
        unregularised pseudo-tensor.  Ranges are unions of whole tensors by construction (part_buckets)."""
        import ctypes
        key = tuple(ranges)
        tab = self._adam_tables.get(key)
        if tab is not None:
            return tab
        base = self.P.data_ptr()
        rows = []
        for l in self.layers:
            off = (l.w.data_ptr() - base) // 4
            n = l.w.numel()
            hit = [r for r in ranges if r[0] <= off < r[1]]
            if not hit:
                continue
            assert off + n <= hit[0][1], "flat range cuts through %s" % l.name
            if l.uses_planes() and self.n_planes:
                taps, R, Cc = l.wplane_view()
                rows.append((l.w.data_ptr(), taps, R, Cc, l.wpl_d.data_ptr(), l.wpl_t.data_ptr(), 1))
            else:
                sh = l.wshape()
                rows.append((l.w.data_ptr(), sh[0] * sh[1], sh[2], sh[3], 0, 0, 1))
        for lo, hi in ranges:
            blo = max(lo, self.n_weights)
            if hi > blo:
                rows.append((base + 4 * blo, 1, 1, hi - blo, 0, 0, 0))
        n = len(rows)
        col = lambda i, ct: (ct * n)(*[r[i] for r in rows])                 # noqa: E731
        tab = (n, col(0, ctypes.c_void_p), col(1, ctypes.c_int), col(2, ctypes.c_int), col(3, ctypes.c_int), col(4, ctypes.c_void_p),
               col(5, ctypes.c_void_p), col(6, ctypes.c_int))
        self._adam_tables[key] = tab
        return tab

    def adam_ranges_fused(self, ranges, lr_t, grad_scale=1.0, beta1=0.9, beta2=0.999, eps=1e-8):
        """The fused L2 + Adam update of the flat ranges AND the re-split of their weights into the operand planes, one launch
        (per 64 tensors) on the current stream; afterwards the planes of these ranges are current."""
        ranges = [(lo, hi) for lo, hi in ranges if hi > lo]
        if not ranges:
            return
        n, w, taps, R, Cc, d, t, reg = self._adam_table(ranges)
        if n == 0:
            return
        check(_lib.lib().unflow_adam_planes_batched(n, w, taps, R, Cc, d, t, reg, self.n_planes, ptr(self.P), ptr(self.G), ptr(self.M),
                                                    ptr(self.V), cf(grad_scale), cf(L2_SCALE), cf(lr_t), cf(beta1), cf(beta2), cf(eps),
                                                    ptr(self.loss_acc if self.defer_l2 else None), self.stream()), "adam_planes")

    def adam_step(self, lr, grad_scale=1.0, beta1=0.9, beta2=0.999, eps=1e-8):
        """tf.train.AdamOptimizer(beta1=0.9, beta2=0.999) update (train.py:151-152), TF formulation, with the
        slim.l2_regularizer(0.0004) gradient added for the weight tensors (biases are not regularised)."""
        self.step_count += 1
        t = self.step_count
        lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
        self._wplanes_version = None       # the parameters change behind torch's back: their operand planes are stale
        if self.fused_adam and self.dev.type == 'cuda':
            self.adam_ranges_fused([(0, self.n_params)], lr_t, grad_scale, beta1, beta2, eps)
            self._wplanes_version = self.P._version        # ... and were re-split by the same launch
            return
        if self.defer_l2:   # the loss of THIS step gets its regularisation term from the pre-update parameters here
            check(_lib.lib().unflow_adam_step_regloss(ptr(self.P), ptr(self.G), ptr(self.M), ptr(self.V),
                                                      cl(self.n_params), cl(self.n_weights), cf(grad_scale), cf(L2_SCALE),
                                                      cf(lr_t), cf(beta1), cf(beta2), cf(eps), ptr(self.loss_acc),
                                                      self.stream()), "adam")
            return
        check(_lib.lib().unflow_adam_step(ptr(self.P), ptr(self.G), ptr(self.M), ptr(self.V), cl(self.n_params),
                                          cl(self.n_weights), cf(grad_scale), cf(L2_SCALE), cf(lr_t), cf(beta1),
                                          cf(beta2), cf(eps), self.stream()), "adam")

    def check_device_faults(self, world_sync=False):
        """Sticky device-side fault counters of the library, read (and cleared) with a host sync: the bounded spins of the
        stream-K fix-up (csrc/conv_streamk.hip: a waiter that gave up has added a slab that was never written, so an
        output tile of that launch is WRONG).  Raises instead of returning a count: callers sit at points where the host
        synchronises anyway (Trainer's display interval and checkpoint, the end of bench.py's timed region).  world_sync: in a
        multi-rank run, agree on the outcome first (one small all-reduce) so that all ranks raise or none does."""
        import contextlib
        with (torch.cuda.device(self.dev) if self.dev.type == 'cuda' else contextlib.nullcontext()):
            n = _lib.lib().unflow_debug_streamk_timeouts()      # the counter is a per-device symbol: read THIS engine's device
        if world_sync:
            # every rank raises together (a lone raise in front of a collective leaves the other ranks hanging in it): the
            # MAX over ranks of (count, read failure)
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                t = torch.tensor([max(n, 0), 1 if n < 0 else 0], dtype=torch.int64,
                                 device=self.dev if dist.get_backend() == 'nccl' else 'cpu')
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                n = -1 if int(t[1]) else int(t[0])
        if n < 0:
            raise RuntimeError("stream-K conv kernels: the device-side fault counter could not be read (hipMemcpyFromSymbol "
                               "failed on %s, or on another rank) — the state of the run is unknown" % (self.dev,))
        if n != 0:
            raise RuntimeError("stream-K conv kernels: %d fix-up wait(s) timed out since the last check%s — results of those "
                               "launches are wrong; set UNFLOW_OPT_STREAMK=0 and report"
                               % (n, " (max over ranks)" if world_sync else ""))
        return 0

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
        """final_flow_fw / _bw: resize_bilinear(flow2, im_shape) * 5 * 4 (unsupervised.py:103-104), or flow0 * 20 with
        full_res (unsupervised.py:95-97)."""
        if self.full_res:
            f0 = self.act['flow0']
            check(_lib.lib().unflow_scale(ptr(f0), cf(FLOW_SCALE * 4), ptr(self.final_flow), cl(f0.numel()), self.stream()), "scale")
            return self.final_flow[:self.B], self.final_flow[self.B:]
        f2 = self.act['flow2']
        N, h, w, _ = f2.shape
        check(_lib.lib().unflow_resize_bilinear_tf1(ptr(f2), ptr(self.final_flow), N, h, w, 2, self.H, self.W,
                                                    cf(FLOW_SCALE * 4), self.stream()), "resize_bilinear")
        return self.final_flow[:self.B], self.final_flow[self.B:]

    def flows(self):
        """(flows_fw, flows_bw): lists [flow2..flow6] ([flow0, flow1, flow2..] with full_res), NHWC, like
        flownet(..., backward_flow=True)[-1]."""
        B = self.B
        lv = self.stages[-1].flow_levels
        fw = [self.act['flow%d' % l][:B] for l in lv]
        bw = [self.act['flow%d' % l][B:] for l in lv]
        return fw, bw


def flow_error_avg(flow_1, flow_2, mask=None):
    """core/flow_util.py:98-103 (EPE)."""
    f1, f2 = flow_1.contiguous(), flow_2.contiguous()
    out = torch.empty(2, dtype=torch.float32, device=f1.device)
    npix = f1.numel() // 2
    check(_lib.lib().unflow_flow_error_sums(ptr(f1), ptr(f2), ptr(None if mask is None else mask.contiguous()),
                                            ptr(out), cl(npix), stream(f1.device)), "flow_error")
    return out[0] / out[1]


FlowNetCEngine = FlowNetEngine   # the FlowNetC training step is the default spec ('C')
