"""HIP path vs the oracle AT THE BENCHMARKED SHAPES (BASELINE.json configs[1] and configs[3]).

The step runs through the same launch path bench.py times (split-K plans and tile configs of the real shapes, fused
4-launch loss pyramid, two captured hipGraphs) and is compared with the fp64 shadow of the oracle
(oracle/model_ref.py; torch-CPU autograd) on the same seeded inputs:

  * FlowNetC 384x512, B = 1 and B = 4 (the benchmark batch, bench.py's weights and images): loss (rel 1e-4), final flows
    (EPE 1e-3 px, the north-star bar), EVERY parameter gradient against the untouched fp64 oracle (2e-2 of its max, 3e-3
    mean-relative) and against the fp64 oracle differentiated along the engine's leaky-ReLU branches (2e-4 / 2e-4), with the
    number of units on the other side of the kink bounded (<= 1e-5 of the units);
  * the KITTI loss variant (fb 0.2, occ 12.4, mask_occlusion 'fb') at B = 4, 384x512 with mixed occlusion masks;
  * FlowNetCSS 768x1024, B = 1: end-to-end flows and loss vs the fp32 oracle, and the trained (last) network's flows,
    loss and gradients vs the fp64 oracle fed the engine's own stage-2 flow (isolates the stage from the amplification
    of fp32 noise through the two frozen stages in front of it);
  * the 441-channel correlation at the step's shape (8 directed samples of 256x48x64, written into the concat slice).

reference: src/e2eflow/core/unsupervised.py:27-164, core/flownet.py:14-81, ops/correlation_op.cu.cc:51-248."""
import os

import numpy as np
import pytest
import torch

from parity_util import BranchAligned, check_grads, flownet_c_order, flownet_s_order, graph_step, images, oracle_step

pytestmark = pytest.mark.gpu


MAX_FLIP_FRACTION = 1e-5        # leaky-ReLU units allowed on the other side of the kink in fp64 (measured: ~3e-7 of the units)


def _bench_or_seeded_inputs(eng, B, H, W):
    """B = 4: the weights and images bench.py uses (seed 0 / generator 1234); otherwise seeded shifted-frame pairs."""
    if B == 4:
        tf_params = eng.init_params(seed=0)
        g = torch.Generator().manual_seed(1234)
        return tf_params, torch.rand(B, H, W, 3, generator=g) * 255, torch.rand(B, H, W, 3, generator=g) * 255
    return (eng.init_params(seed=100 + B),) + images(B, H, W, 200 + B)


@pytest.mark.parametrize("B", [1, 4])
def test_flownetc_step_384x512_vs_fp64_oracle(B, dev):
    """B = 1 and the BENCHMARK batch B = 4: loss, final flows, and every parameter gradient against BOTH the untouched fp64
    oracle (loose: set by the leaky-ReLU units within fp32 noise of the kink) and the fp64 oracle differentiated along the
    engine's branches (tight) — with the number of such units BOUNDED, so that a wrong derivative branch on more than
    1e-5 of the units (planes_shared.h leaky_grad_from_bits takes the sign from the bf16 hi plane) fails here."""
    from unflow_amd.core.engine import FlowNetCEngine, flow_error_avg
    H, W = 384, 512
    eng = FlowNetCEngine(B, H, W, device=dev, seed=None)
    tf_params, im1, im2 = _bench_or_seeded_inputs(eng, B, H, W)
    loss = graph_step(eng, im1.to(dev), im2.to(dev))
    fw, bw = eng.final_flows()
    got = eng.export_tf_grads()
    loss_ref, ffw, fbw, grads = oracle_step(tf_params, im1, im2, dtype=torch.float64)
    assert abs(loss - loss_ref) <= 1e-4 * abs(loss_ref), (loss, loss_ref)
    for a, r in ((fw, ffw), (bw, fbw)):
        r = r.float().to(dev)
        assert (a - r).abs().max().item() < 1e-3
        assert flow_error_avg(a, r).item() < 1e-3           # north-star bar: EPE within 1e-3 px of the reference path
    # (a) the oracle as is (loose): set by the handful of leaky-ReLU units whose pre-activation lies within
    # fp32 noise of the kink (parity_util.BranchAligned) — measured 6 of 1.8e7 units, worst element 6e-3, mean 1e-3
    # (b) the oracle differentiated along the branch the engine took at every leaky-ReLU: everything else, tight
    with BranchAligned(eng.act, flownet_c_order(B)) as al:
        _, _, _, grads_al = oracle_step(tf_params, im1, im2, dtype=torch.float64)
    print("B=%d: %d of %d leaky-ReLU units on the other side of the kink in fp64" % (B, al.flips, al.units))
    assert al.flips <= MAX_FLIP_FRACTION * al.units, (al.flips, al.units)
    # (tensors with <= 64 elements — the 2-element flow-head biases, the 2 -> 2 upsamplers — are sums of a few hundred terms
    # that cancel heavily: measured 2.1e-4 on flow6/biases at B = 4; bound 2e-3 there)
    worst = check_grads(got, grads_al, tf_params, max_tol=2e-4, mean_tol=2e-4, small_tol=2e-3 if B == 4 else None,
                        label="B=%d branch-aligned fp64 oracle:" % B)
    plain = check_grads(got, grads, tf_params, max_tol=2e-2, mean_tol=3e-3, label="B=%d plain fp64 oracle:" % B)
    print("B=%d 384x512: loss rel %.2e; %d of %d leaky-ReLU units on the other side of the kink in fp64; gradients: plain oracle "
          "worst max-rel %.2e mean-rel %.2e, branch-aligned worst max-rel %.2e mean-rel %.2e"
          % (B, abs(loss - loss_ref) / abs(loss_ref), al.flips, al.units, plain[0], plain[1], worst[0], worst[1]))


def test_flownetc_b4_384x512_kitti_loss_variant_vs_fp64_oracle(dev):
    """SURVEY 8(d) config 2 at the benchmark shape: the KITTI training loss — forward-backward consistency + occlusion
    penalty with the occlusion mask thresholded from the flows (losses.py:43-56, config.ini:172-174: fb_weight 0.2,
    occ_weight 12.4, mask_occlusion 'fb') — B = 4, 384x512, with flow heads rescaled so that every pyramid level sees
    flows of ~0.3-0.7 px and a MIXED occlusion mask (bench.kitti_variant; the mask fractions are asserted).  Loss and final
    flows vs the fp64 oracle; parameter gradients vs the branch-aligned fp64 oracle on all but a bounded fraction of elements (a
    pixel whose |fw + bw_warped|^2 lies within fp32 noise of the threshold carries a different mask bit in fp32 and fp64)."""
    import bench
    from unflow_amd.core.engine import FlowNetCEngine, flow_error_avg
    from oracle import model_ref as M
    B, H, W = 4, 384, 512
    params = bench.kitti_variant_params()
    eng = FlowNetCEngine(B, H, W, params=params, device=dev, seed=None)
    tf_params = bench.kitti_variant_weights(eng.init_params(seed=0))
    eng.load_tf_params(tf_params)
    im1, im2 = images(B, H, W, 77)
    loss = graph_step(eng, im1.to(dev), im2.to(dev))
    fw, bw = eng.final_flows()
    got = eng.export_tf_grads()
    # the masks must be non-trivial at every level, or this test says nothing about them
    lfw, lbw = eng.flows()
    for i, (a, b) in enumerate(zip(lfw, lbw)):
        s = M.FLOW_SCALE / 2 ** i
        a, b = a.cpu().double() * s, b.cpu().double() * s
        bww = M.image_warp(b, a)
        occ = (M.length_sq(a + bww) > 0.01 * (M.length_sq(a) + M.length_sq(bww)) + 0.5).double().mean().item()
        print("level %d: mean |flow| %.3f px, fb-occluded fraction %.3f" % (i, a.abs().mean().item(), occ))
        assert 0.03 < occ < 0.97, (i, occ)
    with BranchAligned(eng.act, flownet_c_order(B)) as al:
        loss_ref, ffw, fbw, grads = oracle_step(tf_params, im1, im2, params, dtype=torch.float64)
    assert al.flips <= MAX_FLIP_FRACTION * al.units, (al.flips, al.units)
    assert abs(loss - loss_ref) <= 2e-4 * abs(loss_ref), (loss, loss_ref)
    assert flow_error_avg(fw, ffw.float().to(dev)).item() < 1e-3
    assert flow_error_avg(bw, fbw.float().to(dev)).item() < 1e-3
    # mask-bit flips are isolated pixels: per tensor, all but 0.2 % of the elements within 5e-4 of the tensor's max, and the
    # worst element within 2e-2 (the plain-oracle bound of the default loss)
    worst = 0.0
    for k, gr in grads.items():
        ref = gr.double() - (0.0004 * tf_params[k].double() if k.endswith('/weights') else 0.0)
        d = (got[k].double().cpu() - ref).abs() / (ref.abs().max() + 1e-30)
        worst = max(worst, d.max().item())
        assert d.max().item() < 2e-2, (k, d.max().item())
        if ref.numel() >= 1024:
            assert (d > 5e-4).double().mean().item() < 2e-3, (k, (d > 5e-4).double().mean().item())
    print("B=4 384x512 KITTI loss variant: loss %.4f oracle %.4f rel %.2e; worst gradient element %.2e of its tensor's max; "
          "%d of %d leaky units flipped" % (loss, loss_ref, abs(loss - loss_ref) / abs(loss_ref), worst, al.flips, al.units))


def test_flownet_css_768x1024_b2_end_to_end_vs_fp64_fixture(dev):
    """BASELINE configs[3] at its BENCHMARKED batch (B = 2, bench.py measure_secondary): C -> S -> S at 768x1024 end to end against
    the fp64 oracle's outputs, committed as tests/golden/css_768x1024_b2_fp64.npz by tests/golden/make_css_fp64_fixture.py (an
    fp64 pass over three networks at this size is minutes of host time: run once, cached; the weights and images are
    regenerated here from their seeds).  The bound is the north star's own, with the noise floor of ANY fp32 evaluation of this
    graph beside it: |HIP - fp64| <= max(1e-3 px, 4 x |oracle32 - fp64|), the latter measured by the same script (4e-5 px)."""
    import os
    import numpy as np
    from unflow_amd.core.engine import FlowNetEngine
    sys_path_golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    import importlib.util
    spec_ = importlib.util.spec_from_file_location('make_css_fp64_fixture', os.path.join(sys_path_golden, 'make_css_fp64_fixture.py'))
    mk = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mk)
    fx = np.load(os.path.join(sys_path_golden, 'css_768x1024_b2_fp64.npz'))
    B, H, W = mk.B, mk.H, mk.W
    assert list(fx['meta']) == [B, H, W, mk.WSEED, mk.ISEED, mk.STEP] and float(fx['head_scale']) == mk.HEAD_SCALE
    params = dict(flownet=mk.SPEC, pyramid_loss=True, border_mask=True, ternary_weight=1.0, smooth_2nd_weight=3.0)
    eng = FlowNetEngine(B, H, W, params=params, device=dev, seed=None)
    eng.load_tf_params(mk.css_params())
    im1, im2 = images(B, H, W, mk.ISEED)
    loss = graph_step(eng, im1.to(dev), im2.to(dev))          # the bench's launch path: captured hipGraphs
    fw, bw = eng.final_flows()
    st = mk.STEP

    def epe(a, ref):
        d = a[:, ::st, ::st].cpu().double() - torch.from_numpy(ref).double()
        return (d * d).sum(-1).sqrt().mean().item()
    e_loss = abs(loss - float(fx['loss64'])) / abs(float(fx['loss64']))
    e_fw, e_bw = epe(fw, fx['fw64']), epe(bw, fx['bw64'])
    floor = max(float(fx['epe32_lat_fw']), float(fx['epe32_lat_bw']))
    print("CSS 768x1024 B=2 end to end vs the fp64 oracle (lattice [::%d, ::%d]): loss rel %.2e (fp32 oracle: %.2e), EPE fw %.2e bw %.2e px "
          "(fp32 oracle against fp64: %.2e / %.2e px), max |flow| %.0f px"
          % (st, st, e_loss, abs(float(fx['loss32']) - float(fx['loss64'])) / float(fx['loss64']), e_fw, e_bw,
             float(fx['epe32_lat_fw']), float(fx['epe32_lat_bw']), float(fx['fmax'])))
    assert e_loss <= 1e-4, e_loss
    assert e_fw <= max(1e-3, 4 * floor) and e_bw <= max(1e-3, 4 * floor), (e_fw, e_bw, floor)
    assert e_fw < 1e-3 and e_bw < 1e-3, (e_fw, e_bw)      # the north star's plain bound holds at these flow magnitudes


def test_flownet_css_768x1024_vs_oracle(dev):
    """BASELINE configs[3]: C -> S -> S at 768x1024; every gradient of the trained network (B = 1 keeps the CPU oracle's fp64
    backward pass within minutes; the benchmarked batch B = 2 is covered end to end by the fixture test above)."""
    from unflow_amd.core.engine import FlowNetEngine, flow_error_avg, FLOW_SCALE
    from oracle import model_ref as M
    B, H, W = 1, 768, 1024
    spec = 'CSS'
    params = dict(flownet=spec, pyramid_loss=True, border_mask=True, ternary_weight=1.0, smooth_2nd_weight=3.0)
    eng = FlowNetEngine(B, H, W, params=params, device=dev, seed=None)
    tf_params = M.init_params_spec(spec, seed=31)
    # Random-initialised stacks blow the flow up to ~700 px by the third network; warp sample points then sit within fp32
    # noise (3e-5 px at that magnitude) of pixel boundaries, where the derivative of bilinear interpolation jumps, for a
    # dozen pixels per pass, and one such pixel moves the 12x16-position gradients of the deep layers by ~1e-3 (measured:
    # 4e-3 with the unscaled weights, with every leaky-ReLU branch aligned).  A trained stack refines by a few pixels:
    # scale the flow heads so that the flows stay in that regime (tens of pixels).
    for k in tf_params:
        if k.split('/')[-2].startswith('flow') and k.endswith('/weights'):      # flowN and flowN_upM layers
            tf_params[k] = tf_params[k] * 0.3
    eng.load_tf_params(tf_params)
    im1, im2 = images(B, H, W, 32)
    loss = graph_step(eng, im1.to(dev), im2.to(dev))
    fw, bw = eng.final_flows()
    got = eng.export_tf_grads()
    reg = 0.0004 * 0.5 * sum((v.double() ** 2).sum().item() for k, v in tf_params.items() if k.endswith('/weights'))

    # (1) end to end, fp32 oracle: at these flow magnitudes (tens of pixels) the plain 1e-3 px bound
    loss32, ffw, fbw, _ = oracle_step(tf_params, im1, im2, params, dtype=torch.float32, backward=False)
    e_loss = abs(loss - loss32) / abs(loss32)
    e_fw = flow_error_avg(fw, ffw.to(dev)).item()
    e_bw = flow_error_avg(bw, fbw.to(dev)).item()
    fmax = max(ffw.abs().max().item(), fbw.abs().max().item())
    print("CSS 768x1024 end-to-end vs fp32 oracle: loss rel %.2e, EPE fw %.2e bw %.2e px (max |flow| %.0f px)"
          % (e_loss, e_fw, e_bw, fmax))
    assert e_loss <= 2e-4 and e_fw < 1e-3 and e_bw < 1e-3

    # (2) the trained network alone, fp64, fed the ENGINE's stage-2 flows: loss, flows and every gradient, tight bounds
    scope = 'stack_2_flownet/'
    P64 = {k: v.clone().double().requires_grad_() for k, v in tf_params.items() if k.startswith(scope)}
    prev = eng.stages[1].act['flow2'].detach().cpu().double()
    mean = torch.tensor(M.CHANNEL_MEAN, dtype=torch.float64) / 255.0
    a, b = im1.double() / 255.0 - mean, im2.double() / 255.0 - mean

    def stage(x, y, flow):      # flownet.py:46-57 with stop_gradient
        flow = M.resize_bilinear_tf1(flow, H, W) * 4 * M.FLOW_SCALE
        warp = M.image_warp(y, flow)
        inputs = torch.cat([x, y, flow, warp, torch.abs(warp - x)], 3)
        return M.flownet_s(P64, inputs, pre=scope + 'flownet_s/')
    flows_fw, flows_bw = stage(a, b, prev[:B]), stage(b, a, prev[B:])
    comb, _ = M.pyramid_loss_from_flows(im1.double(), im2.double(), flows_fw, flows_bw, params)
    comb.backward()
    assert abs((loss - reg) - comb.item()) <= 1e-4 * abs(comb.item()), (loss - reg, comb.item())
    f2 = eng.stages[2].act['flow2']
    ref2 = torch.cat([flows_fw[0], flows_bw[0]], 0).detach()
    # flow2 in network units; the final flow is 20x this (resize is a convex combination).  Random-weight stacks
    # produce flows of hundreds of pixels, so the bound is 1e-3 px + 1e-5 of the flow magnitude (fp32 carries 6e-8)
    err_px = (f2.cpu().double() - ref2).abs().max().item() * FLOW_SCALE * 4
    assert err_px < 1e-3 + 1e-5 * ref2.abs().max().item() * FLOW_SCALE * 4, (err_px, ref2.abs().max().item() * 20)
    grads = {k: v.grad for k, v in P64.items()}
    for v in P64.values():
        v.grad = None
    with BranchAligned(eng.stages[2].act, flownet_s_order(B)) as al:
        fa, fb = stage(a, b, prev[:B]), stage(b, a, prev[B:])
        comb2, _ = M.pyramid_loss_from_flows(im1.double(), im2.double(), fa, fb, params)
        comb2.backward()
    print("CSS last stage: %d of %d leaky-ReLU units on the other side of the kink in fp64" % (al.flips, al.units))
    # mean-relative bound 2e-3: these gradients are heavy-tailed (max|g| >> mean|g|) while fp32 noise is uniform
    worst = check_grads(got, {k: v.grad for k, v in P64.items()}, tf_params, max_tol=2e-4, mean_tol=2e-3, small_tol=5e-3,
                        label="CSS last stage, branch-aligned fp64 oracle:")
    plain = check_grads(got, grads, tf_params, max_tol=2e-2, mean_tol=3e-3, label="CSS last stage, plain fp64 oracle:")
    print("CSS last stage vs fp64 oracle: loss rel %.2e, flow2 error %.2e px (max |flow| %.0f px); %d of %d leaky units flipped; "
          "gradients plain max-rel %.2e mean-rel %.2e, branch-aligned max-rel %.2e mean-rel %.2e"
          % (abs((loss - reg) - comb.item()) / abs(comb.item()), err_px, ref2.abs().max().item() * 20, al.flips, al.units,
             plain[0], plain[1], worst[0], worst[1]))
    for k, v in got.items():      # frozen stages: no data gradient at all
        if not k.startswith(scope):
            assert v.abs().max().item() == 0.0, k


def test_correlation_step_shape_b4_vs_oracle(dev, oracle_lib):
    """The correlation exactly as the B = 4 step issues it (engine.py forward / _backward_shallow): 8 directed samples of
    256 x 48 x 64 from ONE shared feature tensor, sample n paired with (n + 4) % 8, output written into channels 32..473
    of the 476-wide concat buffer; backward fused over both roles of every sample."""
    from unflow_amd import _lib
    from unflow_amd._lib import check, ptr, stream
    B, N, C, h, w = 4, 8, 256, 48, 64
    rs = np.random.RandomState(77)
    feat = rs.randn(N, h, w, C).astype(np.float32)
    f = torch.from_numpy(feat).to(dev)
    cat = torch.zeros(N, h, w, 476, device=dev)
    lib = _lib.lib()
    check(lib.unflow_correlation_nhwc_fwd(ptr(f), ptr(f), C, B, ptr(cat[..., 32:473]), 476, N, C, h, w, 1, 20, 20, 1, 2,
                                          stream()), "correlation")
    a_nchw = np.ascontiguousarray(feat.transpose(0, 3, 1, 2))
    b_nchw = np.ascontiguousarray(np.roll(a_nchw, -B, axis=0))        # partner of sample n is (n + B) % N
    attrs = dict(pad=20, kernel_size=1, max_displacement=20, stride_1=1, stride_2=2)
    ref = oracle_lib.correlation(a_nchw, b_nchw, **attrs)             # [N,441,h,w]
    got = cat[..., 32:473].permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    assert cat[..., :32].abs().max().item() == 0 and cat[..., 473:].abs().max().item() == 0   # neighbours untouched
    go = rs.randn(*ref.shape).astype(np.float32)
    gcat = torch.zeros(N, h, w, 476, device=dev)
    gcat[..., 32:473] = torch.from_numpy(go).to(dev).permute(0, 2, 3, 1)
    gfeat = torch.full((N, h, w, C), 7.0, device=dev)                  # must be overwritten, not accumulated into
    check(lib.unflow_correlation_nhwc_bwd(ptr(gcat[..., 32:473]), 476, ptr(f), ptr(f), C, B, ptr(gfeat), ptr(None), C, 1,
                                          N, C, h, w, 1, 20, 20, 1, 2, stream()), "correlation_grad")
    g0, g1 = oracle_lib.correlation_grad(go, a_nchw, b_nchw, **attrs)
    want = g0 + np.roll(g1, B, axis=0)                                 # g1[n] is the gradient of sample (n + B) % N
    gotg = gfeat.permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(gotg - want).max() <= 2e-4 * np.abs(want).max()


def _corr_bwd_pl_vs_oracle(dev, oracle_lib, B, C, h, w, md, s2, ld_cat, lo, rot=None):
    """unflow_correlation_nhwc_bwd_pl WITH planes — the entry point and kernel the step runs (engine.py backward: corr_bwd_pl_kernel,
    feature operand by LDS-DMA from the bf16 planes) — against the scalar C oracle (correlation_op.cu.cc:119-248): dOut read
    from channels [lo, lo + oc) of an ld_cat-wide buffer, gradient buffer pre-filled (must be overwritten), both roles of every
    sample fused with the step's pairing n <-> (n + B) % N."""
    import ctypes
    from unflow_amd import _lib
    from unflow_amd._lib import check, ptr, stream
    from unflow_amd.core import layers as L
    N = 2 * B
    rs = np.random.RandomState(1000 + C + h + md)
    feat = rs.randn(N, h, w, C).astype(np.float32)
    F = L.PT.alloc((N, h, w, C), dev, 3)
    F.t.copy_(torch.from_numpy(feat))
    L.planes_from_f32(F.t, F.pl)
    o3 = (ctypes.c_int * 3)()
    assert _lib.lib().unflow_correlation_out_shape(h, w, 1, md, md, 1, s2, o3) == 0
    oc, oh, ow = tuple(o3)
    a_nchw = np.ascontiguousarray(feat.transpose(0, 3, 1, 2))
    b_nchw = np.ascontiguousarray(np.roll(a_nchw, -B, axis=0))
    attrs = dict(pad=md, kernel_size=1, max_displacement=md, stride_1=1, stride_2=s2)
    go = rs.randn(N, oc, oh, ow).astype(np.float32)
    gcat = torch.zeros(N, oh, ow, ld_cat, device=dev)
    gcat[..., lo:lo + oc] = torch.from_numpy(go).to(dev).permute(0, 2, 3, 1)
    g0, g1 = oracle_lib.correlation_grad(go, a_nchw, b_nchw, **attrs)
    want = g0 + np.roll(g1, B, axis=0)
    saved = _lib.get_option("corr_bwd_rot")
    try:
        if rot is not None:
            _lib.set_option("corr_bwd_rot", rot)
        gfeat = torch.full((N, h, w, C), 7.0, device=dev)
        pl = _lib.planes_of(F.pl)
        check(_lib.lib().unflow_correlation_nhwc_bwd_pl(ptr(gcat[..., lo:lo + oc]), ld_cat, ptr(F.t), ptr(F.t), C, pl, pl, B,
                                                        ptr(gfeat), ptr(None), C, 1, N, C, h, w, 1, md, md, 1, s2, stream()),
              "correlation_bwd_pl")
    finally:
        _lib.set_option("corr_bwd_rot", saved)
    gotg = gfeat.permute(0, 3, 1, 2).cpu().numpy()
    err = np.abs(gotg - want).max() / np.abs(want).max()
    assert err <= 2e-4, err
    return err


def test_correlation_bwd_planes_step_shape_vs_oracle(dev, oracle_lib):
    """The step's correlation backward at the step's shape: N = 8 directed samples of 256 x 48 x 64, dOut in channels 32..473 of
    the 476-wide concat gradient."""
    _corr_bwd_pl_vs_oracle(dev, oracle_lib, 4, 256, 48, 64, 20, 2, 476, 32)


@pytest.mark.parametrize("rot", [0, 1])
def test_correlation_bwd_planes_step_shape_row_orders(rot, dev, oracle_lib):
    """Both displacement-row orders of corr_bwd_pl_kernel (option corr_bwd_rot) at half the step's batch."""
    _corr_bwd_pl_vs_oracle(dev, oracle_lib, 2, 256, 48, 64, 20, 2, 476, 32, rot=rot)


def test_correlation_bwd_planes_c96_vs_oracle(dev, oracle_lib):
    """The 3/8-width nets' correlation (C = 96: flownet.py:22-23) through the same entry point."""
    _corr_bwd_pl_vs_oracle(dev, oracle_lib, 2, 96, 24, 32, 20, 2, 476, 12)


@pytest.mark.parametrize("rot", [0, 1])
@pytest.mark.parametrize("shape", [(1, 64, 20, 131, 4, 1), (2, 128, 12, 100, 6, 1)])
def test_correlation_bwd_planes_narrow_band_vs_oracle(shape, rot, dev, oracle_lib):
    """Narrow-band tiling (the north star's +-4 cost volume: 81 channels; and r = 6) in both row orders."""
    B, C, h, w, md, s2 = shape
    oc = (2 * (md // s2) + 1) ** 2
    _corr_bwd_pl_vs_oracle(dev, oracle_lib, B, C, h, w, md, s2, oc + 3, 3, rot=rot)


@pytest.mark.parametrize("mode", ["bf16x3", "fp32"])
def test_conv_wide_dynamic_range_elementwise(mode, dev):
    """Accuracy class of the fp32-equivalent 3-way bf16 split, in pytest: a 3x3 conv whose inputs
    span 4 decades per channel group, K = 9*256 = 2304, checked ELEMENT-WISE against fp64 — every output against
    sum|x||w| of its own receptive field, the natural scale of a dot product's rounding error — in both math modes
    (the mode is a per-process library knob, so each runs in a sub-process)."""
    import subprocess
    import sys
    env = dict(os.environ, UNFLOW_CONV_MATH=mode, UNFLOW_DYNRANGE_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.abspath(__file__),
                        "-k", "dynrange_child"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_dynrange_child(dev):
    if os.environ.get("UNFLOW_DYNRANGE_CHILD") != "1":
        pytest.skip("runs as a sub-process of test_conv_wide_dynamic_range_elementwise")
    from unflow_amd.core import layers as L
    B, H, W, Cin, Cout, k = 2, 48, 64, 256, 128, 3      # 6144 sites: the 128x128 bf16x3 tiles of the real layers
    g = torch.Generator().manual_seed(5)
    scale = (10.0 ** (torch.arange(Cin) % 5 - 2).float()).view(1, 1, 1, Cin)        # 1e-2 .. 1e2 across channels
    x = torch.randn(B, H, W, Cin, generator=g) * scale
    wgt = torch.randn(k, k, Cin, Cout, generator=g) / scale.view(1, 1, Cin, 1) * (10.0 ** (torch.arange(Cout) % 3).float())
    xd, wd = x.to(dev), wgt.to(dev).contiguous()
    y = torch.zeros(B, H, W, Cout, device=dev)
    L.conv2d_fwd(xd, wd, None, y, 1, False)
    x64, w64 = x.double().permute(0, 3, 1, 2), wgt.double().permute(3, 2, 0, 1)
    ref = torch.nn.functional.conv2d(x64, w64, padding=1).permute(0, 2, 3, 1)
    mag = torch.nn.functional.conv2d(x64.abs(), w64.abs(), padding=1).permute(0, 2, 3, 1)   # sum |x||w| per output
    err = ((y.cpu().double() - ref).abs() / mag).max().item()
    # fp32 accumulation over K = 2304: ~K^0.5 * 2^-24 typical, a few e-7 worst case (the fp32 MFMA measures the same)
    assert err < 1e-6, err
    # data gradient and filter gradient through the same check
    dz = torch.randn(B, H, W, Cout, generator=g) * (10.0 ** (torch.arange(Cout) % 4 - 2).float())
    dzd = dz.to(dev)
    dx = torch.zeros(B, H, W, Cin, device=dev)
    L.conv2d_bwd_data(dzd, wd, dx, 1)
    dz64 = dz.double().permute(0, 3, 1, 2)
    rdx = torch.nn.functional.conv_transpose2d(dz64, w64, padding=1).permute(0, 2, 3, 1)
    mdx = torch.nn.functional.conv_transpose2d(dz64.abs(), w64.abs(), padding=1).permute(0, 2, 3, 1)
    e2 = ((dx.cpu().double() - rdx).abs() / mdx).max().item()
    assert e2 < 1e-6, e2
    dw = torch.zeros(k, k, Cin, Cout, device=dev)
    L.conv2d_bwd_filter(xd, dzd, dw, None, 1)
    xp = torch.nn.functional.pad(x64, (1, 1, 1, 1))
    rdw = torch.zeros(k, k, Cin, Cout, dtype=torch.float64)
    mdw = torch.zeros_like(rdw)
    for ky in range(k):
        for kx in range(k):
            patch = xp[:, :, ky:ky + H, kx:kx + W]
            rdw[ky, kx] = torch.einsum('bchw,bohw->co', patch, dz64)
            mdw[ky, kx] = torch.einsum('bchw,bohw->co', patch.abs(), dz64.abs())
    e3 = ((dw.cpu().double() - rdw).abs() / mdw).max().item()
    assert e3 < 1e-6, e3
    print("mode %s: element-wise error / sum|a||b|: fwd %.2e dgrad %.2e wgrad %.2e"
          % (os.environ.get("UNFLOW_CONV_MATH"), err, e2, e3))
