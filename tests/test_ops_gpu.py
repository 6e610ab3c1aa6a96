"""-m gpu: the four custom ops + image_warp through the C ABI vs the CPU oracle (oracle/ops_ref.c)
and the reference's own known-answer vectors (tests/golden/ref_kats.json).

Tolerances: integer outputs (warp indices, splat footprints) bit-exact; fp32 values rtol/atol 1e-5
(summation order differs from the 32-lane order of the CUDA kernel); gradients 1e-4."""
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def t(a, dev):
    return torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)


def close(a, b, tol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol)


# ---------------------------------------------------------------- reference KATs through the HIP ops
def test_kat_correlation(kats, dev):
    from unflow_amd import ops
    for name in ("correlation_trivial", "correlation_batch"):
        k = kats[name]
        out = ops.correlation(t(k["first"], dev), t(k["second"], dev), **k["attrs"])
        close(out, np.array(k["expected"], np.float32), 1e-6)


def test_kat_warps(kats, dev):
    from unflow_amd import ops
    from unflow_amd.core.image_warp import image_warp
    for name in ("warp_move", "warp_interpolate", "backward_warp_batches"):
        k = kats[name]
        im = t(np.array(k["image"])[..., None], dev)
        fl = t(k["flow"], dev)
        close(ops.backward_warp(im, fl)[..., 0], np.array(k["expected"], np.float32), 1e-6)
    for name in ("warp_move", "warp_interpolate", "image_warp_batches"):
        k = kats[name]
        im = t(np.array(k["image"])[..., None], dev)
        fl = t(k["flow"], dev)
        close(image_warp(im, fl)[..., 0], np.array(k["expected"], np.float32), 1e-6)


def test_kat_downsample_forward_warp(kats, dev):
    from unflow_amd import ops
    k = kats["downsample"]
    close(ops.downsample(t(np.array(k["image"])[..., None], dev), k["scale"])[..., 0], np.array(k["expected"]), 0)
    z = torch.zeros(1, 20, 20, 2, device=dev)
    for det in (True, False):
        v = ops.forward_warp(z, deterministic=det)[0, 10, 10, 0].item()
        assert abs(v - kats["forward_warp_zero_flow_interior"]["expected"]) < 1e-5


def test_error_statuses(dev):
    from unflow_amd import ops, _lib
    a = torch.zeros(1, 4, 8, 8, device=dev)
    with pytest.raises(_lib.UnflowError, match="kernel_size must be odd"):
        ops.correlation(a, a, kernel_size=2)
    with pytest.raises(_lib.UnflowError, match="Input shapes have to be the same"):
        ops.correlation(a, torch.zeros(1, 4, 8, 9, device=dev))
    with pytest.raises(_lib.UnflowError, match="Invalid correlation settings"):
        ops.correlation(a, a, max_displacement=20, pad=0)
    with pytest.raises(_lib.UnflowError, match="divisible by scale"):
        ops.downsample(torch.zeros(1, 6, 8, 3, device=dev), 4)
    with pytest.raises(TypeError):
        ops.downsample(torch.zeros(1, 8, 8, 3), 2)  # CPU tensor: no CPU path exists


# ---------------------------------------------------------------- correlation vs oracle
CORR_CASES = [
    # B, C, H, W, attrs
    (2, 16, 12, 14, dict(kernel_size=1, max_displacement=4, pad=4, stride_1=1, stride_2=2)),
    (1, 8, 9, 11, dict(kernel_size=3, max_displacement=2, pad=3, stride_1=1, stride_2=1)),
    (2, 5, 10, 9, dict(kernel_size=3, max_displacement=4, pad=4, stride_1=2, stride_2=2)),
    (1, 32, 16, 24, dict(kernel_size=1, max_displacement=6, pad=6, stride_1=1, stride_2=1)),
    (2, 64, 8, 16, dict(kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2)),
    (1, 12, 7, 9, dict(kernel_size=1, max_displacement=3, pad=5, stride_1=1, stride_2=1)),
]


@pytest.mark.parametrize("case", CORR_CASES)
def test_correlation_vs_oracle(case, dev, oracle_lib):
    from unflow_amd import ops
    B, C, H, W, attrs = case
    rs = np.random.RandomState(zlib.crc32(str(case).encode()))
    a = rs.randn(B, C, H, W).astype(np.float32)
    b = rs.randn(B, C, H, W).astype(np.float32)
    ref = oracle_lib.correlation(a, b, **attrs)
    ta, tb = t(a, dev).requires_grad_(), t(b, dev).requires_grad_()
    out = ops.correlation(ta, tb, **attrs)
    assert tuple(out.shape) == ref.shape
    close(out, ref, 1e-5)
    go = rs.randn(*ref.shape).astype(np.float32)
    out.backward(t(go, dev))
    g0, g1 = oracle_lib.correlation_grad(go, a, b, **attrs)
    close(ta.grad, g0, 1e-4)
    close(tb.grad, g1, 1e-4)


def test_correlation_flownetc_shape_full(dev, oracle_lib):
    """FlowNetC configuration (flownet.py:221-222) at the real feature-map size, one sample."""
    from unflow_amd import ops
    rs = np.random.RandomState(7)
    a = rs.randn(1, 256, 48, 64).astype(np.float32)
    b = rs.randn(1, 256, 48, 64).astype(np.float32)
    attrs = dict(pad=20, kernel_size=1, max_displacement=20, stride_1=1, stride_2=2)
    ref = oracle_lib.correlation(a, b, **attrs)
    ta, tb = t(a, dev).requires_grad_(), t(b, dev).requires_grad_()
    out = ops.correlation(ta, tb, **attrs)
    assert tuple(out.shape) == (1, 441, 48, 64)
    close(out, ref, 2e-5)
    go = rs.randn(*ref.shape).astype(np.float32)
    out.backward(t(go, dev))
    g0, g1 = oracle_lib.correlation_grad(go, a, b, **attrs)
    close(ta.grad, g0, 2e-4)
    close(tb.grad, g1, 2e-4)


def test_correlation_reference_boundary_with_and_without_plane_room(dev, oracle_lib):
    """unflow_correlation_fwd / _bwd (the reference op's NCHW boundary): with the workspace unflow_correlation_workspace_bytes
    asks for, operand planes are built in it and the matrix-core kernels run; with the fp32 part alone the fp32 kernels do —
    both against the oracle, and one byte less than the fp32 part is refused."""
    import ctypes
    from unflow_amd import _lib
    from unflow_amd._lib import check, ptr, stream
    L = _lib.lib()
    B, C, H, W = 2, 64, 10, 37
    attrs = dict(kernel_size=1, max_displacement=4, pad=4, stride_1=1, stride_2=1)
    args = (1, 4, 4, 1, 1)
    rs = np.random.RandomState(11)
    a = rs.randn(B, C, H, W).astype(np.float32)
    b = rs.randn(B, C, H, W).astype(np.float32)
    ref = oracle_lib.correlation(a, b, **attrs)
    go = rs.randn(*ref.shape).astype(np.float32)
    g0r, g1r = oracle_lib.correlation_grad(go, a, b, **attrs)
    ta, tb, tg = t(a, dev), t(b, dev), t(go, dev)
    full = L.unflow_correlation_workspace_bytes(B, C, H, W, *args)
    fp32_part = (4 * a.size + ref.size) * 4
    assert full > fp32_part
    ws = torch.zeros(full // 4 + 64, device=dev)
    for nbytes in (full, fp32_part):
        out = torch.full(ref.shape, float('nan'), device=dev)
        check(L.unflow_correlation_fwd(ptr(ta), ptr(tb), ptr(out), B, C, H, W, *args, ptr(ws), ctypes.c_size_t(nbytes), stream()))
        close(out, ref, 1e-5)
        g0, g1 = torch.full(a.shape, float('nan'), device=dev), torch.full(a.shape, float('nan'), device=dev)
        check(L.unflow_correlation_bwd(ptr(tg), ptr(ta), ptr(tb), ptr(g0), ptr(g1), B, C, H, W, *args, ptr(ws), ctypes.c_size_t(nbytes),
                                       stream()))
        close(g0, g0r, 1e-4)
        close(g1, g1r, 1e-4)
    assert L.unflow_correlation_bwd(ptr(tg), ptr(ta), ptr(tb), ptr(g0), ptr(g1), B, C, H, W, *args, ptr(ws),
                                    ctypes.c_size_t(fp32_part - 1), stream()) == -9


# ---------------------------------------------------------------- warps vs oracle
def _flows(rs, B, H, W, kind):
    if kind == "normal":
        return (rs.randn(B, H, W, 2) * 4).astype(np.float32)
    if kind == "zero":
        return np.zeros((B, H, W, 2), np.float32)
    if kind == "oob":
        return (rs.choice([-50.0, 50.0], size=(B, H, W, 2)) + rs.randn(B, H, W, 2)).astype(np.float32)
    if kind == "tiny":   # x+u crosses an integer only in fp32: op and image_warp index formulas differ here
        return (rs.choice([-1e-8, 1e-8, -1.0 + 1e-8, 0.99999994], size=(B, H, W, 2))).astype(np.float32)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["normal", "zero", "oob", "tiny"])
@pytest.mark.parametrize("C", [1, 3])
def test_backward_warp_vs_oracle(kind, C, dev, oracle_lib):
    from unflow_amd import ops
    rs = np.random.RandomState(11)
    B, H, W = 2, 17, 23
    im = rs.rand(B, H, W, C).astype(np.float32)
    fl = _flows(rs, B, H, W, kind)
    tim, tfl = t(im, dev), t(fl, dev).requires_grad_()
    out = ops.backward_warp(tim, tfl)
    close(out, oracle_lib.backward_warp(im, fl), 1e-6)
    assert np.array_equal(ops.backward_warp_indices(tfl.detach()).cpu().numpy(), oracle_lib.backward_warp_indices(fl))
    go = rs.randn(B, H, W, C).astype(np.float32)
    out.backward(t(go, dev))
    close(tfl.grad, oracle_lib.backward_warp_grad(go, im, fl), 1e-5)


@pytest.mark.parametrize("kind", ["normal", "zero", "oob", "tiny"])
@pytest.mark.parametrize("C", [2, 3])
def test_image_warp_vs_oracle(kind, C, dev, oracle_lib):
    from unflow_amd.core.image_warp import image_warp, image_warp_indices
    rs = np.random.RandomState(13)
    B, H, W = 2, 17, 23
    im = rs.rand(B, H, W, C).astype(np.float32)
    fl = _flows(rs, B, H, W, kind)
    tim, tfl = t(im, dev).requires_grad_(), t(fl, dev).requires_grad_()
    out = image_warp(tim, tfl)
    ref, idx = oracle_lib.image_warp(im, fl, return_indices=True)
    close(out, ref, 1e-6)
    assert np.array_equal(image_warp_indices(tim.detach(), tfl.detach()).cpu().numpy(), idx)   # bit-exact gather indices
    go = rs.randn(B, H, W, C).astype(np.float32)
    out.backward(t(go, dev))
    d_im, d_fl = oracle_lib.image_warp_grad(go, im, fl)
    close(tim.grad, d_im, 1e-5)
    close(tfl.grad, d_fl, 1e-5)


@pytest.mark.parametrize("kind", ["normal", "zero", "oob"])
def test_forward_warp_vs_oracle(kind, dev, oracle_lib):
    from unflow_amd import ops
    rs = np.random.RandomState(17)
    B, H, W = 2, 19, 21
    fl = _flows(rs, B, H, W, kind)
    tfl = t(fl, dev).requires_grad_()
    ref = oracle_lib.forward_warp(fl)
    out = ops.forward_warp(tfl, deterministic=True)
    close(out, ref, 1e-5)
    out2 = ops.forward_warp(tfl.detach(), deterministic=True)
    assert torch.equal(out.detach(), out2)           # deterministic mode is bit-reproducible
    close(ops.forward_warp(tfl.detach(), deterministic=False), ref, 1e-5)
    assert np.array_equal(ops.forward_warp_ranges(tfl.detach()).cpu().numpy(), oracle_lib.forward_warp_ranges(fl))
    go = rs.randn(B, H, W, 1).astype(np.float32)
    out.backward(t(go, dev))
    close(tfl.grad, oracle_lib.forward_warp_grad(go, fl), 1e-5)


@pytest.mark.parametrize("field", ["torn", "mixed"])
def test_forward_warp_far_sources_binned_by_target_tile(field, dev, oracle_lib):
    """Flows of tens of pixels tear a source tile apart: most footprints leave the tile's LDS window.  With the full workspace
    (unflow_forward_warp_workspace_bytes) such sources are binned by 32 x 32 target tile and gathered without global atomics;
    with the minimum workspace their taps are global atomics.  Both must give the oracle's sums; in deterministic mode the two
    paths must agree BIT FOR BIT (the same fixed-point terms, integer sums) and be stable run to run."""
    import ctypes
    from unflow_amd import _lib, ops
    from unflow_amd._lib import check, ptr, stream
    rs = np.random.RandomState(29)
    B, H, W = 3, 100, 170                                   # ragged against the 64 x 16 source tiles and the 32 x 32 target tiles
    if field == "torn":
        fl = (rs.rand(B, H, W, 2) * 100 - 50).astype(np.float32)
    else:                                                   # a coherent field with a fifth of the pixels flung far away
        fl = (rs.randn(B, H, W, 2) * 1.5 + np.array([6.0, -3.0])).astype(np.float32)
        far = rs.rand(B, H, W) < 0.2
        fl[far] += (rs.rand(int(far.sum()), 2) * 160 - 80).astype(np.float32)
    ref = oracle_lib.forward_warp(fl)
    tfl = t(fl, dev)
    lib = _lib.lib()
    npx = B * H * W
    full = lib.unflow_forward_warp_workspace_bytes(B, H, W, 1)
    assert full > 8 * npx
    outs = {}
    for det in (1, 0):
        for name, nbytes in (("binned", full), ("atomics", 8 * npx)):
            ws = torch.empty(nbytes // 4 + 16, dtype=torch.float32, device=dev)
            out = torch.full((B, H, W, 1), 7.0, device=dev)
            for rep in range(3):
                check(lib.unflow_forward_warp_fwd(ptr(tfl), ptr(out), B, H, W, det, ptr(ws), _lib.csz(nbytes), stream()), "forward_warp")
                close(out, ref, 1e-5)
                if det:
                    assert torch.equal(out, outs.setdefault((det, name), out.clone())), (name, rep)      # bit-reproducible
    assert torch.equal(outs[(1, "binned")], outs[(1, "atomics")])
    # the autograd op takes the binned path
    assert torch.equal(ops.forward_warp(tfl, deterministic=True), outs[(1, "binned")])


@pytest.mark.parametrize("scale", [2, 4])
def test_downsample_vs_oracle(scale, dev, oracle_lib):
    from unflow_amd import ops
    rs = np.random.RandomState(19)
    im = rs.rand(3, 16, 24, 3).astype(np.float32)
    out = ops.downsample(t(im, dev), scale)
    assert np.array_equal(out.cpu().numpy(), oracle_lib.downsample(im, scale))   # same order of adds -> bit-exact


def test_image_pyramid5_is_the_chain_of_downsamples_bit_for_bit(dev, oracle_lib):
    """unflow_image_pyramid5 (the loss pyramid's images in one launch) == downsample(., 4) then four downsample(., 2)
    (unsupervised.py:99-100,145-146), every level bit-identical to the chained op and to the C oracle."""
    import ctypes
    from unflow_amd import _lib, ops
    from unflow_amd._lib import check, ptr, stream
    rs = np.random.RandomState(23)
    for N, H, W in ((2, 64, 128), (3, 192, 64), (8, 384, 512)):
        im = rs.rand(N, H, W, 3).astype(np.float32)
        tim = t(im, dev)
        outs = [torch.full((N, H >> (2 + k), W >> (2 + k), 3), 7.0, device=dev) for k in range(5)]
        ptrs = (ctypes.c_void_p * 5)(*[o.data_ptr() for o in outs])
        check(_lib.lib().unflow_image_pyramid5(ptr(tim), ptrs, N, H, W, stream()), "image_pyramid5")
        cur, ref = tim, im
        for k in range(5):
            cur = ops.downsample(cur, 4 if k == 0 else 2)
            ref = oracle_lib.downsample(ref, 4 if k == 0 else 2)
            assert torch.equal(outs[k], cur), (N, H, W, k)
            assert np.array_equal(outs[k].cpu().numpy(), ref), (N, H, W, k)
    bad = torch.zeros(1, 96, 128, 3, device=dev)
    outs = [torch.zeros(1, 96 >> (2 + k), 128 >> (2 + k), 3, device=dev) for k in range(4)] + [torch.zeros(1, 2, 2, 3, device=dev)]
    ptrs = (ctypes.c_void_p * 5)(*[o.data_ptr() for o in outs])
    assert _lib.lib().unflow_image_pyramid5(ptr(bad), ptrs, 1, 96, 128, stream()) == -4        # H not a multiple of 64


def test_warp_linearity_full_size(dev):
    """Size-independent property at the benchmark size: warps are linear in the image."""
    from unflow_amd.core.image_warp import image_warp
    g = torch.Generator(device="cpu").manual_seed(3)
    a = torch.rand(4, 384, 512, 3, generator=g).to(dev)
    b = torch.rand(4, 384, 512, 3, generator=g).to(dev)
    fl = (torch.randn(4, 384, 512, 2, generator=g) * 4).to(dev)
    lhs = image_warp(2.0 * a + b, fl)
    rhs = 2.0 * image_warp(a, fl) + image_warp(b, fl)
    assert (lhs - rhs).abs().max().item() < 1e-5
    ident = image_warp(a, torch.zeros_like(fl))
    assert torch.equal(ident, a)     # zero flow is the identity, bit-exact


def test_empty_batch_is_a_no_op(dev):
    """Empty inputs (B = 0): outputs of the right shape, nothing launched, no error."""
    from unflow_amd import ops
    im = torch.zeros(0, 8, 12, 3, device=dev)
    fl = torch.zeros(0, 8, 12, 2, device=dev)
    assert tuple(ops.backward_warp(im, fl).shape) == (0, 8, 12, 3)
    assert tuple(ops.forward_warp(fl).shape) == (0, 8, 12, 1)
    assert tuple(ops.downsample(im, 2).shape) == (0, 4, 6, 3)
    from unflow_amd import _lib
    from unflow_amd._lib import ptr, stream
    lib = _lib.lib()
    one = torch.zeros(4, device=dev)
    # the C ABI itself: zero-sized problems return UNFLOW_OK before touching memory
    assert lib.unflow_backward_warp_fwd(ptr(one), ptr(one), ptr(one), 0, 8, 12, 3, stream()) == 0
    assert lib.unflow_downsample_fwd(ptr(one), ptr(one), 0, 8, 12, 3, 2, stream()) == 0
    assert lib.unflow_adam_step(ptr(one), ptr(one), ptr(one), ptr(one), _lib.cl(0), _lib.cl(0), _lib.cf(1.0),
                                _lib.cf(0.0), _lib.cf(1e-3), _lib.cf(0.9), _lib.cf(0.999), _lib.cf(1e-8), stream()) == 0


def test_correlation_backward_both_math_modes(dev):
    """The correlation backward defaults to fp32-equivalent products on the bf16 matrix cores (corr_bwd_b3_kernel: 3-way bf16
    split in registers, six terms, fp32 accumulation); UNFLOW_CORR_MATH=fp32 keeps v_mfma_f32_32x32x2_f32.  The knob is read
    once per process, so the other mode runs the correlation tests of this file in a sub-process, at the same tolerances."""
    import os
    import subprocess
    import sys
    if os.environ.get("UNFLOW_CORR_MATH_SUBTEST"):
        pytest.skip("already inside the sub-process")
    other = "bf16x3" if os.environ.get("UNFLOW_CORR_MATH", "bf16x3") == "fp32" else "fp32"
    env = dict(os.environ, UNFLOW_CORR_MATH=other, UNFLOW_CORR_MATH_SUBTEST="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-k", "correlation"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]


def _area_weights(n_in, n_out):
    """[n_out, n_in] weights of tf.image.resize_area along one axis (TF's ResizeAreaOp): output i averages the source interval
    [i*s, (i+1)*s), s = n_in / n_out, every source pixel weighted by the covered fraction — the independent restatement."""
    import math
    s = n_in / n_out
    w = torch.zeros(n_out, n_in, dtype=torch.float64)
    for i in range(n_out):
        lo, hi = i * s, (i + 1) * s
        for j in range(int(math.floor(lo)), min(n_in, int(math.ceil(hi)))):
            w[i, j] = (min(hi, j + 1) - max(lo, j)) / s
    return w


def test_resize_area_integer_fractional_and_odd_sizes(dev):
    """core/util.py:12-14,26 (tf.image.resize_area): integer ratios are box means, 5 -> 2 splits the middle pixel in half, and
    the odd sizes the reference's util.downsample sends here (int(H / num)) match the separable area weights in fp64."""
    from unflow_amd.core.util import resize_area, downsample
    x = torch.arange(2 * 4 * 6, dtype=torch.float32).reshape(2, 4, 6, 1)
    y = resize_area(x.to(dev), torch.empty(1, 2, 3, 1)).cpu()
    assert torch.allclose(y, x.reshape(2, 2, 2, 3, 2, 1).mean(dim=(2, 4)))
    z = resize_area(torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0]).view(1, 1, 5, 1).to(dev), torch.empty(1, 1, 2, 1)).cpu()
    assert torch.allclose(z.flatten(), torch.tensor([(1 + 2 + 1.5) / 2.5, (1.5 + 4 + 5) / 2.5]))
    g = torch.Generator().manual_seed(3)
    for (H, W, num) in ((375, 1241, 2), (93, 155, 4), (47, 31, 2)):
        t = torch.rand(2, H, W, 3, generator=g)
        got = downsample(t.to(dev), num).cpu().double()           # odd size -> the resize_area branch
        oh, ow = int(H / num), int(W / num)
        ref = torch.einsum('yh,bhwc,xw->byxc', _area_weights(H, oh), t.double(), _area_weights(W, ow))
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() < 2e-6
    with pytest.raises(TypeError):
        resize_area(x, torch.empty(1, 2, 3, 1))                    # host tensors: no CPU fallback
