"""CPU (-m "not gpu"): the oracle is pinned against every known-answer vector the reference's own tests hold
for the hot path (tests/golden/ref_kats.json, transcribed from src/e2eflow/test/**), cross-checked against an
independent dense fp64 restatement, and its hand-written backward passes are checked against numeric Jacobians
with the reference's own recipe (gradient_checker, rtol/atol 1e-3: test/ops/correlation.py:21-28)."""
import numpy as np
import pytest
import torch

from oracle import model_ref as M


def _np(a):
    return np.array(a, dtype=np.float32)


# ------------------------------------------------------------------ reference KATs
def test_kat_correlation(kats, oracle_lib):
    for name in ("correlation_trivial", "correlation_batch"):
        k = kats[name]
        out = oracle_lib.correlation(_np(k["first"]), _np(k["second"]), stride_1=1, **k["attrs"])
        np.testing.assert_allclose(out, _np(k["expected"]), rtol=1e-6, atol=1e-6)


def test_kat_flownetc_correlation_shape(kats, oracle_lib):
    k = kats["flownetc_correlation_shape"]
    assert oracle_lib.correlation_out_shape(*k["in_hw"], **k["attrs"]) == tuple(k["expected_chw"])


@pytest.mark.parametrize("name,fn", [("warp_move", "backward_warp"), ("warp_interpolate", "backward_warp"),
                                     ("backward_warp_batches", "backward_warp"), ("warp_move", "image_warp"),
                                     ("warp_interpolate", "image_warp"), ("image_warp_batches", "image_warp")])
def test_kat_warps(kats, oracle_lib, name, fn):
    k = kats[name]
    im, fl = _np(k["image"])[..., None], _np(k["flow"])
    out = getattr(oracle_lib, fn)(im, fl)
    np.testing.assert_allclose(out[..., 0], _np(k["expected"]), rtol=1e-6, atol=1e-6)
    if fn == "image_warp":   # the torch restatement used by the model oracle must agree too
        t = M.image_warp(torch.tensor(im), torch.tensor(fl)).numpy()
        np.testing.assert_allclose(t[..., 0], _np(k["expected"]), rtol=1e-6, atol=1e-6)


def test_kat_downsample(kats, oracle_lib):
    k = kats["downsample"]
    out = oracle_lib.downsample(_np(k["image"])[..., None], k["scale"])
    assert np.array_equal(out[..., 0], _np(k["expected"]))
    t = M.downsample(torch.tensor(_np(k["image"])[..., None]), k["scale"]).numpy()
    assert np.array_equal(t[..., 0], _np(k["expected"]))


def test_kat_forward_warp_zero_flow(kats, oracle_lib):
    out = oracle_lib.forward_warp(np.zeros((1, 20, 20, 2), np.float32))
    assert abs(out[0, 10, 10, 0] - kats["forward_warp_zero_flow_interior"]["expected"]) < 2e-6
    assert out[0, 0, 0, 0] < out[0, 10, 10, 0]   # borders receive fewer splats


def test_kat_smoothness_deltas(kats):
    k = kats["smoothness_deltas"]
    flow = torch.zeros(1, 3, 3, 2)
    flow[0, :, :, 0] = torch.tensor(k["flow_u"], dtype=torch.float32)
    flow[0, :, :, 1] = torch.tensor(k["flow_v"], dtype=torch.float32)
    du, dv, mask = M._smoothness_deltas(flow)
    assert torch.equal(mask[0, :, :, 0], torch.tensor(k["mask_x"], dtype=torch.float32))
    assert torch.equal(mask[0, :, :, 1], torch.tensor(k["mask_y"], dtype=torch.float32))
    for d in (du, dv):
        d = d * mask
        assert torch.equal(d[0, :, :, 0], torch.tensor(k["delta_x"], dtype=torch.float32))
        assert torch.equal(d[0, :, :, 1], torch.tensor(k["delta_y"], dtype=torch.float32))


@pytest.mark.parametrize("name", ["outgoing_mask_all_directions", "outgoing_mask_large_movement"])
def test_kat_outgoing_mask(kats, name):
    k = kats[name]
    flow = torch.zeros(1, 3, 3, 2)
    flow[0, :, :, 0] = torch.tensor(k["flow_u"], dtype=torch.float32)
    flow[0, :, :, 1] = torch.tensor(k["flow_v"], dtype=torch.float32)
    assert torch.equal(M.create_outgoing_mask(flow)[0, :, :, 0], torch.tensor(k["expected"], dtype=torch.float32))


def test_kat_gradient_loss(kats):
    k = kats["gradient_loss_constant_offset"]
    im1 = torch.tensor(k["im1_channel"], dtype=torch.float32).view(1, 3, 3, 1).repeat(1, 1, 1, 3)
    im2 = torch.tensor(k["im2_channel"], dtype=torch.float32).view(1, 3, 3, 1).repeat(1, 1, 1, 3)
    loss = M.gradient_loss(im1, im2, torch.ones(1, 3, 3, 1))
    assert abs(loss.item() - k["expected"]) < k["atol"]


# ------------------------------------------------------------------ independent cross-checks
def test_correlation_c_vs_dense_fp64(oracle_lib):
    rs = np.random.RandomState(0)
    a = rs.randn(2, 16, 12, 14).astype(np.float32)
    b = rs.randn(2, 16, 12, 14).astype(np.float32)
    attrs = dict(max_displacement=4, pad=4, stride_2=2)
    c = oracle_lib.correlation(a, b, **attrs)
    ta, tb = torch.tensor(a).double().requires_grad_(), torch.tensor(b).double().requires_grad_()
    d = M.correlation_dense(ta, tb, **attrs)
    np.testing.assert_allclose(c, d.detach().numpy(), rtol=1e-5, atol=1e-6)
    go = rs.randn(*c.shape).astype(np.float32)
    (d * torch.tensor(go).double()).sum().backward()
    g0, g1 = oracle_lib.correlation_grad(go, a, b, **attrs)
    np.testing.assert_allclose(g0, ta.grad.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(g1, tb.grad.numpy(), rtol=1e-4, atol=1e-6)


def test_image_warp_grad_c_vs_torch(oracle_lib):
    rs = np.random.RandomState(3)
    im = rs.rand(2, 9, 11, 3).astype(np.float32)
    fl = (rs.randn(2, 9, 11, 2) * 3).astype(np.float32)
    gw = rs.randn(2, 9, 11, 3).astype(np.float32)
    tim, tfl = torch.tensor(im).double().requires_grad_(), torch.tensor(fl).double().requires_grad_()
    (M.image_warp(tim, tfl) * torch.tensor(gw).double()).sum().backward()
    d_im, d_fl = oracle_lib.image_warp_grad(gw, im, fl)
    np.testing.assert_allclose(d_im, tim.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(d_fl, tfl.grad.numpy(), rtol=1e-4, atol=1e-5)


def _numeric_jacobian_check(f, x, g_analytic, eps=1e-2, tol=1e-3):
    """Central differences of sum(f(x) * w) vs the analytic gradient, tf gradient_checker style."""
    rs = np.random.RandomState(1)
    out = f(x)
    w = rs.randn(*out.shape).astype(np.float32)
    ga = g_analytic(w)
    idx = rs.choice(x.size, size=min(40, x.size), replace=False)
    for i in idx:
        xp, xm = x.copy().ravel(), x.copy().ravel()
        xp[i] += eps
        xm[i] -= eps
        num = ((f(xp.reshape(x.shape)).astype(np.float64) - f(xm.reshape(x.shape)).astype(np.float64)) * w).sum() / (2 * eps)
        assert abs(num - ga.ravel()[i]) <= tol + tol * abs(num), (i, num, ga.ravel()[i])


def test_correlation_3x3_jacobian(kats, oracle_lib):
    """The reference's disabled test_correlation_3x3 inputs (correlation.py:74-89) through its Jacobian recipe."""
    k = kats["correlation_3x3_inputs"]
    a, b = _np(k["first"]), _np(k["second"])
    attrs = dict(stride_1=1, **k["attrs"])
    _numeric_jacobian_check(lambda x: oracle_lib.correlation(x, b, **attrs), a,
                            lambda w: oracle_lib.correlation_grad(w, a, b, **attrs)[0])
    _numeric_jacobian_check(lambda x: oracle_lib.correlation(a, x, **attrs), b,
                            lambda w: oracle_lib.correlation_grad(w, a, b, **attrs)[1])


def test_backward_warp_jacobian(oracle_lib):
    rs = np.random.RandomState(2)
    im = rs.rand(1, 6, 7, 2).astype(np.float32)
    fl = (rs.rand(1, 6, 7, 2) * 2.4 - 1.2 + 0.05).astype(np.float32)
    fl = np.where(np.abs(fl - np.round(fl)) < 0.05, fl + 0.1, fl).astype(np.float32)   # stay away from the bilinear kinks
    _numeric_jacobian_check(lambda x: oracle_lib.backward_warp(im, x), fl,
                            lambda w: oracle_lib.backward_warp_grad(w, im, fl), eps=1e-3, tol=2e-3)


def test_forward_warp_jacobian(oracle_lib):
    """src/e2eflow/test/ops/forward_warp.py:9-19 (Jacobian only; the reference pins no values)."""
    rs = np.random.RandomState(4)
    fl = (rs.randn(1, 10, 10, 2) * 1.5).astype(np.float32)
    _numeric_jacobian_check(lambda x: oracle_lib.forward_warp(x), fl,
                            lambda w: oracle_lib.forward_warp_grad(w, fl), eps=1e-3, tol=2e-3)


def test_oracle_error_conditions(oracle_lib):
    a = np.zeros((1, 2, 8, 8), np.float32)
    with pytest.raises(ValueError, match="kernel_size must be odd"):
        oracle_lib.correlation(a, a, kernel_size=2)
    with pytest.raises(ValueError, match="Invalid correlation settings"):
        oracle_lib.correlation(a, a, max_displacement=20, pad=0)
    with pytest.raises(ValueError, match="Input shapes have to be the same"):
        oracle_lib.correlation(a, np.zeros((1, 2, 8, 9), np.float32))
    with pytest.raises(ValueError, match="divisible by scale"):
        oracle_lib.downsample(np.zeros((1, 6, 8, 3), np.float32), 4)


# ------------------------------------------------------------------ TF-semantics pieces of the model oracle
def test_same_padding_geometry():
    # conv1 k7 s2 -> (2,3); conv2/3 k5 s2 -> (1,2); conv4/5/6 k3 s2 -> (0,1); stride 1 symmetric
    assert M.same_pads(384, 7, 2) == (2, 3)
    assert M.same_pads(192, 5, 2) == (1, 2)
    assert M.same_pads(48, 3, 2) == (0, 1)
    assert M.same_pads(48, 3, 1) == (1, 1)
    assert M.same_pads(48, 1, 1) == (0, 0)


def test_conv_transpose_is_gradient_of_conv():
    """slim.conv2d_transpose(k4, s2, SAME) == d/dx of a SAME k4 s2 conv on the 2x grid (Appendix B)."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 3, 5, 6, generator=g, dtype=torch.float64)            # NCHW, Cin = 3
    w = torch.randn(4, 4, 2, 3, generator=g, dtype=torch.float64)            # [k,k,Cout=2,Cin=3]
    y = M.conv2d_transpose(x, w, None, act=False)
    assert tuple(y.shape) == (1, 2, 10, 12)
    big = torch.zeros(1, 2, 10, 12, dtype=torch.float64, requires_grad=True)  # the conv whose input-gradient it is
    w_conv = w.permute(0, 1, 2, 3)                                            # HWIO with I = Cout(2), O = Cin(3)
    out = M.conv2d(big, w_conv, None, stride=2, act=False)                    # -> [1,3,5,6]
    (out * x).sum().backward()
    assert torch.allclose(big.grad, y, atol=1e-12)


def test_resize_bilinear_tf1_identity_and_scale():
    x = torch.arange(2 * 3 * 4 * 2, dtype=torch.float32).reshape(2, 3, 4, 2)
    assert torch.equal(M.resize_bilinear_tf1(x, 3, 4), x)
    up = M.resize_bilinear_tf1(x, 6, 8)
    assert torch.equal(up[:, ::2, ::2], x)                      # src = dst/2: even outputs hit source pixels exactly
    assert torch.allclose(up[:, 1, 0], 0.5 * (x[:, 0, 0] + x[:, 1, 0]))
    assert torch.equal(up[:, 5, :], up[:, 4, :])                # last row: hi index clamps to in-1


def test_parameter_count():
    P = M.init_params('C', 0)
    assert sum(v.numel() for v in P.values()) == 39175298       # 39.17 M parameters


def test_adam_tf_form():
    P = {'w': torch.tensor([1.0, -2.0])}
    G = {'w': torch.tensor([0.5, -0.25])}
    m = {'w': torch.zeros(2)}
    v = {'w': torch.zeros(2)}
    M.adam_step_tf(P, G, m, v, 1, 0.1)
    # first step of TF Adam moves every weight by lr * g/|g| up to the epsilon term
    assert torch.allclose(P['w'], torch.tensor([0.9, -1.9]), atol=1e-6)


def test_lr_schedule():
    p = dict(learning_rate=1e-4, decay_interval=100000, decay_after=200000)
    assert M.learning_rate_at(p, 0) == 1e-4
    assert M.learning_rate_at(p, 199999) == 1e-4
    assert M.learning_rate_at(p, 200000) == 1e-4
    assert M.learning_rate_at(p, 300000) == 0.5e-4
    assert M.learning_rate_at(p, 450000) == 0.25e-4


def test_fp32_vs_fp64_loss_small():
    """fp32 oracle vs its fp64 shadow on a small FlowNetC step (bounds the oracle's own rounding)."""
    P = M.init_params('C', 1)
    g = torch.Generator().manual_seed(2)
    im1 = torch.rand(1, 128, 128, 3, generator=g) * 255
    im2 = torch.rand(1, 128, 128, 3, generator=g) * 255
    l32 = M.unsupervised_loss(P, im1, im2).item()
    P64 = {k: v.double() for k, v in P.items()}
    l64 = M.unsupervised_loss(P64, im1.double(), im2.double()).item()
    assert abs(l32 - l64) <= 1e-4 * abs(l64)
