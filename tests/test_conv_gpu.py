"""-m gpu: conv / conv_transpose fwd, dgrad, wgrad (fp32 MFMA implicit GEMM) through the C ABI vs the
torch-CPU oracle restatement of slim.conv2d / slim.conv2d_transpose (oracle/model_ref.py).

Tolerance: fp32 accumulation in a different order (MFMA k-order, split-K) -> relative error
<= 2e-5 of the output scale; stated per assertion."""
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def pad_c(c):
    return (c + 3) // 4 * 4


# (B, H, W, Cin, Cout, k, stride)
CONV_CASES = [
    (2, 24, 32, 3, 64, 7, 2),      # conv1-like: Cin padded 3 -> 4, multi-tap K tiles
    (2, 16, 24, 64, 128, 5, 2),    # conv2-like
    (1, 12, 16, 128, 256, 5, 2),
    (2, 12, 16, 256, 32, 1, 1),    # conv_redir
    (1, 12, 16, 473, 256, 3, 1),   # conv3_1: Cin 473 -> 476
    (2, 6, 8, 256, 512, 3, 2),     # conv4-like, asymmetric pad (0,1)
    (8, 6, 8, 512, 1024, 3, 2),    # conv6 size: split-K path
    (8, 6, 8, 1024, 1024, 3, 1),   # conv6_1: split-K
    (1, 10, 14, 36, 40, 3, 1),     # odd sizes, N tail
    (2, 6, 8, 1026, 2, 3, 1),      # flow5 head: skinny kernels
    (2, 12, 16, 194, 2, 3, 1),     # flow2 head
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_vs_oracle(case, dev):
    from unflow_amd.core import layers
    from oracle import model_ref as M
    B, H, W, Cin, Cout, k, stride = case
    g = torch.Generator().manual_seed(zlib.crc32(str(case).encode()))
    act = Cout > 4
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(k, k, Cin, Cout, generator=g) * (1.0 / np.sqrt(k * k * Cin))
    b = torch.randn(Cout, generator=g) * 0.1
    xr = x.clone().requires_grad_()
    wr = w.clone().requires_grad_()
    br = b.clone().requires_grad_()
    y_ref = M.conv2d(xr.permute(0, 3, 1, 2), wr, br, stride, act=act).permute(0, 2, 3, 1)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)

    Cp = pad_c(Cin)
    # place x inside a wider buffer (channel-slice view), padded channels zero
    xbuf = torch.zeros(B, H, W, Cp + 8, device=dev)
    xbuf[..., :Cin] = x.to(dev)
    xv = xbuf[..., :Cp]
    wp = torch.zeros(k, k, Cp, Cout, device=dev)
    wp[:, :, :Cin] = w.to(dev)
    Ho, Wo = layers.out_hw(H, W, stride)
    ybuf = torch.full((B, Ho, Wo, Cout + 4), 7.0, device=dev)
    yv = ybuf[..., 2:2 + Cout] if Cout % 4 == 0 and False else ybuf[..., :Cout]
    layers.conv2d_fwd(xv, wp, b.to(dev), yv, stride, leaky=act)
    assert rel_err(yv, y_ref) < 2e-5
    assert torch.all(ybuf[..., Cout:] == 7.0)   # neighbours of the slice untouched

    # dz = dy * leaky'(y)
    dz = torch.zeros(B, Ho, Wo, pad_c(Cout), device=dev)
    dz[..., :Cout] = gy.to(dev)
    dzv = dz[..., :Cout]
    if act:
        # elements with |y| ~ 1e-7 may land on different sides of the leaky kink on CPU and GPU; the
        # kink is checked where it is unambiguous and the oracle's side is used for the rest
        layers.leaky_bwd_inplace(dzv, yv)
        dz_ref = (gy * torch.where(y_ref.detach() > 0, 1.0, 0.1)).to(dev)
        clear = (y_ref.detach().abs() > 1e-4).to(dev)
        assert torch.equal(dzv[clear], dz_ref[clear])
        dzv.copy_(dz_ref)
    dx = torch.full((B, H, W, Cp), 3.0, device=dev)
    layers.conv2d_bwd_data(dzv, wp, dx, stride, accumulate=False)
    assert rel_err(dx[..., :Cin], xr.grad) < 2e-5
    # accumulate + activation-derivative epilogue
    base = torch.randn(B, H, W, Cp, generator=g).to(dev)
    dx2 = base.clone()
    src = torch.randn(B, H, W, Cp, generator=g).to(dev)
    layers.conv2d_bwd_data(dzv, wp, dx2, stride, accumulate=True, act_src=src, act_lo=0, act_hi=Cp // 2)
    expect = (base + dx)
    slope = torch.where(src > 0, torch.ones_like(src), torch.full_like(src, 0.1))
    expect[..., :Cp // 2] *= slope[..., :Cp // 2]
    assert rel_err(dx2, expect) < 2e-5

    dw = torch.empty_like(wp)
    db = torch.empty(Cout, device=dev)
    layers.conv2d_bwd_filter(xv, dzv, dw, db, stride)
    assert rel_err(dw[:, :, :Cin], wr.grad) < 3e-5
    assert rel_err(db, br.grad) < 3e-5
    if Cp != Cin:
        assert torch.all(dw[:, :, Cin:] == 0)   # padded weight rows get exactly zero gradient


# (B, H, W, Cin, Cout)   H,W = INPUT size; output is 2H x 2W
DECONV_CASES = [
    (8, 6, 8, 1024, 512),     # deconv5
    (2, 12, 16, 1026, 256),   # deconv4 (Cin 1026 -> 1028)
    (1, 24, 32, 770, 128),    # deconv3
    (1, 24, 32, 386, 64),     # deconv2-like
    (2, 12, 16, 2, 2),        # flowN_upM
]


@pytest.mark.parametrize("case", DECONV_CASES)
def test_conv2d_transpose_vs_oracle(case, dev):
    from unflow_amd.core import layers
    from oracle import model_ref as M
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(zlib.crc32(str(case).encode()))
    act = Cout > 2
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(4, 4, Cout, Cin, generator=g) * (1.0 / np.sqrt(4 * Cin))
    b = torch.randn(Cout, generator=g) * 0.1
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y_ref = M.conv2d_transpose(xr.permute(0, 3, 1, 2), wr, br, act=act).permute(0, 2, 3, 1)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)

    tiny = Cin == 2
    Cp = Cin if tiny else pad_c(Cin)
    xbuf = torch.zeros(B, H, W, Cp, device=dev)
    xbuf[..., :Cin] = x.to(dev)
    wp = torch.zeros(4, 4, Cout, Cp, device=dev)
    wp[..., :Cin] = w.to(dev)
    y = torch.empty(B, 2 * H, 2 * W, Cout, device=dev)
    layers.conv2d_transpose_fwd(xbuf, wp, b.to(dev), y, leaky=act)
    assert rel_err(y, y_ref) < 2e-5
    dz = gy.to(dev).contiguous()
    if act:
        dz = (gy * torch.where(y_ref.detach() > 0, 1.0, 0.1)).to(dev).contiguous()
    dx = torch.full((B, H, W, Cp), 5.0, device=dev)
    layers.conv2d_transpose_bwd_data(dz, wp, dx, accumulate=False)
    assert rel_err(dx[..., :Cin], xr.grad) < 2e-5
    dx2 = torch.ones(B, H, W, Cp, device=dev)
    layers.conv2d_transpose_bwd_data(dz, wp, dx2, accumulate=True)
    assert rel_err(dx2, dx + 1.0) < 2e-5
    dw = torch.empty_like(wp)
    db = torch.empty(Cout, device=dev)
    layers.conv2d_transpose_bwd_filter(xbuf, dz, dw, db)
    assert rel_err(dw[..., :Cin], wr.grad) < 3e-5
    assert rel_err(db, br.grad) < 3e-5


def test_conv_full_size_linearity(dev):
    """Property at benchmark size (B=8 directed samples, conv3_1 473->256 on 48x64): linear in x, and equal
    to the sum of per-channel-group convolutions (no oracle needed at this size)."""
    from unflow_amd.core import layers
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout, k = 8, 48, 64, 476, 256, 3
    x1 = torch.randn(B, H, W, Cin, generator=g).to(dev)
    x2 = torch.randn(B, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(k, k, Cin, Cout, generator=g) / np.sqrt(9 * Cin)).to(dev)
    y1 = torch.empty(B, H, W, Cout, device=dev)
    y2 = torch.empty_like(y1)
    y12 = torch.empty_like(y1)
    layers.conv2d_fwd(x1, w, None, y1, 1, leaky=False)
    layers.conv2d_fwd(x2, w, None, y2, 1, leaky=False)
    layers.conv2d_fwd(x1 + 2 * x2, w, None, y12, 1, leaky=False)
    assert rel_err(y12, y1 + 2 * y2) < 1e-5
    # channel split: conv(x[:, :240]) + conv(x[:, 240:]) == conv(x)
    ya = torch.empty_like(y1)
    yb = torch.empty_like(y1)
    layers.conv2d_fwd(x1[..., :240], w[:, :, :240].contiguous(), None, ya, 1, leaky=False)
    layers.conv2d_fwd(x1[..., 240:], w[:, :, 240:].contiguous(), None, yb, 1, leaky=False)
    assert rel_err(ya + yb, y1) < 1e-5


def test_both_conv_math_modes(dev):
    """The gather kernels default to the fp32-equivalent 3-way bf16 split on the bf16 matrix cores; UNFLOW_CONV_MATH=fp32
    puts them on v_mfma_f32_32x32x2_f32.  The library reads the knob once per process, so the other mode runs this file
    in a sub-process: every parity case must hold at the SAME tolerances in both modes."""
    import os
    import subprocess
    import sys
    if os.environ.get("UNFLOW_CONV_MATH_SUBTEST"):
        pytest.skip("already inside the sub-process")
    other = "bf16x3" if os.environ.get("UNFLOW_CONV_MATH", "bf16x3") == "fp32" else "fp32"
    env = dict(os.environ, UNFLOW_CONV_MATH=other, UNFLOW_CONV_MATH_SUBTEST="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
