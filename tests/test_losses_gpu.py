"""The loss module's callables (unflow_amd/core/losses.py, mirror of src/e2eflow/core/losses.py:12-322) through the C ABI vs the
oracle's fp64 restatement, one test per reference function + compute_losses in every mask mode."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(seed, B=2, H=96, W=128, flow_mag=2.0):
    g = torch.Generator().manual_seed(seed)
    im1 = torch.rand(B, H, W, 3, generator=g)
    im2 = (torch.roll(im1, shifts=(1, -2), dims=(1, 2)) * 0.95 + torch.rand(B, H, W, 3, generator=g) * 0.05).clamp(0, 1)
    fw = torch.randn(B, H, W, 2, generator=g) * flow_mag
    bw = -fw + torch.randn(B, H, W, 2, generator=g) * 0.4
    return im1, im2, fw, bw


def _close(got, ref, rel):
    got, ref = float(got), float(ref)
    assert abs(got - ref) <= rel * max(abs(ref), 1e-6), (got, ref)


def test_length_sq_vs_oracle(dev):
    from unflow_amd.core import losses as Lh
    from oracle import model_ref as M
    x = torch.randn(2, 33, 47, 5)
    got = Lh.length_sq(x.to(dev)).cpu()
    assert got.shape == (2, 33, 47, 1)
    torch.testing.assert_close(got, M.length_sq(x), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("case", [dict(), dict(beta=255.0), dict(alpha=0.5, epsilon=0.01), dict(truncate=0.3), dict(mask=1),
                                  dict(mask=3, beta=255.0), dict(mask=1, truncate=0.05, alpha=0.3)])
def test_charbonnier_loss_vs_oracle(case, dev):
    """charbonnier_loss(x, mask, truncate, alpha, beta, epsilon) (losses.py:298-322), defaults and every argument."""
    from unflow_amd.core import losses as Lh
    from oracle import model_ref as M
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 40, 56, 3, generator=g) * 0.2
    kw = dict(case)
    mc = kw.pop('mask', None)
    mask = None if mc is None else (torch.rand(2, 40, 56, mc, generator=g) > 0.3).float()
    got = Lh.charbonnier_loss(x.to(dev), None if mask is None else mask.to(dev), **kw)
    ref = M.charbonnier_loss(x.double(), None if mask is None else mask.double(), **kw)
    _close(got, ref, 2e-5)
    if mask is not None and mc == 1:       # a [1,H,W,1] mask broadcasts over the batch
        got = Lh.charbonnier_loss(x.to(dev), mask[:1].to(dev), **kw)
        _close(got, M.charbonnier_loss(x.double(), mask[:1].double(), **kw), 2e-5)
    with pytest.raises(ValueError):
        Lh.charbonnier_loss(x.to(dev), torch.ones(2, 40, 56, 2, device=dev))


def test_photometric_smoothness_gradient_losses_vs_oracle(dev):
    """photometric_loss (:198-199), smoothness_loss (:250-255), second_order_loss (:290-295), gradient_loss (:241-247),
    ternary_loss (:90-122) as stand-alone callables."""
    from unflow_amd.core import losses as Lh
    from unflow_amd.core.image_warp import image_warp
    from oracle import model_ref as M
    im1, im2, fw, bw = _inputs(11)
    mask = Lh.create_border_mask(im1, 0.1)
    im2w = M.image_warp(im2.double(), fw.double())
    _close(Lh.photometric_loss((im1 - im2w.float()).to(dev), mask.to(dev)), M.photometric_loss(im1.double() - im2w, mask.double()), 5e-5)
    _close(Lh.smoothness_loss(fw.to(dev)), M.smoothness_loss(fw.double()), 2e-5)
    _close(Lh.second_order_loss(fw.to(dev)), M.second_order_loss(fw.double()), 2e-5)
    _close(Lh.gradient_loss(im1.to(dev), im2w.float().to(dev), mask.to(dev)), M.gradient_loss(im1.double(), im2w, mask.double()), 5e-5)
    for D in (1, 2, 3):
        _close(Lh.ternary_loss(im1.to(dev), im2w.float().to(dev), mask.to(dev), max_distance=D),
               M.ternary_loss(im1.double(), im2w, mask.double(), max_distance=D), 1e-4)
    # the HIP image_warp feeding the HIP losses: same value as the chain above
    got = Lh.photometric_loss(im1.to(dev) - image_warp(im2.to(dev), fw.to(dev)), mask.to(dev))
    _close(got, M.photometric_loss(im1.double() - im2w, mask.double()), 1e-4)


@pytest.mark.parametrize("mode", ['', 'fb', 'disocc'])
@pytest.mark.parametrize("border", [True, False])
def test_compute_losses_vs_oracle(mode, border, dev):
    """compute_losses(im1, im2, flow_fw, flow_bw, border_mask, mask_occlusion, data_max_distance) (losses.py:16-87): all
    eight entries in every mask mode, with the border mask and with create_outgoing_mask.  The thresholded masks (fb_occ,
    disocc, outgoing) are discontinuous: a pixel whose bit differs between fp32 and fp64 moves a masked mean by 1 / (B*H*W),
    hence the looser bound on the terms that sum a mask."""
    from unflow_amd.core import losses as Lh
    from oracle import model_ref as M
    im1, im2, fw, bw = _inputs(23 + len(mode), flow_mag=3.0)
    bm = Lh.create_border_mask(im1, 0.1) if border else None
    got = Lh.compute_losses(im1.to(dev), im2.to(dev), fw.to(dev), bw.to(dev), border_mask=None if bm is None else bm.to(dev),
                            mask_occlusion=mode, data_max_distance=2)
    ref = M.compute_losses(im1.double(), im2.double(), fw.double(), bw.double(), border_mask=None if bm is None else bm.double(),
                           mask_occlusion=mode, data_max_distance=2)
    assert set(got) == {'sym', 'occ', 'photo', 'grad', 'smooth_1st', 'smooth_2nd', 'fb', 'ternary'}
    for k, v in ref.items():
        _close(got[k], v, 2e-5 if k in ('smooth_1st', 'smooth_2nd') else 2e-3)
    with pytest.raises(ValueError):
        Lh.compute_losses(im1.to(dev), im2.to(dev), fw.to(dev), bw.to(dev), mask_occlusion='both')
