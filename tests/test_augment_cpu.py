"""CPU: the host half of the augmentation mirror (core/augment.py of the reference: draws, theta composition, crop) and
the oracle's spatial transformer on hand-checkable cases.  The GPU half is in tests/test_engine_gpu.py."""
import math

import numpy as np
import torch


def test_affine_theta_composition_matches_oracle_and_formula():
    from unflow_amd.core import augment as A
    from oracle import model_ref as M
    g = torch.Generator().manual_seed(0)
    B = 5
    tx, ty = torch.rand(B, generator=g) - 0.5, torch.rand(B, generator=g) - 0.5
    rot = (torch.rand(B, generator=g) - 0.5) * 60
    sc = 0.8 + 0.4 * torch.rand(B, generator=g)
    flip = torch.tensor([1.0, -1.0, 1.0, -1.0, -1.0])
    th = A.affine_theta(tx, ty, rot, sc, flip)
    assert th.shape == (B, 2, 3)
    assert torch.allclose(th, M.affine_theta(tx, ty, rot, sc, flip), atol=0, rtol=0)
    # augment.py:31-48: [[cos,-sin,tx],[sin,cos,ty]] @ diag(scale*flip, scale, 1)
    for b in range(B):
        r = math.radians(float(rot[b]))
        t1 = np.array([[math.cos(r), -math.sin(r), float(tx[b])], [math.sin(r), math.cos(r), float(ty[b])]])
        t2 = np.diag([float(sc[b] * flip[b]), float(sc[b]), 1.0])
        assert np.allclose(th[b].numpy(), t1 @ t2, atol=1e-6)


def test_training_draws_have_reference_ranges():
    from unflow_amd.core import augment as A
    aug = A.draw_training_augmentation(256, torch.Generator().manual_seed(1))
    tg, tl = aug['theta_global'], aug['theta_local']
    # no rotation / translation in the training config (unsupervised.py:40-50): pure scale, global one may flip x
    assert torch.all(tg[:, :, 2] == 0) and torch.all(tl[:, :, 2] == 0)
    assert torch.all(tg[:, 0, 1] == 0) and torch.all(tg[:, 1, 0] == 0)
    assert torch.all((tg[:, 1, 1] >= 0.9) & (tg[:, 1, 1] <= 1.1))
    assert torch.all((tg[:, 0, 0].abs() >= 0.9) & (tg[:, 0, 0].abs() <= 1.1))
    assert (tg[:, 0, 0] < 0).any() and (tg[:, 0, 0] > 0).any()          # horizontal_flipping=True
    assert torch.all(tl[:, 0, 0] > 0)                                     # the local transform never flips
    assert torch.all((aug['contrast'] >= -0.3) & (aug['contrast'] <= 0.3))
    assert torch.all((aug['gamma'] >= 0.7) & (aug['gamma'] <= 1.5))
    assert torch.all((aug['colour'] >= 0.9) & (aug['colour'] <= 1.1)) and aug['colour'].shape == (256, 3)
    assert aug['noise'].shape == (256,) and abs(float(aug['noise'].std()) - 0.04) < 0.01      # one value per SAMPLE
    assert abs(float(aug['brightness'].std()) - 0.02) < 0.006


def test_random_crop_same_window_for_all_tensors():
    from unflow_amd.core import augment as A
    a = torch.arange(2 * 10 * 12 * 3, dtype=torch.float32).view(2, 10, 12, 3)
    b = a + 1000
    ca, cb = A.random_crop([a, b], [2, 6, 8, 3], seed=3)
    assert ca.shape == (2, 6, 8, 3) and torch.equal(cb, ca + 1000)
    ca2, _ = A.random_crop([a, b], [2, 6, 8, 3], seed=3)
    assert torch.equal(ca, ca2)
    # with two tensors the limit is the elementwise minimum of the shapes (augment.py:118-121)
    c_small, c_big = A.random_crop([a[:, :8], b], [2, 8, 12, 3], seed=0)
    assert c_small.shape == (2, 8, 12, 3) and c_big.shape == (2, 8, 12, 3)


def test_oracle_transformer_known_answers():
    """spatial_transformer.py:56-175 on cases that can be checked by hand."""
    from oracle import model_ref as M
    B, H, W = 1, 4, 5
    ramp = torch.arange(W, dtype=torch.float32).view(1, 1, W, 1).expand(B, H, W, 1).contiguous()
    eye = torch.tensor([[[1.0, 0, 0], [0, 1.0, 0]]])
    out = M.stn_transformer(ramp, eye)
    # x_t = linspace(-1,1,5) -> x = (x_t+1)*5/2 = 0, 1.25, 2.5, 3.75, 5: a 1.25x zoom about the left edge, and the
    # last column samples x = W where both clipped taps coincide with opposite-sign weights -> 0; same for the last row
    assert torch.allclose(out[0, 0, :4, 0], torch.tensor([0.0, 1.25, 2.5, 3.75]), atol=1e-6)
    assert out[0, :, 4].abs().max() < 1e-6 and out[0, 3].abs().max() < 1e-6
    # horizontal flip (theta[0,0] = -1): x = 5, 3.75, 2.5, 1.25, 0
    flip = torch.tensor([[[-1.0, 0, 0], [0, 1.0, 0]]])
    outf = M.stn_transformer(ramp, flip)
    assert torch.allclose(outf[0, 0, 1:, 0], torch.tensor([3.75, 2.5, 1.25, 0.0]), atol=1e-6)
    assert abs(float(outf[0, 0, 0, 0])) < 1e-6


def test_oracle_photometric_known_answers():
    from oracle import model_ref as M
    im = torch.tensor([0.0, 0.25, 0.5, 1.0]).view(1, 1, 4, 1).expand(1, 1, 4, 3).contiguous()
    one, zero = torch.ones(1), torch.zeros(1)
    # contrast 1 (x2), brightness 0.1, colour 1, gamma 2 (sqrt), noise 0.05: clamp((x*2+0.1))^(1/2) + 0.05
    out = M.random_photometric_apply([im], one, 2 * one, torch.ones(1, 3), 0.05 * one, 0.1 * one)[0]
    ref = torch.tensor([0.1, 0.6, 1.0, 1.0]).sqrt() + 0.05
    assert torch.allclose(out[0, 0, :, 0], ref, atol=1e-6)
    assert out.requires_grad is False
