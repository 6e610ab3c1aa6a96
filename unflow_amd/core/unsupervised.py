"""Mirror of src/e2eflow/core/unsupervised.py:27-164."""
import torch

from .engine import LOSSES, DEFAULT_PARAMS, CHANNEL_MEAN  # noqa: F401
from .flownet import get_engine


def unsupervised_loss(batch, params, normalization=None, augment=True, return_flow=False, engine=None,
                      backward=False, generator=None):
    """batch = (im1, im2), NHWC float32 in [0,255].  `normalization` is accepted for signature compatibility; the
    channel means are the reference's (core/input.py:45).  augment: True draws the random affine + photometric
    augmentation of unsupervised.py:39-58 (host RNG `generator`), a dict replays given draws
    (core.augment.draw_training_augmentation), False/None disables it.  With backward=True the parameter
    gradients are left in engine.G (what opt.compute_gradients would return)."""
    if normalization is not None:
        mean = [float(v) for v in normalization[0]]
        if max(abs(a - b) for a, b in zip(mean, CHANNEL_MEAN)) > 1e-3:
            raise NotImplementedError("custom channel means")
    im1, im2 = batch
    B, H, W, _ = im1.shape
    eng = engine or get_engine(B, H, W, params=params, device=im1.device)     # flownet / full_res / train_all / loss keys
    if augment is True:
        from .augment import draw_training_augmentation
        augment = draw_training_augmentation(B, generator)
    eng.set_input(im1, im2, augment=augment or None)
    eng.forward_net()
    loss = eng.forward_loss(with_grad=backward)
    if backward:
        eng.backward_net()
    if not return_flow:
        return loss[0]
    fw, bw = eng.final_flows()
    return loss[0], fw, bw
