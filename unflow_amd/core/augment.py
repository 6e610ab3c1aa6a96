"""Mirror of src/e2eflow/core/augment.py (random_affine :7-56, random_photometric :59-110, random_crop :113-134) on
the HIP kernels of csrc/augment.hip.

The random draws are made on the host with a torch.Generator (TF's RNG streams cannot be reproduced; what is
checked against the oracle is the deterministic transform given the draws), the resampling / photometric maths run
on the GPU.  Every function can return the draws it used so a test or a caller can replay them.
"""
import math

import torch

from .. import _lib
from .._lib import check, ptr, stream


def _uniform(n, lo, hi, generator):
    return torch.rand(n, generator=generator, dtype=torch.float32) * (hi - lo) + lo


def affine_theta(tx, ty, rot_deg, scale, flip=None):
    """theta = [[cos,-sin,tx],[sin,cos,ty]] @ diag(scale*flip, scale, 1) (augment.py:31-48).  [B] fp32 host tensors."""
    rad = (rot_deg * math.pi) / 180.0
    sx = scale if flip is None else scale * flip
    c, s = torch.cos(rad), torch.sin(rad)
    row0 = torch.stack([c * sx, -s * scale, tx], 1)
    row1 = torch.stack([s * sx, c * scale, ty], 1)
    return torch.stack([row0, row1], 1).contiguous()


def draw_affine(num_batch, *, max_translation_x=0.0, max_translation_y=0.0, max_rotation=0.0, min_scale=1.0,
                max_scale=1.0, horizontal_flipping=False, generator=None):
    """The draws of augment.py:23-39 -> theta [B,2,3] (host fp32)."""
    tx = _uniform(num_batch, -max_translation_x, max_translation_x, generator)
    ty = _uniform(num_batch, -max_translation_y, max_translation_y, generator)
    rot = _uniform(num_batch, -max_rotation, max_rotation, generator)
    scale = _uniform(num_batch, min_scale, max_scale, generator)
    flip = None
    if horizontal_flipping:
        f = _uniform(num_batch, 0.0, 1.0, generator)
        flip = torch.where(f > 0.5, -torch.ones(num_batch), torch.ones(num_batch))
    return affine_theta(tx, ty, rot, scale, flip)


def transformer(U, theta, out_size=None, out=None, n_samples=None):
    """spatial_transformer.transformer (spatial_transformer.py:19): U [n_u,H,W,C] CUDA NHWC (channel-slice views
    allowed), theta [n_theta,2,3] (host or device).  Output sample b reads U[b % n_u] with theta[b % n_theta]."""
    assert U.is_cuda and U.dtype == torch.float32 and U.dim() == 4 and U.stride(3) == 1
    n_u, H, W, C = U.shape
    ld_u = U.stride(2)
    assert U.stride(1) == W * ld_u and U.stride(0) == H * W * ld_u
    theta = theta.to(device=U.device, dtype=torch.float32).reshape(-1, 6).contiguous()
    B = n_samples or max(n_u, theta.shape[0])
    Ho, Wo = (H, W) if out_size is None else out_size
    if out is None:
        out = torch.empty(B, Ho, Wo, C, dtype=torch.float32, device=U.device)
    ld_o = out.stride(2)
    check(_lib.lib().unflow_stn_affine_fwd(ptr(U), n_u, ld_u, ptr(theta), theta.shape[0], ptr(out), ld_o, B, H, W, C,
                                           Ho, Wo, stream()), "stn_affine")
    return out


def random_affine(tensors, *, max_translation_x=0.0, max_translation_y=0.0, max_rotation=0.0, min_scale=1.0,
                  max_scale=1.0, horizontal_flipping=False, generator=None, theta=None, return_theta=False):
    """augment.random_affine (augment.py:7-56): every tensor of the list gets the same per-sample transform.
    `theta` replays given draws."""
    num_batch = tensors[0].shape[0]
    if theta is None:
        theta = draw_affine(num_batch, max_translation_x=max_translation_x, max_translation_y=max_translation_y,
                            max_rotation=max_rotation, min_scale=min_scale, max_scale=max_scale,
                            horizontal_flipping=horizontal_flipping, generator=generator)
    out = [transformer(t, theta) for t in tensors]
    return (out, theta) if return_theta else out


def draw_photometric(num_batch, *, noise_stddev=0.0, min_contrast=0.0, max_contrast=0.0, brightness_stddev=0.0,
                     min_colour=1.0, max_colour=1.0, min_gamma=1.0, max_gamma=1.0, generator=None):
    """The draws of augment.py:78-91 (noise and brightness are ONE value per sample, shape [num_batch,1])."""
    d = dict(contrast=_uniform(num_batch, min_contrast, max_contrast, generator),
             gamma=_uniform(num_batch, min_gamma, max_gamma, generator),
             colour=_uniform(num_batch * 3, min_colour, max_colour, generator).view(num_batch, 3))
    z = torch.zeros(num_batch)
    d['noise'] = torch.randn(num_batch, generator=generator) * noise_stddev if noise_stddev > 0.0 else z
    d['brightness'] = torch.randn(num_batch, generator=generator) * brightness_stddev if brightness_stddev > 0.0 \
        else z.clone()
    return d


def photometric(im, draws, out=None, mean=None):
    """The deterministic part of random_photometric (augment.py:93-108) for one image batch [N,H,W,>=3 stride]."""
    assert im.is_cuda and im.dtype == torch.float32 and im.dim() == 4
    N, H, W, _ = im.shape
    dev = im.device
    if out is None:
        out = torch.empty(N, H, W, 3, dtype=torch.float32, device=dev)
    g = {k: v.to(device=dev, dtype=torch.float32).contiguous() for k, v in draws.items()}
    n_par = g['contrast'].numel()
    mean_host = None if mean is None else (_lib.ctypes.c_float * 3)(*[float(v) for v in mean])
    check(_lib.lib().unflow_photometric_augment(ptr(im), im.stride(2), ptr(out), out.stride(2), ptr(g['contrast']),
                                                ptr(g['brightness']), ptr(g['colour']), ptr(g['gamma']),
                                                ptr(g['noise']), n_par, mean_host, N, H, W, stream()), "photometric")
    return out


def random_photometric(ims, *, noise_stddev=0.0, min_contrast=0.0, max_contrast=0.0, brightness_stddev=0.0,
                       min_colour=1.0, max_colour=1.0, min_gamma=1.0, max_gamma=1.0, generator=None, draws=None,
                       return_draws=False):
    """augment.random_photometric (augment.py:59-110): ims = list of [B,H,W,3] batches in [0,1]."""
    if draws is None:
        draws = draw_photometric(ims[0].shape[0], noise_stddev=noise_stddev, min_contrast=min_contrast,
                                 max_contrast=max_contrast, brightness_stddev=brightness_stddev,
                                 min_colour=min_colour, max_colour=max_colour, min_gamma=min_gamma,
                                 max_gamma=max_gamma, generator=generator)
    out = [photometric(im, draws) for im in ims]
    return (out, draws) if return_draws else out


def random_crop(tensors, size, seed=None, name=None):
    """augment.random_crop (augment.py:113-134): the same random window of `size` (full-rank, like tf.slice) from
    every tensor; with two tensors the limit is the elementwise minimum of their shapes."""
    g = None
    if seed is not None:
        g = torch.Generator().manual_seed(int(seed))
    shape = list(tensors[0].shape)
    if len(tensors) == 2:
        shape = [min(a, b) for a, b in zip(tensors[0].shape, tensors[1].shape)]
    offset = [int(torch.randint(0, s - z + 1, (1,), generator=g)) for s, z in zip(shape, size)]
    sl = tuple(slice(o, o + z) for o, z in zip(offset, size))
    return [t[sl] for t in tensors]


def draw_training_augmentation(num_batch, generator=None):
    """All draws of one training step, with the reference's ranges (unsupervised.py:39-58)."""
    aug = dict(theta_global=draw_affine(num_batch, horizontal_flipping=True, min_scale=0.9, max_scale=1.1,
                                        generator=generator),
               theta_local=draw_affine(num_batch, min_scale=0.9, max_scale=1.1, generator=generator))
    aug.update(draw_photometric(num_batch, noise_stddev=0.04, min_contrast=-0.3, max_contrast=0.3,
                                brightness_stddev=0.02, min_colour=0.9, max_colour=1.1, min_gamma=0.7,
                                max_gamma=1.5, generator=generator))
    return aug
