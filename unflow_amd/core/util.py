"""Mirror of src/e2eflow/core/util.py:12-26."""
import math

import torch

from ..ops import downsample as downsample_ops


def _area_weights(n_in, n_out, device):
    """[n_out, n_in] weights of tf.image.resize_area along one axis: output i averages the source interval
    [i*s, (i+1)*s), s = n_in / n_out, every source pixel weighted by the covered fraction (TF's ResizeAreaOp)."""
    s = n_in / n_out
    w = torch.zeros(n_out, n_in, dtype=torch.float64)
    for i in range(n_out):
        lo, hi = i * s, (i + 1) * s
        for j in range(int(math.floor(lo)), min(n_in, int(math.ceil(hi)))):
            w[i, j] = (min(hi, j + 1) - max(lo, j)) / s
    return w.to(device=device, dtype=torch.float32)


def resize_area(tensor, like):
    """tf.image.resize_area(tensor, like.shape[1:3]) (core/util.py:12-14; stop_gradient there — no gradient here either)."""
    _, h, w, _ = like.shape
    B, H, W, C = tensor.shape
    wy, wx = _area_weights(H, h, tensor.device), _area_weights(W, w, tensor.device)
    return torch.einsum('yh,bhwc,xw->byxc', wy, tensor.detach(), wx)


def downsample(tensor, num):
    """core/util.py:21-26: the downsample op when H and W are even (the reference tests % 2; the op itself requires
    divisibility by num, downsample_op.cc:37-40), else tf.image.resize_area to (int(H / num), int(W / num))."""
    _, height, width, _ = tensor.shape
    if height % 2 == 0 and width % 2 == 0:
        return downsample_ops(tensor, num)
    like = torch.empty(1, int(height / num), int(width / num), 1)
    return resize_area(tensor, like)
