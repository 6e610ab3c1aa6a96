"""Mirror of src/e2eflow/core/losses.py for the terms of the default [train] configuration
(config_template/config.ini:114,123: ternary_weight, smooth_2nd_weight) plus the mask constructors.
Values are computed by the fused HIP kernels of csrc/loss.hip; gradients of these terms inside the training step
are produced by the engine (core/engine.py), not by autograd."""
import math

import torch

from .. import _lib
from .._lib import check, ptr, stream, cf, cl
from ..ops import _dev

DISOCC_THRESH = 0.8  # losses.py:9


def create_mask(tensor, paddings):
    """losses.py:325-335; paddings = [[top, bottom], [left, right]]."""
    B, H, W, _ = tensor.shape
    m = torch.zeros(B, H, W, 1, dtype=torch.float32, device=tensor.device)
    m[:, paddings[0][0]:H - paddings[0][1], paddings[1][0]:W - paddings[1][1]] = 1.0
    return m


def create_border_mask(tensor, border_ratio=0.1):
    """losses.py:338-344."""
    _, H, W, _ = tensor.shape
    sz = int(math.ceil(min(H, W) * border_ratio))
    return create_mask(tensor, [[sz, sz], [sz, sz]])


def create_outgoing_mask(flow):
    """losses.py:347-366."""
    B, H, W, _ = flow.shape
    gx = torch.arange(W, dtype=torch.float32, device=flow.device).view(1, 1, W)
    gy = torch.arange(H, dtype=torch.float32, device=flow.device).view(1, H, 1)
    px, py = gx + flow[..., 0], gy + flow[..., 1]
    inside = (px <= W - 1) & (px >= 0) & (py <= H - 1) & (py >= 0)
    return inside.float().unsqueeze(3)


def ternary_loss(im1, im2_warped, mask, max_distance=1):
    """losses.py:90-122 (census / ternary), value only."""
    im1, im2w, mask = _dev(im1, 'im1'), _dev(im2_warped, 'im2_warped'), _dev(mask, 'mask')
    B, H, W, _ = im1.shape
    lib = _lib.lib()
    g1 = torch.empty(B, H, W, device=im1.device)
    g2 = torch.empty_like(g1)
    dist = torch.empty_like(g1)
    acc = torch.zeros(1, device=im1.device)
    st = stream()
    check(lib.unflow_rgb_to_gray255(ptr(im1), 3, ptr(g1), cl(B * H * W), st))
    check(lib.unflow_rgb_to_gray255(ptr(im2w), 3, ptr(g2), cl(B * H * W), st))
    m = mask.reshape(-1, H, W).contiguous()
    check(lib.unflow_ternary_fwd(ptr(g1), ptr(g2), ptr(m), m.shape[0], ptr(dist), ptr(acc), cf(1.0), cf(B * H * W),
                                 int(max_distance), B, H, W, st), "ternary_loss")
    return acc[0]


def second_order_loss(flow):
    """losses.py:290-295, value only."""
    flow = _dev(flow, 'flow')
    B, H, W, _ = flow.shape
    acc = torch.zeros(1, device=flow.device)
    check(_lib.lib().unflow_second_order_fwd_bwd(ptr(flow), cf(1.0), ptr(acc), ptr(None), 0, cf(1.0),
                                                 cf(B * H * W * 4), B, H, W, stream()), "second_order_loss")
    return acc[0]


def occlusion(flow_fw, flow_bw):
    """losses.py:125-134: forward-backward occlusion maps (1 = occluded) of a flow pair, NHWC.  Note the reference's
    magnitude term here is |fw|^2 + |bw|^2 of the UNWARPED fields (compute_losses :43-46 uses the warped partner)."""
    from .image_warp import image_warp
    lsq = lambda t: (t * t).sum(3, keepdim=True)       # noqa: E731
    mag_sq = lsq(flow_fw) + lsq(flow_bw)
    flow_diff_fw = flow_fw + image_warp(flow_bw, flow_fw)
    flow_diff_bw = flow_bw + image_warp(flow_fw, flow_bw)
    occ_thresh = 0.01 * mag_sq + 0.5
    return (lsq(flow_diff_fw) > occ_thresh).float(), (lsq(flow_diff_bw) > occ_thresh).float()
