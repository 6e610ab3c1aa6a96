"""Mirror of src/e2eflow/core/losses.py: every callable of the reference module that the training / evaluation path uses
— compute_losses (:16-87), ternary_loss (:90-122), occlusion (:125-134), photometric_loss (:198-199), gradient_loss
(:241-247), smoothness_loss (:250-255), second_order_loss (:290-295), charbonnier_loss (:298-322), length_sq (:12-13) and the
mask constructors (:325-366) — same names, arguments and defaults.  Values are computed by the HIP kernels of csrc/loss.hip
through the C ABI (thin callers: no torch arithmetic on the image-sized tensors); gradients of these terms inside the
training step are produced by the engine (core/engine.py), not by autograd.  divergence / norm / diffusion_loss (:148-195)
are dead code in the reference (no caller) and are not mirrored."""
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


def length_sq(x):
    """losses.py:12-13: sum of squares over the channel axis, keepdims."""
    x = _dev(x, 'x')
    B, H, W, C = x.shape
    out = torch.empty(B, H, W, 1, dtype=torch.float32, device=x.device)
    check(_lib.lib().unflow_length_sq(ptr(x), ptr(out), cl(B * H * W), C, stream()), "length_sq")
    return out


def charbonnier_loss(x, mask=None, truncate=None, alpha=0.45, beta=1.0, epsilon=0.001):
    """losses.py:298-322: sum(min(mask * ((x*beta)^2 + epsilon^2)^alpha, truncate)) / numel(x).  mask: [B,H,W,1] or
    [B,H,W,C] (a [1,H,W,.] mask is broadcast over the batch like TF does)."""
    x = _dev(x, 'x')
    B, H, W, C = x.shape
    m, mc = None, 1
    if mask is not None:
        m = _dev(mask, 'mask')
        mc = m.shape[3]
        if mc not in (1, C):
            raise ValueError("mask channels must be 1 or the channels of x (losses.py:305-307)")
        if m.shape[0] != B:
            m = m.expand(B, H, W, mc).contiguous()
    acc = torch.zeros(1, device=x.device)
    check(_lib.lib().unflow_charbonnier_loss(ptr(x), ptr(m), mc, cf(-1.0 if truncate is None else truncate), cf(alpha), cf(beta),
                                             cf(epsilon), ptr(acc), cf(1.0), cl(B * H * W), C, stream()), "charbonnier_loss")
    return acc[0]


def photometric_loss(im_diff, mask):
    """losses.py:198-199."""
    return charbonnier_loss(im_diff, mask, beta=255)


def smoothness_loss(flow):
    """losses.py:250-255 (first-order), value only."""
    flow = _dev(flow, 'flow')
    B, H, W, _ = flow.shape
    acc = torch.zeros(1, device=flow.device)
    check(_lib.lib().unflow_smooth_1st_fwd_bwd(ptr(flow), cf(1.0), ptr(acc), ptr(None), 0, cf(1.0), cf(B * H * W * 2), B, H, W,
                                               stream()), "smoothness_loss")
    return acc[0]


def _mask3(mask, B, H, W):
    m = _dev(mask, 'mask').reshape(-1, H, W)
    return m, m.shape[0]


def gradient_loss(im1, im2_warped, mask):
    """losses.py:241-247 (Sobel-gradient constancy), value only."""
    im1, im2w = _dev(im1, 'im1'), _dev(im2_warped, 'im2_warped')
    B, H, W, _ = im1.shape
    m, n_mask = _mask3(mask, B, H, W)
    gdiff = torch.empty(B, H, W, 6, device=im1.device)
    acc = torch.zeros(1, device=im1.device)
    check(_lib.lib().unflow_gradient_loss_fwd(ptr(im1), 3, ptr(im2w), ptr(m), n_mask, ptr(gdiff), ptr(acc), cf(1.0),
                                              cf(B * H * W * 6), B, H, W, stream()), "gradient_loss")
    return acc[0]


def compute_losses(im1, im2, flow_fw, flow_bw, border_mask=None, mask_occlusion='', data_max_distance=1):
    """losses.py:16-87: the eight loss values {'sym', 'occ', 'photo', 'grad', 'smooth_1st', 'smooth_2nd', 'fb', 'ternary'} of
    an image pair and its two flows (NHWC, images in [0,1]), each the sum over both directions.  Both directions ride in one
    directed batch of 2B samples through the same kernels the training step uses (unflow_mask_terms builds the masks: border
    or outgoing, 'fb' / 'disocc' occlusion).  Values only; the step's gradients come from the engine."""
    from ..ops import forward_warp
    if mask_occlusion not in ('', 'fb', 'disocc'):
        raise ValueError("mask_occlusion must be one of 'fb', 'disocc', ''")
    im1, im2, flow_fw, flow_bw = _dev(im1, 'im1'), _dev(im2, 'im2'), _dev(flow_fw, 'flow_fw'), _dev(flow_bw, 'flow_bw')
    B, H, W, _ = im1.shape
    N = 2 * B
    dev = im1.device
    lib = _lib.lib()
    st = stream()
    im = torch.cat([im1, im2], 0)
    flow = torch.cat([flow_fw, flow_bw], 0)
    z = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)     # noqa: E731
    acc = lambda: torch.zeros(1, dtype=torch.float32, device=dev)               # noqa: E731
    n1 = B * H * W
    # partner fields: image_warp(flow_other, flow_own) (:38-39), forward_warp of both flows (:28-29)
    warped = z(N, H, W, 2)
    check(lib.unflow_image_warp_fwd(ptr(flow), 2, ptr(flow), cf(1.0), ptr(warped), ptr(None), B, N, H, W, 2, st), "image_warp(flow)")
    fwm = forward_warp(flow).reshape(N, H, W).contiguous()
    base, n_base = (None, 1) if border_mask is None else _mask3(border_mask, B, H, W)
    occl = {'': 0, 'fb': 1, 'disocc': 2}[mask_occlusion]
    mask = z(N, H, W)
    out = {}
    for name, wts in (('fb', (1.0, 0.0, 0.0)), ('occ', (0.0, 1.0, 0.0)), ('sym', (0.0, 0.0, 1.0))):
        a = acc()
        check(lib.unflow_mask_terms(ptr(flow), ptr(warped), ptr(fwm), ptr(base), n_base, cf(1.0), occl, ptr(mask), ptr(a), ptr(None),
                                    ptr(None), 0, cf(wts[0]), cf(wts[1]), cf(wts[2]), B, B, N, H, W, st), "mask_terms")
        out[name] = a[0]
    a = acc()
    check(lib.unflow_photometric_fwd_bwd(ptr(im), 3, ptr(flow), cf(1.0), ptr(mask), N, ptr(a), ptr(None), 0, cf(1.0), cf(n1 * 3), B, N,
                                         H, W, st), "photometric")
    out['photo'] = a[0]
    imw = z(N, H, W, 3)
    check(lib.unflow_image_warp_fwd(ptr(im), 3, ptr(flow), cf(1.0), ptr(imw), ptr(None), B, N, H, W, 3, st), "image_warp(im)")
    a, gdiff = acc(), z(N, H, W, 6)
    check(lib.unflow_gradient_loss_fwd(ptr(im), 3, ptr(imw), ptr(mask), N, ptr(gdiff), ptr(a), cf(1.0), cf(n1 * 6), N, H, W, st),
          "gradient_loss")
    out['grad'] = a[0]
    a = acc()
    check(lib.unflow_smooth_1st_fwd_bwd(ptr(flow), cf(1.0), ptr(a), ptr(None), 0, cf(1.0), cf(n1 * 2), N, H, W, st), "smooth_1st")
    out['smooth_1st'] = a[0]
    a = acc()
    check(lib.unflow_second_order_fwd_bwd(ptr(flow), cf(1.0), ptr(a), ptr(None), 0, cf(1.0), cf(n1 * 4), N, H, W, st), "smooth_2nd")
    out['smooth_2nd'] = a[0]
    a, g1, g2, dist = acc(), z(N, H, W), z(N, H, W), z(N, H, W)
    check(lib.unflow_gray_pair(ptr(im), 3, ptr(flow), cf(1.0), ptr(g1), ptr(g2), B, N, H, W, st), "gray_pair")
    check(lib.unflow_ternary_fwd(ptr(g1), ptr(g2), ptr(mask), N, ptr(dist), ptr(a), cf(1.0), cf(n1), int(data_max_distance), N, H, W,
                                 st), "ternary")
    out['ternary'] = a[0]
    return out
