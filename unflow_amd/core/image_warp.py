"""image_warp — replacement for src/e2eflow/core/image_warp.py:4-76.

Same signature and semantics (bilinear backward warp, x + int(floor(u)) indexing, clamp-to-edge,
gradients to image AND flow exactly as TF autodiff derives them), executed by one HIP gather
kernel (fwd) and one gather/scatter kernel (bwd) instead of ~15 TF ops."""
import torch

from .. import _lib
from .._lib import check, ptr, stream, cf
from ..ops import _dev


def image_warp_indices(im, flow):
    """The 4 flat gather indices (a,b,c,d of image_warp.py:61-66) per pixel, int32 [B,H,W,4]."""
    im, flow = _dev(im, 'im'), _dev(flow, 'flow')
    B, H, W, C = im.shape
    out = torch.empty_like(im)
    idx = torch.empty((B, H, W, 4), dtype=torch.int32, device=im.device)
    check(_lib.lib().unflow_image_warp_fwd(ptr(im), C, ptr(flow), cf(1.0), ptr(out), ptr(idx), 0, B, H, W, C,
                                           stream()), "image_warp")
    return idx


class _ImageWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, flow):
        B, H, W, C = im.shape
        if tuple(flow.shape) != (B, H, W, 2):
            raise ValueError("flow must be [B,H,W,2] matching im")
        ctx.save_for_backward(im, flow)
        out = torch.empty_like(im)
        check(_lib.lib().unflow_image_warp_fwd(ptr(im), C, ptr(flow), cf(1.0), ptr(out), ptr(None), 0, B, H, W, C,
                                               stream()), "image_warp")
        return out

    @staticmethod
    def backward(ctx, g):
        im, flow = ctx.saved_tensors
        B, H, W, C = im.shape
        g = g.contiguous()
        d_im = torch.zeros_like(im) if ctx.needs_input_grad[0] else None
        d_flow = torch.empty_like(flow)
        check(_lib.lib().unflow_image_warp_bwd(ptr(g), ptr(im), C, ptr(flow), cf(1.0), ptr(d_im), ptr(d_flow), 0, 0,
                                               B, H, W, C, stream()), "image_warp_grad")
        return d_im, d_flow


def image_warp(im, flow):
    """Performs a backward warp of an image using the predicted flow (image_warp.py:4-13).
    im: [num_batch, height, width, channels]; flow: [num_batch, height, width, 2]."""
    return _ImageWarp.apply(_dev(im, 'im'), _dev(flow, 'flow'))
