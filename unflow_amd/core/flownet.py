"""Mirror of src/e2eflow/core/flownet.py:14-81 for the single FlowNetC network: `flownet(im1, im2, 'C',
backward_flow=True)` returns (flows_fw, flows_bw), each a one-element list holding [flow2..flow6] (NHWC), exactly
like the reference's return value for a one-network spec.  Stacked specs (CS, CSS) and FlowNetS are §8f "next"."""
from .engine import FlowNetCEngine, FLOW_SCALE  # noqa: F401

_engines = {}


def get_engine(batch, height, width, params=None, device=None):
    key = (batch, height, width, None if device is None else str(device), tuple(sorted((params or {}).items())))
    if key not in _engines:
        _engines[key] = FlowNetCEngine(batch, height, width, params=params, device=device)
    return _engines[key]


def flownet(im1, im2, flownet_spec='S', full_resolution=False, train_all=False, backward_flow=False, engine=None):
    """im1, im2: mean-subtracted images in [-1, 1]-ish (what unsupervised_loss feeds), NHWC."""
    if flownet_spec != 'C' or full_resolution:
        raise NotImplementedError("only flownet_spec='C' without full_res is implemented")
    B, H, W, _ = im1.shape
    eng = engine or get_engine(B, H, W, device=im1.device)
    x0 = eng.act['x0']
    x0[:B, ..., :3] = im1
    x0[B:, ..., :3] = im2
    eng.forward_net()
    fw, bw = eng.flows()
    return ([fw], [bw]) if backward_flow else [fw]
