"""Mirror of src/e2eflow/core/flownet.py:14-81: `flownet(im1, im2, flownet_spec, full_resolution, train_all,
backward_flow)` for every spec the reference accepts — 'C' / 'c' first or 'S' / 's' (3/8-width in lower case), followed by
'S' / 's' refinement networks — returning, like the reference, one list of flows [flow2 .. flow6] (NHWC; [flow0, flow1,
flow2 ..] for the last network with full_resolution) per network, forward and (backward_flow=True) backward direction.
The networks run on the static-plan engine (core/engine.py); engines are cached per (shape, device, configuration)."""
import torch

from .engine import FlowNetEngine, FLOW_SCALE  # noqa: F401

_engines = {}
_ENGINE_KEYS = ('flownet', 'full_res', 'train_all', 'pyramid_loss', 'border_mask', 'mask_occlusion')


def _engine_params(params):
    """The engine-relevant, hashable part of a reference-style params dict (which may also carry lists such as finetune or
    manual_decay_iters)."""
    params = params or {}
    return {k: v for k, v in params.items() if (k in _ENGINE_KEYS or k.endswith('_weight')) and v is not None}


def get_engine(batch, height, width, params=None, device=None):
    ep = _engine_params(params)
    key = (batch, height, width, None if device is None else str(torch.device(device)), tuple(sorted(ep.items())))
    if key not in _engines:
        _engines[key] = FlowNetEngine(batch, height, width, params=ep or None, device=device)
    return _engines[key]


def flownet(im1, im2, flownet_spec='S', full_resolution=False, train_all=False, backward_flow=False, engine=None):
    """im1, im2: mean-subtracted images (what unsupervised_loss feeds, unsupervised.py:67-68), NHWC [B,H,W,3]."""
    B, H, W, _ = im1.shape
    eng = engine or get_engine(B, H, W, params=dict(flownet=flownet_spec, full_res=bool(full_resolution),
                                                    train_all=bool(train_all)), device=im1.device)
    assert eng.spec == flownet_spec and eng.full_res == bool(full_resolution)
    x0 = eng.x0
    x0[:B, ..., :3] = im1
    x0[B:, ..., :3] = im2
    eng._input_planes()
    eng.forward_net()
    flows_fw = [[st.act['flow%d' % l][:B] for l in st.flow_levels] for st in eng.stages]
    flows_bw = [[st.act['flow%d' % l][B:] for l in st.flow_levels] for st in eng.stages]
    return (flows_fw, flows_bw) if backward_flow else flows_fw
