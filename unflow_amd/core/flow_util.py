"""Mirror of the metric part of src/e2eflow/core/flow_util.py:98-123 (the colour-wheel visualisers are out of scope)."""
import torch

from .engine import flow_error_avg  # noqa: F401  (EPE, flow_util.py:98-103)


def euclidean(t):
    return torch.sqrt((t ** 2).sum(3, keepdim=True))


def outlier_ratio(gt_flow, flow, mask, threshold=3.0, relative=0.05):
    """flow_util.py:106-114."""
    diff = euclidean(gt_flow - flow) * mask
    thr = torch.clamp(euclidean(gt_flow) * relative, min=threshold) if relative is not None else threshold
    return (diff >= thr).float().sum() / mask.sum()


def outlier_pct(gt_flow, flow, mask, threshold=3.0, relative=0.05):
    return outlier_ratio(gt_flow, flow, mask, threshold, relative) * 100
