"""The small metric helpers the reference's evaluation / inference scripts import from ``cinema.metric`` (``cinema/metric.py:14-146``).

``stability_score`` and ``get_volumes`` of device tensors count voxels with ``cinema_seg_metric_counts`` (one pass over the channels-first logits, the same
kernel :func:`cinema_amd.segmentation.train.segmentation_metrics` uses); the scalar formulas (ejection fraction, its region, coefficient of variance) are
host arithmetic on whatever the caller passes (floats, numpy arrays, tensors), as in the reference.  Not provided: the landmark heat-map helpers
(``heatmap_argmax``, ``heatmap_soft_argmax``) - the landmark task is outside this build."""

from __future__ import annotations

import numpy as np
import torch

# <= 40 %: reduced EF, > 55 %: normal EF, in between: borderline EF (cinema/metric.py:14-16)
REDUCED_EF = 40
NORMAL_EF = 55


def _counts(logits: torch.Tensor, labels: torch.Tensor | None = None) -> torch.Tensor:
    from cinema_amd import hip as K

    if labels is None:
        labels = torch.zeros((logits.shape[0], *logits.shape[2:]), dtype=torch.int32, device=logits.device)
    return K.seg_metric_counts(logits.float().contiguous(), labels.to(torch.int32).contiguous()).to(torch.float32)  # (batch, n_classes, 6)


def stability_score(logits: torch.Tensor, threshold: float = 0.0, threshold_offset: float = 1.0) -> torch.Tensor:
    """IoU of the masks ``logits - mean_c(logits) >= threshold +/- threshold_offset`` per (sample, class), NaN where the low-threshold mask is empty
    (reference ``stability_score``, ``cinema/metric.py:19-42``, through monai ``compute_iou``).  The kernel implements the reference's defaults
    (threshold 0, offset 1), which is how every call site uses it."""
    if threshold != 0.0 or threshold_offset != 1.0:
        raise NotImplementedError("cinema_amd stability_score: the reference's defaults (threshold 0, offset 1) only")
    c = _counts(logits)
    hi, lo, both = c[..., 3], c[..., 4], c[..., 5]
    return torch.where(lo > 0, both / (hi + lo - both).clamp_min(1e-30), torch.full_like(lo, float("nan")))


def get_volumes(mask: torch.Tensor, spacing: tuple) -> torch.Tensor:
    """Volume in ml of every class of a one-hot mask (batch, n_classes, ...) (reference ``cinema/metric.py:84-96``).  Device tensors are counted by the
    voxel-count kernel (the mask's argmax is the class of a one-hot voxel); host tensors by a plain sum."""
    vol = float(np.prod([float(s) for s in spacing])) / 1000.0
    if mask.is_cuda:
        return _counts(mask)[..., 0] * vol  # column 0: voxels whose argmax is the class
    return mask.sum(dim=tuple(range(2, mask.ndim))) * vol


def ejection_fraction(edv, esv):  # noqa: ANN001, ANN201
    """(EDV - ESV) / EDV x 100 (reference ``cinema/metric.py:99-112``)."""
    return (edv - esv) / edv * 100.0


def coefficient_of_variance(x: np.ndarray, y: np.ndarray) -> float:
    """Coefficient of variance of two measurements (reference ``cinema/metric.py:115-130``)."""
    s2 = (x - y) ** 2 / 2
    m = (x + y) / 2
    return float(np.sqrt(np.mean(s2 / m**2)))


def get_ef_region(x: float) -> int:
    """0 reduced (<= 40), 1 borderline (<= 55), 2 normal (reference ``cinema/metric.py:133-146``)."""
    if x <= REDUCED_EF:
        return 0
    if x <= NORMAL_EF:
        return 1
    return 2
