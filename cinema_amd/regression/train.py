"""Regression loss / forward of the ConvViT fine-tuning recipe on the HIP path (reference ``cinema/regression/train.py:21-123``).

Not rebuilt (outside the hot path): the data-loader evaluation loops and their metric tables."""

from __future__ import annotations

import torch

from cinema_amd import hip as K
from cinema_amd.train import FineTuneStep, patch_average_forward

_METRIC_KEYS = ("mse_loss", "mae_loss", "max_label", "min_label", "max_pred", "min_pred")


class _HeadMSE(torch.autograd.Function):
    """``F.mse_loss`` of the head's outputs with the values the reference reports, value and gradient from ONE launch (``cinema_head_mse``)."""

    @staticmethod
    def forward(ctx, preds: torch.Tensor, label: torch.Tensor):  # noqa: ANN001, ANN205
        out, d = K.head_mse(preds.detach().float().contiguous(), label.float().contiguous())
        ctx.save_for_backward(d)
        ctx.dtype = preds.dtype
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g, _g_out):  # noqa: ANN001, ANN205
        (d,) = ctx.saved_tensors
        return (d * g).to(ctx.dtype), None


def regression_loss_tensors(model, batch: dict, views: list, device: torch.device) -> tuple:  # noqa: ANN001
    """:func:`regression_loss` without the ``.item()`` read-backs: -> (mse, metrics as 0-d device tensors, same keys)."""
    image_dict = {v: batch[f"{v}_image"].to(device) for v in views}
    preds = model(image_dict)
    label = batch["label"].to(device=device, dtype=torch.float32)
    if tuple(label.shape) != tuple(preds.shape):
        raise ValueError(f"predictions {tuple(preds.shape)} and labels {tuple(label.shape)} do not match")
    mse, out = _HeadMSE.apply(preds, label)
    metrics = {k: out[i] for i, k in enumerate(_METRIC_KEYS)}
    metrics["loss"] = out[0]
    return mse, metrics


def regression_loss(model, batch: dict, views: list, device: torch.device) -> tuple:  # noqa: ANN001
    """Reference ``regression_loss`` (``regression/train.py:21-56``): images ``{view}_image``, float ``label`` (batch, n) -> (mse, metric floats)."""
    mse, metrics = regression_loss_tensors(model, batch, views, device)
    return mse, {k: float(v) for k, v in metrics.items()}


def regression_forward(model, image_dict: dict, patch_size_dict: dict, amp_dtype: torch.dtype | None = None) -> torch.Tensor:  # noqa: ANN001, ARG001
    """Reference ``regression_forward`` (``regression/train.py:59-123``): predictions (1, n); with one over-sized view the mean over its
    half-overlapping patches."""
    return patch_average_forward(model, image_dict, patch_size_dict, lambda preds: torch.mean(preds, dim=0, keepdim=True))


class RegTrainStep(FineTuneStep):
    """Fused fine-tuning step of the regression task (``cinema/train.py:85-168`` with ``regression_loss``)."""

    def __init__(self, model, views: list, **kw) -> None:  # noqa: ANN001, ANN003
        super().__init__(model, views, regression_loss_tensors, **kw)
