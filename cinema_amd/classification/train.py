"""Classification loss / forward of the ConvViT fine-tuning recipe on the HIP path (reference ``cinema/classification/train.py:26-178``).

Not rebuilt (outside the hot path): the sklearn metric tables (``binary_/multiclass_classification_metrics``), the data-loader
evaluation loops and the ResNet baselines."""

from __future__ import annotations

import torch

from cinema_amd import hip as K
from cinema_amd.train import FineTuneStep, patch_average_forward


def get_classification_or_regression_model(config):  # noqa: ANN001, ANN201
    """``config.model.name == "convvit"`` of the reference's builder (``classification/train.py:26-78``); its ResNet baselines are not part of this build."""
    if config.model.name != "convvit":
        raise ValueError(f"Invalid model name {config.model.name}: this build provides the ConvViT path only.")
    from cinema_amd.convvit import get_model

    return get_model(config)


class _HeadCE(torch.autograd.Function):
    """Label-smoothed cross entropy of the head's logits [b, c]: value and gradient from ONE launch (``cinema_head_ce``)."""

    @staticmethod
    def forward(ctx, logits: torch.Tensor, labels: torch.Tensor, label_smoothing: float):  # noqa: ANN001, ANN205
        out, d = K.head_ce(logits.detach().float().contiguous(), labels.reshape(-1).to(torch.int32).contiguous(), label_smoothing)
        ctx.save_for_backward(d)
        ctx.dtype = logits.dtype
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):  # noqa: ANN001, ANN205
        (d,) = ctx.saved_tensors
        return (d * g).to(ctx.dtype), None, None


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor, label_smoothing: float = 0.0) -> torch.Tensor:
    """``F.cross_entropy(logits, labels, label_smoothing=...)`` (mean over the batch) for logits (batch, n_classes) on the device."""
    if logits.dim() != 2 or labels.numel() != logits.shape[0]:
        raise ValueError(f"logits {tuple(logits.shape)} and labels {tuple(labels.shape)} do not match")
    return _HeadCE.apply(logits, labels, label_smoothing)


def classification_loss_tensors(model, batch: dict, views: list, device: torch.device, label_smoothing: float = 0.1) -> tuple:  # noqa: ANN001
    """:func:`classification_loss` without the ``.item()`` read-backs: -> (loss, {"cross_entropy", "loss"} as 0-d device tensors)."""
    image_dict = {v: batch[f"{v}_image"].to(device) for v in views}
    logits = model(image_dict)
    ce = cross_entropy(logits, batch["label"].long().to(device), label_smoothing)
    return ce, {"cross_entropy": ce.detach(), "loss": ce.detach()}


def classification_loss(model, batch: dict, views: list, device: torch.device, label_smoothing: float = 0.1) -> tuple:  # noqa: ANN001
    """Reference ``classification_loss`` (``classification/train.py:80-110``): images ``{view}_image``, integer ``label`` -> (loss, metric floats)."""
    ce, metrics = classification_loss_tensors(model, batch, views, device, label_smoothing)
    return ce, {k: float(v) for k, v in metrics.items()}


def classification_forward(model, image_dict: dict, patch_size_dict: dict, amp_dtype: torch.dtype | None = None) -> torch.Tensor:  # noqa: ANN001, ARG001
    """Reference ``classification_forward`` (``classification/train.py:113-178``): logits (1, n_classes); with one over-sized view the per-patch
    probabilities are averaged and the log is returned.  ``amp_dtype`` is accepted for signature compatibility (the path computes in bf16)."""
    return patch_average_forward(model, image_dict, patch_size_dict,
                                 lambda logits: torch.log(torch.mean(torch.softmax(logits, dim=1), dim=0, keepdim=True)))


class ClsTrainStep(FineTuneStep):
    """Fused fine-tuning step of the classification task (``cinema/train.py:85-168`` with ``classification_loss``)."""

    def __init__(self, model, views: list, label_smoothing: float = 0.1, **kw) -> None:  # noqa: ANN001, ANN003
        super().__init__(model, views, lambda m, b, v, d: classification_loss_tensors(m, b, v, d, label_smoothing), **kw)
