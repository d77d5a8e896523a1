"""Loss of the segmentation fine-tuning step (interface of the reference ``cinema/segmentation/train.py:77-146``)."""

from __future__ import annotations

import torch

from cinema_amd import hip as K


class _SegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: torch.Tensor, labels: torch.Tensor):  # noqa: ANN001, ANN205
        b, c = logits.shape[0], logits.shape[1]
        rows = logits.detach().float().movedim(1, -1).contiguous().reshape(-1, c)  # channels-last rows (layout only)
        lab = labels.reshape(-1).to(torch.int32).contiguous()
        out4, coef = K.seg_loss_fwd(rows, lab, b)
        ctx.save_for_backward(rows, lab, coef, out4)
        ctx.shape, ctx.dtype = tuple(logits.shape), logits.dtype
        return out4[0].clone(), out4[1].clone(), out4[2].clone()

    @staticmethod
    def backward(ctx, g_loss, g_ce, g_dice):  # noqa: ANN001, ANN205
        rows, lab, coef, out4 = ctx.saved_tensors
        b, c = ctx.shape[0], ctx.shape[1]
        if g_ce is not None and bool((g_ce != 0).any()) or g_dice is not None and bool((g_dice != 0).any()):
            raise NotImplementedError("only the total loss is differentiable (the metrics are reported values)")
        up = g_loss.reshape(1).float().contiguous()
        d = K.seg_loss_bwd(rows, lab, b, coef, out4, up)
        return d.reshape(b, *ctx.shape[2:], c).movedim(-1, 1).to(ctx.dtype), None


def _segmentation_loss(logits: torch.Tensor, labels: torch.Tensor) -> tuple:
    """Cross entropy (ignore_index = -1) + soft Dice without background for one view (reference ``train.py:77-103``).

    ``logits`` (batch, n_classes, ...), ``labels`` (batch, 1, ...) integer; returns (loss, {"cross_entropy", "mean_dice_loss", "loss"})."""
    if logits.shape[0] != labels.shape[0] or tuple(logits.shape[2:]) != tuple(labels.shape[2:]) or labels.shape[1] != 1:
        raise ValueError(f"logits {tuple(logits.shape)} and labels {tuple(labels.shape)} do not match")
    loss, ce, dice = _SegLoss.apply(logits, labels)
    return loss, {"cross_entropy": ce.detach(), "mean_dice_loss": dice.detach(), "loss": loss}


def segmentation_loss(model, batch: dict, views: list, device: torch.device, loss_fn=_segmentation_loss) -> tuple:  # noqa: ANN001
    """Mean of the per-view losses and the metric dict of floats, keys as the reference builds them (``train.py:106-146``, including its
    ``{view}_{view}_loss`` entry): images ``{view}_image``, labels ``{view}_label``."""
    image_dict = {v: batch[f"{v}_image"].to(device) for v in views}
    label_dict = {v: batch[f"{v}_label"].to(device) for v in views}
    logits_dict = model(image_dict)
    metrics, losses, metric_keys = {}, [], []
    for v, logits in logits_dict.items():
        loss_v, metrics_v = loss_fn(logits, label_dict[v])
        metric_keys = list(metrics_v.keys())
        metrics_v[f"{v}_loss"] = loss_v
        losses.append(loss_v)
        metrics.update({f"{v}_{k}": val for k, val in metrics_v.items()})
    loss = sum(losses) / len(logits_dict)
    metrics["loss"] = loss
    metrics = {k: float(val) for k, val in metrics.items()}
    for k in metric_keys:
        metrics[k] = sum(metrics[f"{v}_{k}"] for v in logits_dict) / len(logits_dict)
    return loss, metrics
