"""Loss of the segmentation fine-tuning step (interface of the reference ``cinema/segmentation/train.py:77-146``)."""

from __future__ import annotations

import torch

from cinema_amd import hip as K
from cinema_amd.train import FineTuneStep


def get_segmentation_model(config):  # noqa: ANN001, ANN201
    """``config.model.name == "convunetr"`` of the reference's builder (``cinema/segmentation/train.py:31-75``); its UNet baseline is not part of this build."""
    if config.model.name != "convunetr":
        raise ValueError(f"Invalid model name {config.model.name}: this build provides the ConvUNetR path only.")
    from cinema_amd.segmentation.convunetr import get_model

    model = get_model(config)
    model.set_grad_ckpt(config.grad_ckpt)
    return model


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


# ---------------------------------------------------------------------------------------------------------------------
# training step on device tensors (no per-metric host read-back) and the fused fine-tuning step
# ---------------------------------------------------------------------------------------------------------------------
def segmentation_loss_tensors(model, batch: dict, views: list, device: torch.device, loss_fn=_segmentation_loss) -> tuple:  # noqa: ANN001
    """:func:`segmentation_loss` without the ``float()`` conversions (one device->host synchronisation per metric in the reference,
    ``train.py:142``): -> (loss, {metric: 0-d device tensor}) with the same keys."""
    image_dict = {v: batch[f"{v}_image"].to(device) for v in views}
    label_dict = {v: batch[f"{v}_label"].to(device) for v in views}
    logits_dict = model(image_dict)
    metrics, losses, metric_keys = {}, [], []
    for v, logits in logits_dict.items():
        loss_v, metrics_v = loss_fn(logits, label_dict[v])
        metric_keys = list(metrics_v.keys())
        metrics_v[f"{v}_loss"] = loss_v
        losses.append(loss_v)
        metrics.update({f"{v}_{k}": val.detach() for k, val in metrics_v.items()})
    loss = sum(losses) / len(logits_dict)
    metrics["loss"] = loss.detach()
    for k in metric_keys:
        metrics[k] = sum(metrics[f"{v}_{k}"] for v in logits_dict) / len(logits_dict)
    return loss, metrics


class SegTrainStep(FineTuneStep):
    """One optimisation step of the segmentation fine-tuning loop (reference ``cinema/train.py:85-168`` with ``segmentation_loss`` as
    ``loss_fn``): forward -> CE + Dice -> backward into the flat gradient buffer -> (data-parallel mean all-reduce) -> global-norm clip ->
    fused AdamW over the layer-decay parameter groups (``param_groups_lr_decay``, ``cinema/train.py:262-270``; ``layer_decay=None``: one group,
    pass ``weight_decay=0.01`` for torch's default as the reference's from-scratch branch uses it) -> zero_grad.  Returns (loss, grad_norm | None,
    metrics) as device tensors."""

    def __init__(self, model, views: list, lr: float = 1e-3, betas: tuple = (0.9, 0.95), weight_decay: float = 0.05, layer_decay: float | None = 0.75,  # noqa: ANN001
                 clip_grad: float | None = 5.0, synchronizer=None, replay: bool = False, audit: bool = False, check_every: int = 100) -> None:  # noqa: ANN001
        super().__init__(model, views, segmentation_loss_tensors, lr=lr, betas=betas, weight_decay=weight_decay, layer_decay=layer_decay,
                         clip_grad=clip_grad, synchronizer=synchronizer, check_every=check_every)
        # replay: forward + loss + backward recorded once per input signature as the flat list of this library's launches and re-issued from it
        # (cinema_amd/replay.py RecordedSegStep): the module code needs ~50 ms of host time for the ~2000 launches of config 4, as long as the GPU needs
        self.replay, self.audit = replay, audit
        self._recorded: dict = {}

    def reset_recordings(self) -> None:
        """Drop the recorded steps (after changing ``requires_grad`` flags, train / eval mode or sub-modules)."""
        self._recorded.clear()

    def __call__(self, batch: dict, n_accum_steps: int = 1, update_grad: bool = True) -> tuple:
        from cinema_amd import tape as T  # noqa: N812

        with T.side_streams_limit(1):  # one weight-gradient stream for this model (measured: the second one costs 0.35 ms here)
            return self._call(batch, n_accum_steps, update_grad)

    def _call(self, batch: dict, n_accum_steps: int, update_grad: bool) -> tuple:
        if not (self.replay and n_accum_steps == 1 and self.model.training):
            return super().__call__(batch, n_accum_steps, update_grad)
        from cinema_amd.replay import RecordedSegStep

        if self.sync is not None:
            self.sync.arm(update_grad)
        key = tuple((v, tuple(batch[f"{v}_image"].shape), tuple(batch[f"{v}_label"].shape)) for v in self.views)
        rec = self._recorded.get(key)
        if rec is None:  # the recording IS this step
            rec = self._recorded[key] = RecordedSegStep(self.model, self.views, batch, audit=self.audit)
            loss, metrics = rec.loss, rec.metrics
        else:
            loss, metrics = rec.run(batch)
        grad_norm = None
        if update_grad:
            if self.sync is not None:
                self.sync.all_reduce()
            grad_norm = self.optimizer.step(self.clip_grad)
            self.optimizer.zero_grad()
            self._updated()
        return loss, grad_norm, metrics


# ---------------------------------------------------------------------------------------------------------------------
# evaluation path (reference cinema/segmentation/train.py:148-368)
# ---------------------------------------------------------------------------------------------------------------------
def segmentation_forward(model, image_dict: dict, patch_size_dict: dict, amp_dtype: torch.dtype | None = None, window_batch: int = 8) -> dict:  # noqa: ANN001, ARG001
    """Logits per view, (1, n_classes, *image_size), with a sliding window over the ONE view whose image is larger than its patch size
    (reference ``train.py:148-221``: windows overlap by half a patch, ``get_patch_grid``; class probabilities are averaged over the windows
    covering a voxel and the log is returned; the other views' outputs are the log of their window-mean probability).

    MI355X path: the windows go through the model ``window_batch`` at a time (the reference runs them one by one; samples are independent and
    the model is in eval mode, so the results are the same), each window's softmax is added into the probability volume by
    ``cinema_seg_window_accumulate`` in grid order, ``cinema_seg_window_finish`` takes the log.  ``amp_dtype`` is accepted for signature
    compatibility (the HIP path always computes bf16-MFMA / fp32-accumulate)."""
    from cinema_amd import hip as K
    from cinema_amd.transform import get_patch_grid, patch_grid_sample

    for view, image in image_dict.items():
        if any(s < p for s, p in zip(image.shape[2:], patch_size_dict[view])):
            raise ValueError(f"For view {view}, image size {image.shape[2:]} is smaller than patch size {patch_size_dict[view]}.")
    views = list(image_dict.keys())
    need_patch = {v: tuple(image_dict[v].shape[2:]) != tuple(patch_size_dict[v]) for v in views}
    if not any(need_patch.values()):
        with torch.no_grad():
            return model(image_dict)
    if sum(need_patch.values()) > 1:
        raise ValueError(f"Only support patching on one view for now, but got {need_patch}.")
    batch_size = image_dict[views[0]].shape[0]
    if batch_size != 1:
        raise ValueError(f"Expected batch size 1 for patching, but got {batch_size}.")

    view_p = next(v for v, need in need_patch.items() if need)
    image = image_dict[view_p][0]  # (channel, *image_size)
    size = tuple(image.shape[1:])
    patch = tuple(patch_size_dict[view_p])
    starts = get_patch_grid(image_size=size, patch_size=patch, patch_overlap=tuple(s // 2 for s in patch))
    patches = patch_grid_sample(image, starts, patch)  # (n_patches, channel, *patch)
    n_patches = patches.shape[0]
    dev = image.device
    n_vox = 1
    for s in size:
        n_vox *= int(s)

    prob_sum, count, other_sum = None, None, {}
    with torch.no_grad():
        for i0 in range(0, n_patches, window_batch):
            chunk = patches[i0:i0 + window_batch]
            nb = chunk.shape[0]
            feed = {v: chunk if v == view_p else image_dict[v].expand(nb, *image_dict[v].shape[1:]) for v in views}
            out = model(feed)
            for v in views:
                lg = out[v].float()
                c = lg.shape[1]
                if v == view_p:
                    if prob_sum is None:
                        prob_sum = K.zeros((n_vox, c), torch.float32, dev)
                        count = K.zeros((n_vox,), torch.float32, dev)
                    rows = lg.movedim(1, -1).contiguous().reshape(nb, -1, c)  # channels-last rows per window (layout only)
                    for j in range(nb):
                        K.seg_window_accumulate(rows[j], patch, tuple(int(s) for s in starts[i0 + j]), size, prob_sum, count)
                else:  # the same full image in every window: one volume covering everything, n_patches times
                    o_size = tuple(lg.shape[2:])
                    if v not in other_sum:
                        nv = lg[0, 0].numel()
                        other_sum[v] = (K.zeros((nv, c), torch.float32, dev), K.zeros((nv,), torch.float32, dev), o_size)
                    rows = lg.movedim(1, -1).contiguous().reshape(nb, -1, c)
                    for j in range(nb):
                        K.seg_window_accumulate(rows[j], o_size, (0,) * len(o_size), o_size, other_sum[v][0], other_sum[v][1])
    result = {}
    for v in views:
        if v == view_p:
            result[v] = K.seg_window_finish(prob_sum, count).reshape(1, -1, *size)
        else:
            ps, cn, o_size = other_sum[v]
            result[v] = K.seg_window_finish(ps, cn).reshape(1, -1, *o_size)
    return result


def _dice_iou_from_counts(pred: torch.Tensor, true: torch.Tensor, inter: torch.Tensor) -> tuple:
    """monai 1.5.2 ``compute_dice`` / ``compute_iou`` with their defaults (``ignore_empty=True``): NaN where the ground truth is empty,
    else 2 I / (T + P) and I / (T + P - I)."""
    nan = torch.full_like(true, float("nan"))
    dice = torch.where(true > 0, 2.0 * inter / (true + pred).clamp_min(1e-30), nan)
    iou = torch.where(true > 0, inter / (true + pred - inter).clamp_min(1e-30), nan)
    return dice, iou


def segmentation_metrics(logits: torch.Tensor, labels: torch.Tensor, spacing: tuple) -> dict:
    """Per-sample evaluation metrics (reference ``train.py:224-286``): Dice, IoU, stability score and volumes per foreground class and their
    means, each of shape (batch,).  All voxel counting (argmax prediction, label, intersections, the two stability masks) is ONE pass of
    ``cinema_seg_metric_counts`` over the channels-first logits; the few (batch, n_classes) ratios are formed from those integer counts.
    ``hausdorff_distance_95`` (monai ``compute_hausdorff_distance``, reference ``train.py:262-267``) comes from :func:`cinema_amd.metric.hausdorff_distance_95`
    (surface extraction + nearest-surface distances on the device; monai's algorithm restated, parity-unpinned)."""
    from cinema_amd import hip as K
    from cinema_amd.metric import hausdorff_distance_95

    n_classes = logits.shape[1] - 1
    lab = labels.squeeze(dim=1).to(torch.int32).contiguous()
    counts = K.seg_metric_counts(logits.float().contiguous(), lab).to(torch.float32)  # (batch, 1 + n_classes, 6)
    pred, true, inter, hi, lo, both = (counts[..., i] for i in range(6))
    dice, iou = _dice_iou_from_counts(pred, true, inter)
    _, stability = _dice_iou_from_counts(hi, lo, both)  # compute_iou(y_pred = high mask, y = low mask), cinema/metric.py:42
    vol = 1.0
    for s in spacing:
        vol *= float(s)
    true_vol, pred_vol = true * vol / 1000.0, pred * vol / 1000.0  # ml (cinema/metric.py:84-96)
    hd95 = hausdorff_distance_95(logits.argmax(dim=1), lab, n_classes, spacing)  # (batch, n_classes), foreground classes
    metrics = {}
    for i in range(n_classes):
        k = i + 1
        metrics[f"class_{k}_dice_score"] = dice[:, k]
        metrics[f"class_{k}_iou_score"] = iou[:, k]
        metrics[f"class_{k}_stability_score"] = stability[:, k]
        metrics[f"class_{k}_hausdorff_distance_95"] = hd95[:, i]
        metrics[f"class_{k}_true_volume"] = true_vol[:, k]
        metrics[f"class_{k}_pred_volume"] = pred_vol[:, k]
    metrics["mean_dice_score"] = torch.mean(dice[:, 1:], dim=-1)
    metrics["mean_iou_score"] = torch.mean(iou[:, 1:], dim=-1)
    metrics["mean_stability_score"] = torch.mean(stability[:, 1:], dim=-1)
    metrics["mean_hausdorff_distance_95"] = torch.mean(hd95, dim=-1)
    return metrics


def segmentation_eval(model, batch: dict, patch_size_dict: dict, spacing_dict: dict, amp_dtype: torch.dtype | None, device: torch.device,  # noqa: ANN001
                      metrics_fn=segmentation_metrics) -> tuple:  # noqa: ANN001
    """(logits per view cropped to the un-padded size, {metric: float}) for one sample (reference ``train.py:288-355``)."""
    import numpy as np

    from cinema_amd.transform import crop_start

    views = list(patch_size_dict.keys())
    image_dict = {v: batch[f"{v}_image"].to(device) for v in views}
    logits_dict = segmentation_forward(model, image_dict, patch_size_dict, amp_dtype)

    def crop(t: torch.Tensor, v: str) -> torch.Tensor:
        width, height = int(batch[f"{v}_width"][0]), int(batch[f"{v}_height"][0])
        if len(patch_size_dict[v]) == 3:
            return crop_start(t, (*t.shape[:2], width, height, int(batch["n_slices"][0])))
        if len(patch_size_dict[v]) == 2:
            return crop_start(t, (*t.shape[:2], width, height))
        raise ValueError(f"Invalid patch size {patch_size_dict[v]}.")

    for v in views:
        logits_dict[v] = crop(logits_dict[v], v)
    if metrics_fn is None:
        return logits_dict, {}
    metrics, metric_keys = {}, []
    for v in views:
        metrics_v = metrics_fn(logits_dict[v], crop(batch[f"{v}_label"].to(device), v), spacing_dict[v])
        metric_keys = list(metrics_v.keys())
        for k, val in metrics_v.items():
            metrics[f"{v}_{k}"] = float(val.detach().to(dtype=torch.float32).reshape(-1)[0])  # batch size 1 (train.py:350)
    for k in metric_keys:
        metrics[k] = float(np.mean([metrics[f"{v}_{k}"] for v in views]))
    return logits_dict, metrics


def segmentation_eval_dataloader(model, dataloader, patch_size_dict: dict, spacing_dict: dict, amp_dtype: torch.dtype | None, device: torch.device,  # noqa: ANN001
                                 metrics_fn=segmentation_metrics) -> dict:  # noqa: ANN001
    """NaN-mean of the per-sample metrics over a batch-size-1 loader (reference ``train.py:358-397``)."""
    import numpy as np
    from collections import defaultdict

    metrics = defaultdict(list)
    for batch in dataloader:
        _, sample = segmentation_eval(model, batch, patch_size_dict, spacing_dict, amp_dtype, device, metrics_fn)
        for k, v in sample.items():
            metrics[k].append(v)
    return {k: float(np.nanmean(v)) for k, v in metrics.items()}
