"""The optimisation step of the reference's generic fine-tuning loop (``cinema/train.py:85-168``, ``train_one_epoch``) for any model of this
build and any ``loss_fn(model, batch, views, device) -> (loss, metrics)``: forward -> loss -> backward into the flat gradient buffer ->
(data-parallel mean all-reduce) -> global-norm clip -> fused AdamW over the layer-decay parameter groups -> zero_grad.

``train_one_epoch`` is the reference's epoch loop around that step.  The hydra / wandb / data-set shell (``run_train``, ``cinema/train.py:171-351``) is the
reference's control plane and is not rebuilt."""

from __future__ import annotations

from typing import Callable

import torch


class FineTuneStep:
    """``step = FineTuneStep(model, views, loss_fn, lr=..., layer_decay=0.75)``; ``loss, grad_norm, metrics = step(batch)``.

    ``layer_decay`` not None: the parameter groups of ``param_groups_lr_decay`` (fine-tuning from a pre-trained checkpoint,
    ``cinema/train.py:262-268``); None: one group of all trainable parameters (``train.py:269-270``).  ``loss_fn`` returns the loss as a tensor that is
    differentiable through the model (the ``*_loss_tensors`` functions of this build keep the metrics on the device; the reference-shaped
    ``*_loss`` functions, which return floats, work as well).  Gradient accumulation as in the reference: ``loss / n_accum_steps`` is
    back-propagated every call, the update happens when ``update_grad``.

    Non-finite losses - a stated difference from the reference: ``train_one_epoch`` / ``pretrain_one_epoch`` there read the loss back and skip only the
    micro-batch whose loss is NaN (``cinema/train.py:138-140``, ``cinema/mae/pretrain.py:255-257``), keeping the gradients accumulated from the
    earlier micro-batches of the window.  Here nothing is read back: a NaN loss back-propagates NaN into the shared flat gradient buffer, the clip kernel
    then zeroes the update and AdamW leaves parameters, moments and the step count untouched - so with ``n_accum_steps > 1`` ONE bad micro-batch drops the
    whole accumulation window, good micro-batches included.  With ``n_accum_steps == 1`` the two behaviours coincide."""

    def __init__(self, model, views: list, loss_fn: Callable, lr: float = 1e-3, betas: tuple = (0.9, 0.95), weight_decay: float = 0.05,  # noqa: ANN001
                 layer_decay: float | None = 0.75, clip_grad: float | None = 5.0, synchronizer=None, check_every: int = 100) -> None:  # noqa: ANN001
        from cinema_amd.convvit import param_groups_lr_decay
        from cinema_amd.optim import FlatModel, FusedAdamW

        self.model, self.views, self.loss_fn, self.clip_grad = model, list(views), loss_fn, clip_grad
        if layer_decay is not None:
            groups = param_groups_lr_decay(model, no_weight_decay_list=[], weight_decay=weight_decay, layer_decay=layer_decay)
        else:
            groups = [{"params": [p for p in model.parameters() if p.requires_grad], "weight_decay": weight_decay}]
        self.flat = FlatModel(model, weight_decay, param_groups=groups)
        self.optimizer = FusedAdamW(self.flat, lr=lr, betas=betas, synchronizer=synchronizer)
        self.sync = self.optimizer.synchronizer
        self.device = self.flat.flat_param.device
        self.check_every, self._n_updates = int(check_every), 0  # error words of the in-launch split reductions, see cinema_amd.optim.TrainStep

    def _updated(self) -> None:
        self._n_updates += 1
        if self.check_every > 0 and self._n_updates % self.check_every == 0:
            from cinema_amd.optim import _check_reductions_on_every_rank

            _check_reductions_on_every_rank(self.sync)  # (collective under data parallelism: every rank raises together)

    def __call__(self, batch: dict, n_accum_steps: int = 1, update_grad: bool = True) -> tuple:
        loss, metrics = self.loss_fn(self.model, batch, self.views, self.device)
        if self.sync is not None:
            self.sync.arm(update_grad)
        (loss / n_accum_steps if n_accum_steps > 1 else loss).backward()
        grad_norm = None
        if update_grad:
            if self.sync is not None:
                self.sync.all_reduce()
            grad_norm = self.optimizer.step(self.clip_grad)
            self.optimizer.zero_grad()
            self._updated()
        return loss.detach(), grad_norm, metrics


def train_one_epoch(step: FineTuneStep, train_dataloader, epoch: int, n_accum_steps: int, n_samples: int, config, log: Callable | None = None) -> int:  # noqa: ANN001
    """One fine-tuning epoch around the fused step (reference ``train_one_epoch``, ``cinema/train.py:85-168``): ``model.train()``, the learning rate of every
    iteration from ``adjust_learning_rate`` at the fractional epoch ``i / len(loader) + epoch`` (scaled per layer-decay group), ``loss / n_accum_steps``,
    the update on ``(i + 1) % n_accum_steps == 0``, ``n_samples += config.train.batch_size_per_device``; ``log(dict)`` on update iterations with the
    ``train_``-prefixed metrics, ``grad_norm``, ``lr``, ``n_samples``, ``epoch``.  The non-finite guard is the optimiser's device-side one (the reference
    reads the loss back every iteration)."""
    from cinema_amd.optim import adjust_learning_rate

    step.model.train()
    tr = config.train
    n_iter = len(train_dataloader)
    for i, batch in enumerate(train_dataloader):
        lr = adjust_learning_rate(optimizer=step.optimizer, step=i / n_iter + epoch, warmup_steps=tr.n_warmup_epochs, max_n_steps=tr.n_epochs, lr=tr.lr,
                                  min_lr=tr.min_lr)
        update_grad = (i + 1) % n_accum_steps == 0
        _, grad_norm, metrics = step(batch, n_accum_steps=n_accum_steps, update_grad=update_grad)
        n_samples += tr.batch_size_per_device
        if update_grad and log is not None:
            out = {f"train_{k}": v for k, v in metrics.items()}
            out.update({"grad_norm": grad_norm, "lr": lr, "n_samples": n_samples, "epoch": epoch})
            log(out)
    return n_samples


def patch_average_forward(model, image_dict: dict, patch_size_dict: dict, combine: Callable) -> torch.Tensor:  # noqa: ANN001
    """Shared body of ``classification_forward`` / ``regression_forward`` (reference ``classification/train.py:113-178``, ``regression/train.py:59-123``):
    the plain forward when every view already has its patch size; otherwise ONE view (batch 1) is cut into half-overlapping patches
    (``get_patch_grid`` / ``patch_grid_sample``), the model runs once per patch with the other views unchanged, and ``combine`` folds the
    per-patch outputs [n_patches, n] into [1, n]."""
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
    view_to_patch = next(v for v, need in need_patch.items() if need)
    image = image_dict[view_to_patch][0]
    patch_size = tuple(patch_size_dict[view_to_patch])
    starts = get_patch_grid(image_size=tuple(image.shape[1:]), patch_size=patch_size, patch_overlap=tuple(s // 2 for s in patch_size))
    patches = patch_grid_sample(image, starts, patch_size)
    outs = []
    with torch.no_grad():
        for i in range(patches.shape[0]):
            patch = patches[i : i + 1].contiguous()
            outs.append(model({v: patch if v == view_to_patch else image_dict[v] for v in views}).float())
    return combine(torch.cat(outs, dim=0))
