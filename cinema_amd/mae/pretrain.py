"""The epoch loop of the MAE pre-training recipe around the fused step (reference ``cinema/mae/pretrain.py:203-284``, ``pretrain_one_epoch``).

What is mirrored: the per-ITERATION learning-rate update with the fractional epoch ``i / len(dataloader) + epoch`` (``pretrain.py:243-250``), the
gradient-accumulation boundary ``(i + 1) % n_accum_steps == 0``, loss / ``n_accum_steps``, clip + AdamW + zero_grad on the boundary, the sample counter,
and the logged values (metrics of the step, ``grad_norm``, ``lr``, ``n_samples``).  What differs, on purpose: the non-finite guard is the device-side one
of :class:`cinema_amd.optim.FusedAdamW` (a non-finite gradient norm skips the update on every rank without a host read-back; the reference reads
``loss`` back each iteration and ``continue``s); metrics stay on the device until a log callback asks for them.  Not rebuilt: data set discovery,
monai transforms, hydra / wandb (control plane around the path)."""

from __future__ import annotations

from typing import Callable, Iterable

import torch

from cinema_amd.optim import TrainStep, adjust_learning_rate, get_n_accum_steps  # noqa: F401  (re-exported like the reference module)


def pretrain_one_epoch(step: TrainStep, dataloader: Iterable, n_accum_steps: int, world_size: int, config, epoch: int, n_samples: int,  # noqa: ANN001
                       log: Callable | None = None) -> int:
    """One epoch of ``step`` (a :class:`TrainStep` holding the model, the fused optimiser and, for data parallel runs, the gradient synchroniser)
    over ``dataloader`` (batches: dict view -> image tensor).  ``config.train``: ``batch_size_per_device``, ``enc_mask_ratio``, ``n_warmup_epochs``,
    ``n_epochs``, ``lr``, ``min_lr`` (``clip_grad`` is the step's).  Returns the updated ``n_samples``; ``log(dict)`` is called on update iterations."""
    tr = config.train
    batch_size_per_step = tr.batch_size_per_device * world_size
    n_iter = len(dataloader)
    device = next(step.model.parameters()).device
    for i, batch in enumerate(dataloader):
        lr = adjust_learning_rate(optimizer=step, step=i / n_iter + epoch, warmup_steps=tr.n_warmup_epochs, max_n_steps=tr.n_epochs, lr=tr.lr,
                                  min_lr=tr.min_lr)
        update_grad = (i + 1) % n_accum_steps == 0
        images = {k: v.to(device, non_blocking=True) for k, v in batch.items()}
        loss, grad_norm, metrics = step(images, tr.enc_mask_ratio, n_accum_steps=n_accum_steps, update_grad=update_grad)
        n_samples += batch_size_per_step
        if update_grad and log is not None:
            views = step.model.views
            prefix = f"{views[0]}_" if len(views) == 1 else ""
            out = {f"{prefix}{k}": v for k, v in metrics.items()}
            out.update({"loss": loss, "grad_norm": grad_norm, "lr": lr, "n_samples": n_samples})
            log(out)
    return n_samples


class SyntheticCine(torch.utils.data.Dataset):
    """U[0, 1) volumes of the model's view shapes (the value range ``ScaleIntensityd`` produces, ``pretrain.py:184``): stands in for the reference's
    NIfTI readers where no data set is mounted (benchmarks, smoke runs)."""

    def __init__(self, image_size_dict: dict, in_chans_dict: dict, length: int, seed: int = 0) -> None:
        self.sizes, self.chans, self.length, self.seed = dict(image_size_dict), dict(in_chans_dict), length, seed

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, idx: int) -> dict:
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + idx)
        return {v: torch.rand(self.chans[v], *self.sizes[v], generator=g) for v in self.sizes}
