"""Device / data-parallel setup helpers with the names of the reference's ``cinema/device.py``.

``ddp_setup`` and ``get_free_port`` live in :mod:`cinema_amd.ddp` (RCCL process group, flat-buffer gradient exchange); this module adds the
remaining callers' symbols (``get_amp_dtype_and_device``, ``print_model_info``, ``setup_ddp_model``) so that
``from cinema.device import ...`` lines of the reference's training scripts resolve unchanged.
"""

from __future__ import annotations

import logging

import torch
from torch import nn

from cinema_amd.ddp import GradientSynchronizer, ddp_setup, get_free_port

logger = logging.getLogger(__name__)


def get_amp_dtype_and_device() -> tuple:
    """(amp dtype, device) as the reference picks them (``cinema/device.py:51-72``): bf16 on a GPU that supports it (every MI355X), fp16
    otherwise, CPU when no GPU is visible.  The HIP path always computes bf16-MFMA / fp32-accumulate, so a surrounding
    ``torch.autocast(dtype=amp_dtype)`` is harmless and unnecessary; MIOpen auto-tuning (the reference's ``cudnn.benchmark``) is not used -
    no ATen convolution runs on this path."""
    amp_dtype = torch.float16
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
        device = torch.device("cuda")
        if torch.cuda.is_bf16_supported():
            amp_dtype = torch.bfloat16
            logger.info("Using bfloat16 for automatic mixed precision.")
    else:
        logger.info("CUDA is not available, using CPU.")
        device = torch.device("cpu")
    return amp_dtype, device


def print_model_info(model: nn.Module) -> None:
    """Parameter counts (reference ``cinema/device.py:75-83``)."""
    n_params = sum(p.numel() for p in model.parameters())
    logger.info(f"number of parameters: {n_params:,}")
    n_trainable_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    logger.info(f"number of trainable parameters: {n_trainable_params:,}")


def setup_ddp_model(model: nn.Module, device: torch.device, rank: int, world_size: int) -> tuple:  # noqa: ARG001
    """-> (model, model_wo_ddp) like the reference (``cinema/device.py:86-104``).  The model is NOT wrapped in ``DistributedDataParallel``:
    its whole forward is one autograd node writing into a flat gradient buffer, so the gradient exchange is the flat-buffer mean all-reduce
    of :class:`cinema_amd.ddp.GradientSynchronizer`, overlapped with the backward pass.  For ``world_size > 1`` the synchroniser is created
    here and left on the model (``model.grad_synchronizer``); ``cinema_amd.optim.TrainStep`` / ``FusedAdamW`` pick it up (rank 0's weights are
    broadcast when the flat buffers are built, DDP's ``_sync_module_states``).  Both returned handles are the model itself."""
    model.to(device)
    if world_size > 1:
        model.grad_synchronizer = GradientSynchronizer(world_size)
    return model, model


__all__ = ["ddp_setup", "get_amp_dtype_and_device", "get_free_port", "print_model_info", "setup_ddp_model"]
