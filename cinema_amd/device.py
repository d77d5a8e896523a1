"""Device / data-parallel setup helpers with the names of the reference's ``cinema/device.py``.

``ddp_setup`` and ``get_free_port`` live in :mod:`cinema_amd.ddp` (RCCL process group, flat-buffer gradient exchange); this module adds the
remaining callers' symbols (``get_amp_dtype_and_device``, ``print_model_info``, ``setup_ddp_model``) so that
``from cinema.device import ...`` lines of the reference's training scripts resolve unchanged.
"""

from __future__ import annotations

import os

import logging

import torch
from torch import nn

from cinema_amd.ddp import GradientSynchronizer, ddp_setup, get_free_port

logger = logging.getLogger(__name__)


def get_amp_dtype_and_device() -> tuple:
    """-> (autocast dtype, device), the pair the reference's scripts ask for first (``cinema/device.py:51-72``): the GPU with bf16 when the device has
    it (every MI355X does), fp16 on an older GPU, and the CPU with fp16 when no GPU is visible.  On the HIP path the dtype is informational - the
    kernels always run bf16 MFMA with fp32 accumulation, a surrounding ``torch.autocast`` changes nothing - and there is no library auto-tuning
    switch to flip (the reference enables ``cudnn.benchmark``; no ATen convolution runs here)."""
    if not torch.cuda.is_available():
        logger.info("no GPU visible: running on the CPU")
        return torch.float16, torch.device("cpu")
    torch.cuda.empty_cache()
    bf16 = torch.cuda.is_bf16_supported()
    logger.info("autocast dtype: %s", "bfloat16" if bf16 else "float16")
    return (torch.bfloat16 if bf16 else torch.float16), torch.device("cuda")


def print_model_info(model: nn.Module) -> None:
    """Log the total and the trainable parameter count (``cinema/device.py:75-83``)."""
    sizes = [(p.numel(), p.requires_grad) for p in model.parameters()]
    logger.info("parameters: %s total, %s trainable", f"{sum(n for n, _ in sizes):,}", f"{sum(n for n, t in sizes if t):,}")


def setup_ddp_model(model: nn.Module, device: torch.device, rank: int, world_size: int) -> tuple:  # noqa: ARG001
    """-> (model, model_wo_ddp) like the reference (``cinema/device.py:86-104``).  The model is NOT wrapped in ``DistributedDataParallel``:
    its whole forward is one autograd node writing into a flat gradient buffer, so the gradient exchange is the flat-buffer mean all-reduce
    of :class:`cinema_amd.ddp.GradientSynchronizer`, overlapped with the backward pass.  For ``world_size > 1`` the synchroniser is created
    here and left on the model (``model.grad_synchronizer``); ``cinema_amd.optim.TrainStep`` / ``FusedAdamW`` pick it up (rank 0's weights are
    broadcast when the flat buffers are built, DDP's ``_sync_module_states``).  Both returned handles are the model itself."""
    model.to(device)
    if world_size > 1:
        # CINEMA_GRAD_EXCHANGE=rs_ag: explicit reduce-scatter + all-gather instead of all_reduce (cinema_amd/ddp.py), for A/B on a real xGMI node
        model.grad_synchronizer = GradientSynchronizer(world_size, algorithm=os.environ.get("CINEMA_GRAD_EXCHANGE", "all_reduce"))
    return model, model


__all__ = ["ddp_setup", "get_amp_dtype_and_device", "get_free_port", "print_model_info", "setup_ddp_model"]
