"""Single-node data parallelism over RCCL / xGMI for the MAE pre-training step.

Replaces ``torch.nn.parallel.DistributedDataParallel`` as used by the reference (``cinema/device.py:35-48,86-104``,
``cinema/mae/pretrain.py:304-305,343``): one process per GPU, the model replicated, the *flat* fp32 gradient buffer of
:class:`cinema_amd.optim.FlatModel` all-reduced (mean) once per optimisation step in a few large buckets.  On the
8-GPU MI355X mesh every GPU has 7 point-to-point xGMI links, so a handful of >= 64 MB messages (RCCL picks its direct
algorithms for those) beat DDP's ~20 buckets of 25 MiB; gradient accumulation steps skip the collective entirely
(the reference all-reduces on every micro-step because it never uses ``no_sync``).
"""

from __future__ import annotations

import datetime
import os
import socket

import torch
import torch.distributed as dist


def get_free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        return s.getsockname()[1]


def ddp_setup(rank: int, world_size: int, port: int | None = None, backend: str | None = None) -> None:
    """Join the process group (reference ``cinema/device.py:35-48``; backend "nccl" IS RCCL on ROCm)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if port is not None:
        os.environ["MASTER_PORT"] = str(port)
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(rank % max(torch.cuda.device_count(), 1))
    dist.init_process_group(backend=backend, rank=rank, world_size=world_size, timeout=datetime.timedelta(seconds=5400))


class GradientSynchronizer:
    """Mean all-reduce of a flat gradient buffer in ``n_buckets`` contiguous chunks, plus the one-off parameter broadcast."""

    def __init__(self, world_size: int | None = None, bucket_bytes: int = 128 << 20, group=None) -> None:  # noqa: ANN001
        self.world_size = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.bucket_bytes = bucket_bytes
        self.group = group
        self.flat = None
        self.buckets: list = []

    def attach(self, flat) -> None:  # noqa: ANN001
        self.flat = flat
        per = max(1, self.bucket_bytes // 4)
        n = flat.flat_grad.numel()
        self.buckets = [flat.flat_grad[i:min(n, i + per)] for i in range(0, n, per)]
        if self.world_size > 1:
            dist.broadcast(flat.flat_param, src=0, group=self.group)  # rank 0's weights everywhere (DDP's _sync_module_states)

    def all_reduce(self) -> None:
        if self.world_size <= 1:
            return
        backend = dist.get_backend(self.group)
        works = []
        for b in self.buckets:
            if backend == "nccl":
                works.append(dist.all_reduce(b, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
            else:  # gloo (CPU tests) has no AVG
                works.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if backend != "nccl":
            self.flat.flat_grad.div_(self.world_size)

    def all_finite(self, loss: torch.Tensor) -> torch.Tensor:
        """Collective NaN decision (a rank-local ``continue`` as in ``pretrain.py:255-257`` would dead-lock DDP)."""
        flag = torch.isfinite(loss.detach()).to(torch.float32).reshape(1)
        if self.world_size > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return flag
