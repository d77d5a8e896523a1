"""Single-node data parallelism over RCCL / xGMI for the MAE pre-training step.

Replaces ``torch.nn.parallel.DistributedDataParallel`` as used by the reference (``cinema/device.py:35-48,86-104``,
``cinema/mae/pretrain.py:304-305,343``): one process per GPU, the model replicated, the *flat* fp32 gradient buffer of
:class:`cinema_amd.optim.FlatModel` all-reduced (mean) once per optimisation step: the ranges of each transformer block go
out as soon as the block's backward kernels are launched (~28 MB per encoder block, overlapped with the rest of the
backward), the remainder in a few large buckets at the end.  On the 8-GPU MI355X mesh every GPU has 7 point-to-point
xGMI links, so tens-of-MB messages (RCCL picks its direct algorithms for those) beat DDP's ~20 buckets of 25 MiB; gradient
accumulation steps skip the collective entirely (the reference all-reduces on every micro-step because it never uses
``no_sync``).
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
    """Mean all-reduce of the flat gradient buffer, overlapped with the backward pass, plus the one-off parameter broadcast.

    ``cinema_amd.tape.mark_params`` tells :meth:`params_done`, during the backward pass, when every gradient kernel of a
    transformer block has been launched; the block's flat ranges are all-reduced right away (``async_op``: RCCL's stream
    waits for the work queued so far and then runs beside the rest of the backward).  :meth:`all_reduce` reduces what is
    left (stem, fusion, heads, tokens) in buckets and waits for everything before the optimiser reads the gradients.
    """

    def __init__(self, world_size: int | None = None, bucket_bytes: int = 128 << 20, group=None, overlap: bool = True,
                 force_collectives: bool = False, exchange_dtype: torch.dtype = torch.float32, min_early_bytes: int = 1 << 20,
                 algorithm: str = "all_reduce") -> None:  # noqa: ANN001
        # "all_reduce": one torch.distributed.all_reduce per range (RCCL picks ring / tree / direct by message size).  "rs_ag": the same mean as an explicit
        # reduce-scatter of the range followed by an all-gather of the reduced shards (both in place on the range).  On the fully connected xGMI mesh
        # (7 point-to-point links per GPU) every rank then sends 1/N of the range straight to its owner and gets the reduced shards straight back:
        # two one-hop phases instead of a ring's 2(N-1) steps.  Same result up to the summation order inside the collective.
        if algorithm not in ("all_reduce", "rs_ag"):
            raise ValueError('algorithm: "all_reduce" or "rs_ag"')
        self.algorithm = algorithm
        self.world_size = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.bucket_bytes = bucket_bytes
        self.group = group
        self.flat = None
        self.buckets: list = []
        self.overlap = overlap
        self.force = force_collectives  # issue the collectives even in a one-process group (exercises the RCCL path on one GPU)
        self.armed = False
        # a block's range goes out early only if it is worth a collective of its own: below ~1 MiB an all-reduce over xGMI is latency-bound
        # (launch + 7 point-to-point hops), so biases / LayerNorm vectors ride in the final buckets
        self.min_early = max(1, min_early_bytes // 4)  # elements
        # bf16 exchange (optional): each rank's fp32 range is rounded to bf16 into a staging buffer, the collective moves and sums bf16 (half the
        # xGMI bytes: 487 -> 244 MB per step for ViT-Base), and the mean is widened back into the fp32 gradient buffer.  Every rank receives the
        # same bf16 result, so the replicas stay bit-identical; the rounding error is bounded by the tests (rel-L2 <= 1e-2 against the fp32 exchange)
        if exchange_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("exchange_dtype: torch.float32 or torch.bfloat16")
        self.exchange_dtype = exchange_dtype
        self._early: list = []   # (begin, end) ranges already handed to a collective in this backward pass
        self._works: list = []
        self._bytes = 0
        self._staged: list = []  # bf16 exchange: (fp32 range view, bf16 staging tensor) pairs to widen back after the wait
        self.n_early_last = 0    # early (overlapped) collectives of the last completed exchange (diagnostics / tests)
        self.disabled = False    # measurement switches (bench.py, N > 1): no exchange at all / everything at the end of the backward pass
        self.defer_all = False
        self.n_collectives_total = 0  # collectives issued since construction (tests: accumulation micro-steps must not communicate)
        self.bytes_last = 0      # payload bytes of the last completed exchange
        self.n_collectives_last = 0  # collectives of the last completed exchange (rs_ag: two per range)
        self._n_coll_armed = 0

    def attach(self, flat) -> None:  # noqa: ANN001
        self.flat = flat
        per = max(1, self.bucket_bytes // 4)
        n = flat.flat_grad.numel()
        self.buckets = [flat.flat_grad[i:min(n, i + per)] for i in range(0, n, per)]
        if self.world_size > 1 or self.force:
            dist.broadcast(flat.flat_param, src=0, group=self.group)  # rank 0's weights everywhere (DDP's _sync_module_states)
            # the masters changed under the bf16 operand shadows: re-derive them (a rank that was initialised differently would otherwise keep
            # computing with its own weights while holding rank 0's masters)
            flat.refresh_shadows()
            from cinema_amd import tape as T

            T.WEIGHTS.invalidate()
        if self.overlap:
            from cinema_amd import tape as T

            T.PARAMS_DONE_HOOK = self.params_done

    def _launch(self, t: torch.Tensor) -> None:
        op = dist.ReduceOp.AVG if dist.get_backend(self.group) == "nccl" else dist.ReduceOp.SUM  # gloo (CPU tests) has no AVG
        if self.exchange_dtype == torch.bfloat16:
            if t.is_cuda:
                from cinema_amd import hip as K

                rec, K.RECORD = K.RECORD, None  # inside a recorded step this runs as a HOST entry (re-run on every replay): its launches are not list entries
                try:
                    stage = K.cast(t, torch.bfloat16)
                finally:
                    K.RECORD = rec
            else:
                stage = t.to(torch.bfloat16)
            self._staged.append((t, stage))
            t = stage
        if self.algorithm == "rs_ag" and (self.world_size > 1 or self.force):
            # (forced one-rank group, bench.py --force-sync: s = numel, the shard IS the range - the RCCL branch below, with its in-place reduce-scatter whose output
            # aliases its input and the all-gather queued behind it without a wait, is what runs on a one-GPU box)
            self._launch_rs_ag(t, op)
        else:
            self._works.append(dist.all_reduce(t, op=op, group=self.group, async_op=True))
            self.n_collectives_total += 1
        self._bytes += t.numel() * t.element_size()

    def _launch_rs_ag(self, t: torch.Tensor, op) -> None:  # noqa: ANN001
        """Mean of ``t`` over the ranks as reduce-scatter + all-gather, in place: rank r reduces elements [r s, (r + 1) s) of the range (s = numel // world),
        then every rank gathers the reduced shards back into the range.  The flat ranges are multiples of 8 elements, so for 2 / 4 / 8 ranks nothing is left
        over; a remainder of < world elements (other world sizes) takes a small all-reduce."""
        world = self.world_size
        rank = dist.get_rank(self.group)
        s = t.numel() // world
        if s > 0:
            main = t[:s * world]
            shard = main[rank * s:(rank + 1) * s]
            w = dist.reduce_scatter_tensor(shard, main, op=op, group=self.group, async_op=True)
            if dist.get_backend(self.group) != "nccl":
                w.wait()  # gloo (CPU tests) runs queued collectives on worker threads: the gather must not overtake the scatter.  RCCL orders both on its stream
            else:
                self._works.append(w)
            self._works.append(dist.all_gather_into_tensor(main, shard, group=self.group, async_op=True))
            self.n_collectives_total += 2
        if s * world < t.numel():
            self._works.append(dist.all_reduce(t[s * world:], op=op, group=self.group, async_op=True))
            self.n_collectives_total += 1

    def arm(self, on: bool) -> None:
        """Called before ``backward()``: only the micro-step that ends with the optimiser update all-reduces."""
        self.armed = bool(on) and (self.world_size > 1 or self.force) and self.flat is not None and not self.disabled
        self._early, self._works, self._staged, self._bytes = [], [], [], 0
        self._n_coll_armed = self.n_collectives_total

    def params_done(self, tape, params: list) -> None:  # noqa: ANN001
        """Backward-pass hook: all gradient kernels of ``params`` are in the stream -> start their all-reduce.  Every rank
        takes the same decisions (they depend on the model only), so the collectives stay matched."""
        if not self.armed or self.defer_all:
            return
        ranges = []
        for p in params:
            r = self.flat.offsets.get(id(p))
            pv = tape.pvars.get(id(p))
            if r is None or pv is None or not pv.direct:  # gradient not accumulated straight into the flat buffer: leave it for the end
                return
            ranges.append(r)
        ranges.sort()
        merged = []
        for a, b in ranges:
            if merged and merged[-1][1] == a:
                merged[-1][1] = b
            else:
                merged.append([a, b])
        for a, b in merged:
            if b - a < self.min_early:  # biases / LayerNorm vectors: latency-bound as separate collectives, they ride in the final buckets
                continue
            self._launch(self.flat.flat_grad[a:b])
            self._early.append((a, b))

    def all_reduce(self) -> None:
        if (self.world_size <= 1 and not self.force) or self.disabled:
            return
        n = self.flat.flat_grad.numel()
        per = max(1, self.bucket_bytes // 4)
        pos = 0
        for a, b in sorted(self._early) + [(n, n)]:  # the complement of what went out early, in buckets
            while pos < a:
                e = min(a, pos + per)
                self._launch(self.flat.flat_grad[pos:e])
                pos = e
            pos = max(pos, b)
        for w in self._works:
            w.wait()
        for dst, stage in self._staged:  # bf16 exchange: widen the reduced values back into the fp32 gradient buffer
            if dst.is_cuda:
                from cinema_amd import hip as K

                K.cast(stage, torch.float32, out=dst)
            else:
                dst.copy_(stage)
        if dist.get_backend(self.group) != "nccl":
            self.flat.flat_grad.div_(self.world_size)
        self.n_early_last = len(self._early)
        self.bytes_last = self._bytes
        self.n_collectives_last = self.n_collectives_total - self._n_coll_armed
        self._early, self._works, self._staged, self.armed = [], [], [], False

    def all_finite(self, loss: torch.Tensor) -> torch.Tensor:
        """Collective NaN decision (a rank-local ``continue`` as in ``pretrain.py:255-257`` would dead-lock DDP)."""
        flag = torch.isfinite(loss.detach()).to(torch.float32).reshape(1)
        if self.world_size > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return flag
