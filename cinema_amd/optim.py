"""Optimisation step of the MAE pre-training hot path on flat HBM buffers (harness semantics of the reference
``cinema/optim.py`` + ``cinema/mae/pretrain.py:242-269``).

``FlatModel`` re-homes every trainable parameter of a model into ONE contiguous fp32 buffer (weight-decay groups are
contiguous ranges) and gives every parameter a gradient view into a second flat buffer.  One optimisation step is then
a handful of kernels regardless of the parameter count: memset(grads) -> [forward/backward accumulate straight into the
flat gradient buffer] -> (RCCL all-reduce of the flat buffer in a few large buckets) -> squared-norm -> clip coefficient
-> fused AdamW per decay group.  No per-tensor Python loops, no host synchronisation (the gradient norm stays on the
device until the caller asks for it).
"""

from __future__ import annotations

import math

import torch
from torch import nn

from cinema_amd import hip as K
from cinema_amd import tape as T


def adjust_learning_rate(optimizer, step: float, warmup_steps: float, max_n_steps: float, lr: float, min_lr: float) -> float:  # noqa: ANN001
    """Linear warm-up then half-cosine decay, applied to every param group (``lr_scale`` aware).

    Same schedule as the reference (``cinema/optim.py:21-52``); ``step`` is a fractional epoch (``pretrain.py:243-250``).
    """
    if step < warmup_steps:
        cur = lr * step / warmup_steps
    else:
        cur = min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (step - warmup_steps) / (max_n_steps - warmup_steps)))
    for group in optimizer.param_groups:
        group["lr"] = cur * group["lr_scale"] if "lr_scale" in group else cur
    return cur


def get_n_accum_steps(batch_size: int, batch_size_per_device: int, world_size: int) -> int:
    """Gradient-accumulation factor (reference ``cinema/optim.py:122-143``)."""
    per_step = batch_size_per_device * world_size
    if per_step > batch_size:
        raise ValueError(f"batch_size_per_step {per_step} should be less than batch_size {batch_size}.")
    if batch_size % per_step != 0:
        raise ValueError(f"batch_size {batch_size} should be divisible by batch_size_per_step {per_step}.")
    return batch_size // per_step


def param_groups_weight_decay(model: nn.Module, weight_decay: float = 1e-5, no_weight_decay_list: tuple = ()) -> list:
    """timm's split used by the reference (``pretrain.py:365``): ``ndim <= 1`` or ``*.bias`` -> no decay; tokens are decayed."""
    skip = set(no_weight_decay_list)
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (p.ndim <= 1 or name.endswith(".bias") or name in skip) else decay).append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


class FlatModel:
    """Flat fp32 parameter / gradient storage for a model's trainable parameters, grouped by weight decay."""

    def __init__(self, model: nn.Module, weight_decay: float) -> None:
        self.model = model
        self.groups = param_groups_weight_decay(model, weight_decay)
        params = [p for g in self.groups for p in g["params"]]
        if not params:
            raise ValueError("model has no trainable parameters")
        dev = params[0].device
        # every view 16-byte aligned in ALL three buffers: 8 elements (the bf16 shadow of a parameter that followed a 4-element bias was only 8-byte
        # aligned, and every GEMM reading it fell back to the generic kernel - 57 of 106 ms of the ConvUNetR step)
        sizes = [((p.numel() + 7) // 8) * 8 for p in params]
        total = sum(sizes)
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        # bf16 shadow of every parameter in the SAME order: the AdamW kernel writes it together with the fp32 master, so the
        # forward needs no per-weight cast kernels, and adjacent weights (attn.q | attn.kv) form one fused GEMM operand for free
        self.flat_shadow = torch.zeros(total, dtype=torch.bfloat16, device=dev) if dev.type == "cuda" else None
        self.ranges = []
        self.params = params
        self.offsets: dict = {}  # id(param) -> (begin, end) element range in the flat buffers (end includes the alignment pad)
        off = 0
        it = iter(sizes)
        for g in self.groups:
            start = off
            for p in g["params"]:
                n = next(it)
                view = self.flat_param[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)
                p._cinema_flat_grad = p.grad  # noqa: SLF001  (tape.PVar accumulates straight into this view)
                if self.flat_shadow is not None:
                    p._cinema_shadow = self.flat_shadow[off:off + p.numel()]  # noqa: SLF001
                self.offsets[id(p)] = (off, off + n)
                off += n
            self.ranges.append((start, off))
        self.numel = total
        self.refresh_shadows()

    def zero_grad(self) -> None:
        self.flat_grad.zero_()

    def refresh_shadows(self) -> None:
        """Re-derive every bf16 shadow from the fp32 masters (after construction / ``load_state_dict``); the optimiser keeps
        them in sync afterwards.  A shadow is trusted only while the parameter's autograd version is the one stamped here."""
        if self.flat_shadow is None:
            return
        K.cast(self.flat_param, torch.bfloat16, out=self.flat_shadow)
        for p in self.params:
            p._cinema_shadow_version = p._version  # noqa: SLF001


class FusedAdamW:
    """AdamW (+ global-norm clipping) over a :class:`FlatModel` with ``torch.optim.AdamW`` update semantics.

    ``param_groups`` mimics the torch optimiser attribute so that :func:`adjust_learning_rate` works unchanged.
    """

    def __init__(self, flat: FlatModel, lr: float = 1e-3, betas: tuple = (0.9, 0.95), eps: float = 1e-8) -> None:
        self.flat = flat
        self.param_groups = [{"params": g["params"], "weight_decay": g["weight_decay"], "lr": lr} for g in flat.groups]
        self.betas, self.eps = tuple(betas), eps
        self.exp_avg = torch.zeros_like(flat.flat_param)
        self.exp_avg_sq = torch.zeros_like(flat.flat_param)
        dev = flat.flat_param.device
        self.sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.coef = torch.ones(1, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        # [updates applied, updates skipped]: lives on the device because the non-finite-gradient decision does (no host sync in a step)
        self.step_state = torch.zeros(2, dtype=torch.int32, device=dev)

    @property
    def step_count(self) -> int:
        """Updates applied so far = the Adam step of the bias corrections (a device read-back; not used inside a step)."""
        return int(self.step_state[0])

    @property
    def n_skipped(self) -> int:
        """Updates skipped because the (all-reduced) gradient norm was not finite."""
        return int(self.step_state[1])

    def zero_grad(self, set_to_none: bool = False) -> None:  # noqa: ARG002
        self.flat.zero_grad()

    def step(self, clip_grad: float | None = None) -> torch.Tensor:
        """One update; returns the pre-clip global gradient norm as a device scalar (``cinema/optim.py:208-210``).

        Non-finite guard, decided on the device: the reference skips the step on a NaN loss (``cinema/mae/pretrain.py:255-257``) and its
        ``GradScaler.step`` skips the optimiser when a gradient is inf / NaN.  Here a NaN loss back-propagates NaN into the flat gradient
        buffer, the squared norm is NaN, ``clip_coef`` writes coef = 0 and the AdamW kernels return without touching parameters, moments or
        bf16 shadows; the Adam step count does not advance.  Under data parallelism the norm is taken AFTER the mean all-reduce, so every
        rank sees the same NaN and skips together (no rank-local ``continue`` that would dead-lock the collectives)."""
        f = self.flat
        self.sq.zero_()
        K.sqnorm(f.flat_grad, self.sq)
        K.clip_coef(self.sq, float(clip_grad) if clip_grad else 0.0, self.coef, self.grad_norm, self.step_state)
        for group, (a, b) in zip(self.param_groups, f.ranges):
            if b > a:
                K.adamw(f.flat_param[a:b], f.flat_grad[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], group["lr"], self.betas[0], self.betas[1],
                        self.eps, group["weight_decay"], 1, clip=self.coef,
                        shadow=None if f.flat_shadow is None else f.flat_shadow[a:b], step_state=self.step_state)
        T.WEIGHTS.invalidate()  # parameters were written through raw pointers: re-laid-out shadows (patch convs) are rebuilt next forward
        return self.grad_norm

    def state_dict(self) -> dict:
        return {"step": self.step_count, "skipped": self.n_skipped, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lrs": [g["lr"] for g in self.param_groups]}

    def load_state_dict(self, state: dict) -> None:
        self.step_state[0] = int(state["step"])
        self.step_state[1] = int(state.get("skipped", 0))
        self.exp_avg.copy_(state["exp_avg"])
        self.exp_avg_sq.copy_(state["exp_avg_sq"])
        for g, lr in zip(self.param_groups, state["lrs"]):
            g["lr"] = lr


class TrainStep:
    """forward -> backward -> (gradient all-reduce) -> clip -> AdamW -> zero_grad, the body of ``pretrain_one_epoch``
    (``cinema/mae/pretrain.py:242-269``) without its per-step host synchronisations."""

    def __init__(self, model: nn.Module, lr: float = 1e-3, betas: tuple = (0.9, 0.95), weight_decay: float = 0.05, clip_grad: float | None = 5.0,
                 synchronizer=None, hip_graph: bool = False, replay: bool = False, audit: bool = False) -> None:  # noqa: ANN001
        self.model = model
        self.flat = FlatModel(model, weight_decay)
        self.optimizer = FusedAdamW(self.flat, lr=lr, betas=betas)
        self.clip_grad = clip_grad
        self.sync = synchronizer
        if self.sync is not None:
            self.sync.attach(self.flat)
        # hip_graph: forward + backward (~2000 launches on two streams) are captured once per input signature and replayed as one HIP
        # graph, which takes the host out of the step; clip + AdamW stay eager (their scalars change every step).  Single process only:
        # the overlapped RCCL collectives are issued from Python hooks in the backward pass.
        self.hip_graph = hip_graph
        # replay: forward + backward are recorded once per input signature as the flat list of this library's launches and re-issued
        # from that list (cinema_amd/replay.py) - the host cost of a step drops from ~33 ms of module code to ~3 us per launch
        self.replay, self.audit = replay, audit
        self._recorded: dict = {}
        self._graphs: dict = {}
        if hip_graph and synchronizer is not None:
            raise ValueError("hip_graph=True captures the single-process step; the data-parallel step runs eagerly")

    def __call__(self, image_dict: dict, enc_mask_ratio: float, enc_mask_dict: dict | None = None, n_accum_steps: int = 1, update_grad: bool = True):  # noqa: ANN204
        if self.replay and enc_mask_dict is None and n_accum_steps == 1:
            return self._replay_step(image_dict, enc_mask_ratio, update_grad)
        if self.hip_graph and enc_mask_dict is None and n_accum_steps == 1 and update_grad:
            return self._graph_step(image_dict, enc_mask_ratio)
        loss, _, _, metrics = self.model(image_dict, enc_mask_ratio, enc_mask_dict=enc_mask_dict)
        if self.sync is not None:
            self.sync.arm(update_grad)  # on the micro-step that ends with the optimiser update, blocks all-reduce as their gradients complete
        (loss / n_accum_steps if n_accum_steps > 1 else loss).backward()
        grad_norm = None
        if update_grad:
            if self.sync is not None:
                self.sync.all_reduce()
            grad_norm = self.optimizer.step(self.clip_grad)
            self.optimizer.zero_grad()
        return loss.detach(), grad_norm, metrics

    # ------------------------------------------------------------------------------------------------ recorded step
    def reset_recordings(self) -> None:
        """Drop the recorded steps (and their memory pools).  A recording is the launch list of the model AS IT WAS when recorded: call this
        after changing ``requires_grad`` flags, swapping sub-modules, switching train / eval behaviour or resizing parameters; input
        shapes and the mask ratio are part of the key and need no reset."""
        self._recorded.clear()
        self._graphs.clear()

    def _replay_step(self, image_dict: dict, enc_mask_ratio: float, update_grad: bool):  # noqa: ANN202
        from cinema_amd.replay import RecordedStep

        if self.sync is not None:
            self.sync.arm(update_grad)
        key = (float(enc_mask_ratio), tuple((k, tuple(v.shape), v.dtype) for k, v in image_dict.items()))
        rec = self._recorded.get(key)
        if rec is None:  # the recording IS this step (an eager forward + backward under hip.RECORD)
            rec = self._recorded[key] = RecordedStep(self.model, image_dict, enc_mask_ratio, audit=self.audit)
            loss, metrics = rec.loss, rec.metrics
        else:
            loss, metrics = rec.run(image_dict)
        grad_norm = None
        if update_grad:
            if self.sync is not None:
                self.sync.all_reduce()
            grad_norm = self.optimizer.step(self.clip_grad)
            self.optimizer.zero_grad()
        return loss, grad_norm, metrics  # static tensors: overwritten by the next step

    # ------------------------------------------------------------------------------------------------ HIP-graph step
    def _capture(self, image_dict: dict, enc_mask_ratio: float) -> tuple:
        static = {k: v.clone() for k, v in image_dict.items()}
        self.optimizer.zero_grad()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up on a non-default stream: per-stream workspaces, allocator pools, neighbour lists
            for _ in range(2):
                loss, _, _, _ = self.model(static, enc_mask_ratio)
                loss.backward()
            self.optimizer.zero_grad()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            loss, _, _, metrics = self.model(static, enc_mask_ratio)
            loss.backward()
        return graph, static, loss.detach(), {k: v.detach() for k, v in metrics.items()}

    def _graph_step(self, image_dict: dict, enc_mask_ratio: float):  # noqa: ANN202
        key = (float(enc_mask_ratio), tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(image_dict.items())))
        entry = self._graphs.get(key)
        if entry is None:
            entry = self._graphs[key] = self._capture(image_dict, enc_mask_ratio)
        graph, static, loss, metrics = entry
        for k, v in image_dict.items():
            if v.data_ptr() != static[k].data_ptr():
                static[k].copy_(v, non_blocking=True)
        graph.replay()  # random masks are drawn inside the graph (torch's graph-safe Philox offsets advance per replay)
        grad_norm = self.optimizer.step(self.clip_grad)
        self.optimizer.zero_grad()
        return loss, grad_norm, metrics  # static output buffers: overwritten by the next replay
