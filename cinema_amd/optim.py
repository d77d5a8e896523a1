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

import os

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


def apply_optim_scheduler(optimizer, lr: float, last_layer_lr: float, weight_decay: float) -> None:  # noqa: ANN001
    """Write this iteration's learning rate / weight decay into every parameter group (reference ``cinema/optim.py:55-68``): groups carry ``lr_scale``,
    ``weight_decay_scale`` and ``is_last_layer`` (DINOv2-style schedules; the MAE pre-training path uses :func:`adjust_learning_rate` instead)."""
    for group in optimizer.param_groups:
        group["weight_decay"] = weight_decay * group["weight_decay_scale"]
        group["lr"] = (last_layer_lr if group["is_last_layer"] else lr) * group["lr_scale"]


class CosineScheduler:
    """freeze (zeros) -> linear warm-up -> half cosine from ``base_value`` to ``final_value``, as a table indexed by the iteration (reference
    ``cinema/optim.py:71-119``; past ``total_iters`` the value stays ``final_value``)."""

    def __init__(self, base_value: float, final_value: float, total_iters: int, warmup_iters: int = 0, start_warmup_value: float = 0.0,
                 freeze_iters: int = 0) -> None:
        import numpy as np

        self.final_value = final_value
        self.total_iters = total_iters
        iters = np.arange(total_iters - warmup_iters - freeze_iters)
        cosine = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * iters / len(iters)))
        self.schedule = np.concatenate((np.zeros((freeze_iters,)), np.linspace(start_warmup_value, base_value, warmup_iters), cosine))
        if len(self.schedule) != self.total_iters:
            raise ValueError(f"Length of schedule {len(self.schedule)} should be equal to total_iters {self.total_iters}.")

    def __getitem__(self, it: int):  # noqa: ANN204
        if it >= self.total_iters:
            return self.final_value
        return self.schedule[it]


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

    def __init__(self, model: nn.Module, weight_decay: float, param_groups: list | None = None) -> None:
        """``param_groups`` (optional): torch-style list of dicts with ``params`` and per-group ``weight_decay`` / ``lr_scale`` (e.g.
        ``cinema_amd.convvit.param_groups_lr_decay``); default: timm's two-group weight-decay split used by the reference pre-training."""
        self.model = model
        self.groups = [dict(g) for g in param_groups] if param_groups is not None else param_groups_weight_decay(model, weight_decay)
        self.groups = [g for g in self.groups if len(g["params"]) > 0] or self.groups
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
                    p._cinema_flat = self  # noqa: SLF001  (tape.w_fp8 asks the owner for the e4m3 shadow)
                self.offsets[id(p)] = (off, off + n)
                off += n
            self.ranges.append((start, off))
        self.numel = total
        self.refresh_shadows()

    def zero_grad(self) -> None:
        self.flat_grad.zero_()

    # ---- e4m3 weight shadows for the fp8 forward GEMMs (BASELINE config 5): one scale per matrix parameter, all of them re-quantised from the
    # bf16 shadows in three launches whenever the weights changed (cinema_quantize_fp8_segments); created on first use
    def fp8_shadow(self, p: nn.Parameter, transposed: bool = False):  # noqa: ANN201
        """-> (uint8 view of p's e4m3 shadow [numel], fp32 [1] scale) in sync with the current bf16 shadow, or None for a parameter outside the
        flat buffers / without a fresh bf16 shadow.  ``transposed``: the [in][out] copy (operand of the fp8 data-gradient GEMM), built for all
        2-D weights in one more launch behind the plain copies."""
        rng = self.offsets.get(id(p))
        if rng is None or self.flat_shadow is None or getattr(p, "_cinema_shadow_version", -1) != p._version or p.dim() < 2:  # noqa: SLF001
            return None
        if getattr(self, "_fp8", None) is None:
            mats = [q for q in self.params if q.dim() >= 2]
            dev, rows = self.flat_param.device, [[self.offsets[id(q)][0], self.offsets[id(q)][0] + q.numel() // 8 * 8] for q in mats]
            # (persistent: tables and shadows that outlive the step must not be carved out of a recording's memory pool, whose replayed launches rewrite it)
            bounds, data, scales = K.persistent(lambda: (torch.tensor(rows, dtype=torch.int64, device=dev), torch.zeros(self.numel, dtype=torch.uint8, device=dev),
                                                         torch.ones(len(mats), dtype=torch.float32, device=dev)))
            self._fp8 = {"index": {id(q): i for i, q in enumerate(mats)}, "bounds": bounds, "epoch": None, "data": data, "scales": scales}
        st = self._fp8
        if p.numel() % 8 or id(p) not in st["index"]:
            return None
        if st["epoch"] != T.WEIGHTS.epoch:  # the optimiser bumps the epoch after every update
            K.quantize_fp8_segments(self.flat_shadow, st["bounds"], st["data"], st["scales"])
            st["epoch"] = T.WEIGHTS.epoch
        i = st["index"][id(p)]
        a = rng[0]
        if transposed:
            if p.dim() != 2 or p.shape[0] % 8 or p.shape[1] % 8:
                return None
            if "data_t" not in st:
                mats = [q for q in self.params if q.dim() >= 2]
                dev, rows = self.flat_param.device, [[self.offsets[id(q)][0], q.shape[0], q.numel() // q.shape[0]] if (q.dim() == 2 and q.shape[0] % 8 == 0 and q.shape[1] % 8 == 0)
                                                     else [self.offsets[id(q)][0], 0, 0] for q in mats]
                st["desc_t"], st["data_t"] = K.persistent(lambda: (torch.tensor(rows, dtype=torch.int64, device=dev), torch.zeros(self.numel, dtype=torch.uint8, device=dev)))
                st["epoch_t"] = None
            if st["epoch_t"] != T.WEIGHTS.epoch:
                K.quantize_fp8_segments_t(self.flat_shadow, st["desc_t"], st["scales"], st["data_t"])
                st["epoch_t"] = T.WEIGHTS.epoch
            return st["data_t"][a:a + p.numel()], st["scales"][i:i + 1]
        return st["data"][a:a + p.numel()], st["scales"][i:i + 1]

    def fp8_shadow_cat_t(self, ps: tuple):  # noqa: ANN201
        """TRANSPOSED e4m3 shadow of several 2-D weights that lie back to back in the flat buffer and share their column count, quantised as ONE matrix with
        ONE scale: -> (uint8 [cols, total rows], fp32 [1] scale) or None.  The fused q | kv projection's data gradient dX = dQKV [W_q; W_kv] is then a single
        e4m3 GEMM with K = 3c (with per-weight scales it was two GEMMs meeting in an fp32 residual, i.e. no faster than the bf16 GEMM)."""
        if self.flat_shadow is None or any(p.dim() != 2 or getattr(p, "_cinema_shadow_version", -1) != p._version for p in ps):  # noqa: SLF001
            return None
        rngs = [self.offsets.get(id(p)) for p in ps]
        if any(r is None for r in rngs):
            return None
        cols = ps[0].shape[1]
        if any(p.shape[1] != cols or p.numel() % 8 for p in ps) or any(rngs[i][0] + ps[i].numel() != rngs[i + 1][0] for i in range(len(ps) - 1)):
            return None
        rows = sum(p.shape[0] for p in ps)
        if rows % 8 or cols % 8 or rngs[0][0] % 8:
            return None
        st = self.__dict__.get("_fp8cat")
        if st is None:
            dev = self.flat_param.device
            st = self.__dict__["_fp8cat"] = {"index": {}, "segs": [], "epoch": None, "tensors": None,
                                           "plain": K.persistent(lambda: torch.zeros(self.numel, dtype=torch.uint8, device=dev)),
                                           "data_t": K.persistent(lambda: torch.zeros(self.numel, dtype=torch.uint8, device=dev))}
        key = tuple(id(p) for p in ps)
        if key not in st["index"]:
            st["index"][key] = len(st["segs"])
            st["segs"].append((rngs[0][0], rows, cols))
            if st["tensors"] is not None:
                st.setdefault("retired", []).append(st["tensors"])  # launches already issued (or RECORDED) with the shorter lists read them: never freed
            st["tensors"] = None
        if st["tensors"] is None:  # descriptors of all joint segments (rebuilt while new ones appear: first step only)
            dev = self.flat_param.device
            segs = list(st["segs"])
            st["tensors"] = K.persistent(lambda: (torch.tensor([[a, a + r * c] for a, r, c in segs], dtype=torch.int64, device=dev),
                                                  torch.tensor([[a, r, c] for a, r, c in segs], dtype=torch.int64, device=dev),
                                                  torch.ones(len(segs), dtype=torch.float32, device=dev)))
            st["epoch"] = None
        if st["epoch"] != T.WEIGHTS.epoch:
            bounds, desc, scales = st["tensors"]
            K.quantize_fp8_segments(self.flat_shadow, bounds, st["plain"], scales)  # (its maxima -> the joint scales; the plain copy is not used)
            K.quantize_fp8_segments_t(self.flat_shadow, desc, scales, st["data_t"])
            st["epoch"] = T.WEIGHTS.epoch
        i = st["index"][key]
        a = rngs[0][0]
        return st["data_t"][a:a + rows * cols].view(cols, rows), st["tensors"][2][i:i + 1]

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

    def __init__(self, flat: FlatModel, lr: float = 1e-3, betas: tuple = (0.9, 0.95), eps: float = 1e-8, synchronizer=None) -> None:  # noqa: ANN001
        self.flat = flat
        # data-parallel gradient exchange of the flat buffer (cinema_amd.ddp.GradientSynchronizer); defaults to the one setup_ddp_model left on the model
        self.synchronizer = synchronizer if synchronizer is not None else getattr(flat.model, "grad_synchronizer", None)
        if self.synchronizer is not None and self.synchronizer.flat is not flat:
            self.synchronizer.attach(flat)
        self.param_groups = [{**{k: v for k, v in g.items() if k != "params"}, "params": g["params"], "weight_decay": g.get("weight_decay", 0.0),
                              "lr": lr * g["lr_scale"] if "lr_scale" in g else lr} for g in flat.groups]
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
        live = [(a, b, group["lr"], group["weight_decay"]) for group, (a, b) in zip(self.param_groups, f.ranges) if b > a]
        if 1 < len(live) <= K.ADAMW_MAX_GROUPS and f.flat_param.is_cuda:
            # the layer-decay groups of a fine-tuning step (28 for ViT-Base) as ONE launch: the ranges are ascending slices of the flat buffers
            K.adamw_groups(f.flat_param, f.flat_grad, self.exp_avg, self.exp_avg_sq, live, self.betas[0], self.betas[1], self.eps, self.coef,
                           f.flat_shadow, self.step_state)
            live = []
        for group, (a, b) in zip(self.param_groups, f.ranges):
            if b > a and live:
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
                 synchronizer=None, hip_graph: bool = False, replay: bool = False, audit: bool = False, param_groups: list | None = None,  # noqa: ANN001
                 check_every: int = 100) -> None:
        self.model = model
        # every ``check_every`` optimiser updates (and whenever a checkpoint is saved) the error words of the in-launch split reductions are read back
        # (hip.check_reduction_workspaces: one 4-byte read per workspace, the only host synchronisation of the step loop); 0 disables
        self.check_every, self._n_updates = int(check_every), 0
        self._flat = FlatModel(model, weight_decay, param_groups=param_groups)
        self._optimizer = FusedAdamW(self._flat, lr=lr, betas=betas, synchronizer=synchronizer)
        self.clip_grad = clip_grad
        self.sync = self._optimizer.synchronizer
        # hip_graph: forward + backward (~2000 launches on two streams) are captured once per input signature and replayed as one HIP
        # graph, which takes the host out of the step; clip + AdamW stay eager (their scalars change every step).  Single process only:
        # the overlapped RCCL collectives are issued from Python hooks in the backward pass.
        self.hip_graph = hip_graph
        # replay: forward + backward are recorded once per input signature as the flat list of this library's launches and re-issued
        # from that list (cinema_amd/replay.py) - the host cost of a step drops from ~33 ms of module code to ~3 us per launch
        self.replay, self.audit = replay, audit
        self._recorded: dict = {}
        self._fp8_calibrated = False  # set by the first eager step taken with the fp8 weight-gradient path on (see __call__)
        self._graphs: dict = {}
        if hip_graph and self.sync is not None:
            raise ValueError("hip_graph=True captures the single-process step; the data-parallel step runs eagerly")
    @property
    def flat(self) -> FlatModel:
        return self._flat

    @property
    def optimizer(self) -> FusedAdamW:
        return self._optimizer

    @property
    def param_groups(self) -> list:
        """The optimiser's groups (``adjust_learning_rate(optimizer=step, ...)`` in a step loop: host-side values only, no stream is made to wait)."""
        return self._optimizer.param_groups

    def _updated(self) -> None:
        self._n_updates += 1
        if self.check_every > 0 and self._n_updates % self.check_every == 0:
            _check_reductions_on_every_rank(self.sync)

    def __call__(self, image_dict: dict, enc_mask_ratio: float, enc_mask_dict: dict | None = None, n_accum_steps: int = 1, update_grad: bool = True):  # noqa: ANN204
        # the first fp8 step of THIS model runs eagerly: it records the maxima its sites take their first scales from and registers the joint e4m3 weight shadows
        # (FlatModel.fp8_shadow_cat_t), both of which a recording must find complete.  The registry of sites is per device, not per model - sites of an earlier
        # model that are still alive say nothing about this one (round 6: a recording taken on a model's very first step after another fp8 model had run in the
        # process replayed re-quantisation launches whose descriptor tensors had been replaced and freed -> memory fault) - so the step keeps its own flag
        calibrating = T.FP8_FORWARD and T.FP8_WGRAD and (not self._fp8_calibrated or T.fp8_calibrating())
        if self.replay and enc_mask_dict is None and n_accum_steps == 1 and not calibrating:
            return self._replay_step(image_dict, enc_mask_ratio, update_grad)
        if self.hip_graph and enc_mask_dict is None and n_accum_steps == 1 and update_grad:
            return self._graph_step(image_dict, enc_mask_ratio)
        loss, _, _, metrics = self.model(image_dict, enc_mask_ratio, enc_mask_dict=enc_mask_dict)
        if self.sync is not None:
            self.sync.arm(update_grad)  # on the micro-step that ends with the optimiser update, blocks all-reduce as their gradients complete
        (loss / n_accum_steps if n_accum_steps > 1 else loss).backward()
        T.fp8_step_end()  # fp8 weight-gradient path: this step's recorded maxima become the next step's scales (no-op otherwise)
        if T.FP8_FORWARD and T.FP8_WGRAD:
            self._fp8_calibrated = True
        grad_norm = None
        if update_grad:
            if self.sync is not None:
                self.sync.all_reduce()
            grad_norm = self._optimizer.step(self.clip_grad)
            self._optimizer.zero_grad()
            self._updated()
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
        T.fp8_step_end()
        grad_norm = None
        if update_grad:
            if self.sync is not None:
                self.sync.all_reduce()
            grad_norm = self._optimizer.step(self.clip_grad)
            self._optimizer.zero_grad()
            self._updated()
        return loss, grad_norm, metrics  # static tensors: overwritten by the next step

    # ------------------------------------------------------------------------------------------------ HIP-graph step
    def _capture(self, image_dict: dict, enc_mask_ratio: float) -> tuple:
        if T.FP8_FORWARD and T.FP8_WGRAD:
            # the delayed-scaling sites need a host call per step (fp8_step_end: maxima -> next step's scales) and a calibration step that runs other
            # kernels than the steady state; a captured graph would replay frozen scales and saturate as magnitudes drift
            raise ValueError("hip_graph=True cannot be combined with the fp8 weight-gradient path (CINEMA_FP8=1, CINEMA_FP8_WGRAD=1): use replay=True")
        static = {k: v.clone() for k, v in image_dict.items()}
        self._optimizer.zero_grad()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up on a non-default stream: per-stream workspaces, allocator pools, neighbour lists
            for _ in range(2):
                loss, _, _, _ = self.model(static, enc_mask_ratio)
                loss.backward()
            self._optimizer.zero_grad()
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
        grad_norm = self._optimizer.step(self.clip_grad)
        self._optimizer.zero_grad()
        self._updated()
        return loss, grad_norm, metrics  # static output buffers: overwritten by the next replay


# ---------------------------------------------------------------------------------------------------------------------
# Harness pieces of the reference's ``cinema/optim.py`` that callers of the path import: same names, arguments and
# return values, so ``pretrain_one_epoch`` / ``train_one_epoch`` style loops (``cinema/mae/pretrain.py:242-269``, ``cinema/train.py:85-168``)
# run unchanged on either a ``torch.optim`` optimiser or the fused flat-buffer optimiser above.
# ---------------------------------------------------------------------------------------------------------------------
def get_grad_norm(parameters, norm_type: float = 2.0) -> torch.Tensor:  # noqa: ANN001
    """Global gradient norm without clipping (reference ``cinema/optim.py:146-170``)."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    parameters = [p for p in parameters if p.grad is not None]
    norm_type = float(norm_type)
    if len(parameters) == 0:
        return torch.tensor(0.0)
    device = parameters[0].grad.device
    if norm_type == math.inf:
        return max(p.grad.detach().abs().max().to(device) for p in parameters)
    return torch.norm(torch.stack([torch.norm(p.grad.detach(), norm_type).to(device) for p in parameters]), norm_type)


class GradScaler:
    """``loss_scaler(loss, optimizer, clip_grad, parameters, update_grad=...) -> grad_norm`` (reference ``cinema/optim.py:173-226``).

    * ``optimizer`` is a :class:`FusedAdamW`: ``loss.backward()`` accumulates straight into the flat gradient buffer (the model's single
      autograd node), then - on ``update_grad`` - the data-parallel mean all-reduce of that buffer (when the model carries a
      ``GradientSynchronizer``, see ``cinema_amd.ddp.setup_ddp_model``), squared norm, clip coefficient and fused AdamW kernels.  bf16 has
      fp32's exponent range, so there is no loss scale to maintain; the inf/NaN skip of ``torch.GradScaler.step`` is the device-side
      non-finite guard of :meth:`FusedAdamW.step`.
    * any other ``torch.optim`` optimiser: the reference's own sequence on ``torch.GradScaler`` (scale -> backward -> unscale_ ->
      ``clip_grad_norm_`` / ``get_grad_norm`` -> step -> update), with the gradient mean all-reduce of DDP done here when a process group
      of more than one rank is up (the models are not wrapped in ``DistributedDataParallel``).
    """

    state_dict_key = "amp_scaler"

    def __init__(self) -> None:
        self._scaler = torch.GradScaler("cuda", enabled=torch.cuda.is_available())

    def __call__(self, loss: torch.Tensor, optimizer, clip_grad: float | None = None, parameters=None, create_graph: bool = False,  # noqa: ANN001
                 update_grad: bool = True):  # noqa: ANN204
        fused = isinstance(optimizer, FusedAdamW)
        sync = getattr(optimizer, "synchronizer", None) if fused else None
        if sync is not None:
            sync.arm(update_grad)
        if fused:
            loss.backward(create_graph=create_graph)
            T.fp8_step_end()  # fp8 weight-gradient path: this step's recorded maxima become the next step's scales (no-op otherwise), as in TrainStep
        else:
            self._scaler.scale(loss).backward(create_graph=create_graph)
        if not update_grad:
            return None
        if parameters is None:
            raise ValueError("parameters must not be None.")
        if fused:
            if sync is not None:
                sync.all_reduce()
            return optimizer.step(clip_grad)
        parameters = list(parameters) if not isinstance(parameters, torch.Tensor) else [parameters]
        _all_reduce_param_grads(parameters)
        self._scaler.unscale_(optimizer)
        norm = torch.nn.utils.clip_grad_norm_(parameters, clip_grad) if clip_grad is not None else get_grad_norm(parameters)
        self._scaler.step(optimizer)
        self._scaler.update()
        return norm

    def state_dict(self) -> dict:
        return self._scaler.state_dict()

    def load_state_dict(self, state_dict: dict) -> None:
        self._scaler.load_state_dict(state_dict)


def _all_reduce_param_grads(parameters: list) -> None:
    """DDP's gradient averaging for the torch-optimiser path (one flattened bucket; the fused path all-reduces its flat buffer instead)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() <= 1:
        return
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def _check_reductions_on_every_rank(sync=None) -> None:  # noqa: ANN001
    """``hip.check_reduction_workspaces`` as a COLLECTIVE decision under data parallelism: a rank that raised alone would leave the others waiting in their next
    gradient collective, so the flag is MAX-reduced over the ranks first (like ``GradientSynchronizer.all_finite``) and every rank raises together.  One 4-byte
    device -> host read every ``check_every`` updates (the only host synchronisation of the step loop) plus, with more than one rank, one scalar all-reduce."""
    err = None
    try:
        K.check_reduction_workspaces()
    except K.HipLibraryError as e:
        err = e
    world = getattr(sync, "world_size", 1) if sync is not None else 1
    if world > 1:
        import torch.distributed as dist

        dev = "cuda" if dist.get_backend(sync.group) == "nccl" else "cpu"
        flag = torch.tensor([1.0 if err is not None else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=sync.group)
        if err is None and float(flag) > 0:
            err = K.HipLibraryError("an in-launch split reduction gave up on ANOTHER rank (error word set there): its gradients were wrong and have been averaged "
                                    "into this rank's - restart from the last checkpoint.")
    if err is not None:
        raise err


def save_checkpoint(ckpt_dir, epoch: int, model_wo_ddp: nn.Module, optimizer, loss_scaler: GradScaler, n_samples: int):  # noqa: ANN001, ANN201
    """``ckpt_dir / f"ckpt_{epoch}.pt"`` with the reference's keys (``cinema/optim.py:229-261``): model, optimizer, epoch, scaler, n_samples."""
    from pathlib import Path

    if torch.cuda.is_available():
        # never write a checkpoint behind a split reduction that gave up (raises HipLibraryError - on every rank when the optimiser carries a synchronizer)
        _check_reductions_on_every_rank(getattr(optimizer, "synchronizer", None))
    ckpt_dir = Path(ckpt_dir)
    ckpt_dir.mkdir(parents=True, exist_ok=True)
    ckpt_path = ckpt_dir / f"ckpt_{epoch}.pt"
    torch.save({"model": model_wo_ddp.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch, "scaler": loss_scaler.state_dict(),
                "n_samples": n_samples}, ckpt_path)
    return ckpt_path


def load_checkpoint_and_optimizer(ckpt_path, model_wo_ddp: nn.Module, optimizer, loss_scaler: GradScaler) -> tuple:  # noqa: ANN001
    """-> (model, optimizer, loss_scaler, epoch, n_samples) (reference ``cinema/optim.py:264-294``).  For a :class:`FusedAdamW` the masters
    are loaded in place into the flat buffer and the bf16 weight shadows are re-derived."""
    ckpt = torch.load(ckpt_path, map_location="cpu")
    model_wo_ddp.load_state_dict(ckpt["model"])
    optimizer.load_state_dict(ckpt["optimizer"])
    if isinstance(optimizer, FusedAdamW):
        optimizer.flat.refresh_shadows()
        T.WEIGHTS.invalidate()
    loss_scaler.load_state_dict(ckpt["scaler"])
    return model_wo_ddp, optimizer, loss_scaler, ckpt["epoch"], ckpt.get("n_samples", 0)


class EarlyStopping:
    """Patience counter on a metric that should decrease (reference ``cinema/optim.py:297-330``)."""

    def __init__(self, min_delta: float, patience: int) -> None:
        self.min_delta = min_delta
        self.best_metric = float("inf")
        self.patience = patience
        self.patience_count = 0
        self.should_stop = False
        self.has_improved = False

    def update(self, metric: float) -> None:
        self.has_improved = self.best_metric > metric  # not necessarily improved enough
        if self.has_improved and self.best_metric >= metric + self.min_delta:
            self.best_metric = metric
            self.patience_count = 0
        else:
            self.patience_count += 1
            self.should_stop = self.patience_count >= self.patience
