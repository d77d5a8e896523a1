"""Recorded training steps: forward + backward of the MAE step as a flat list of HIP launches, re-issued without the Python model code.

Why: one optimisation step is ~2000 kernel launches.  Issued from the module code (tape closures, ctypes wrappers, torch.empty, views)
the host needs ~33 ms for them on a fast EPYC and ~45 ms on a slow one - as long as the GPU needs for the kernels themselves (~33 ms at
per-GPU batch 16), so the step was launch-bound on slow hosts and at small batches (batch 4: 33.7 ms for a quarter of the device work).
The launches of this library are plain C calls with ints, floats and struct pointers as arguments (``include/cinema_hip.h``), so one eager
step under ``hip.RECORD`` yields the exact call list; re-issuing it costs ~3 us per launch (`tools/launch_rate.py`).  HIP graphs
(``hipGraphLaunch``) were measured first and are SLOWER than the eager Python path on ROCm 7.2 (37.6 vs 35.0 ms, `tools/graph_try.py`):
the runtime walks the nodes on the host at launch time; this list is the same idea without that cost.

What makes the list valid on later steps:
  * every tensor allocated during the recording comes from a private ``torch.cuda.MemPool`` that lives as long as the recording, so the
    addresses in the list stay reserved (the allocator's reuse of freed blocks INSIDE the step is replayed too, exactly as in the eager
    order; side-stream operands are held until the recorded join);
  * device work done by torch ops is not in the list: the model code routes it through ``tape.host`` (mask-dependent index tensors:
    re-run on every replay, results copied into the recorded tensors), ``tape.const`` (shape-only tables) and ``hip.zeros/full`` (fills
    as launches).  ``audit=True`` records under a TorchDispatchMode that reports any other ATen kernel inside the step;
  * inputs and random masks are written into static tensors before each replay (masks: the model's own recipe and RNG consumption);
  * clip + AdamW stay eager (their scalars change every step), as does the data-parallel gradient exchange, which enters the list as
    host entries at the points where the eager backward fires its hooks.
"""
from __future__ import annotations

import threading

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from cinema_amd import hip as K
from cinema_amd import tape as T

# ATen ops that do not launch device work (views, metadata, allocation)
_VIEW_OPS = ("view", "reshape", "_unsafe_view", "slice", "select", "detach", "as_strided", "unsqueeze", "squeeze", "expand", "permute", "transpose", "t",
             "alias", "empty", "empty_like", "empty_strided", "_reshape_alias", "unbind", "split", "split_with_sizes", "movedim", "narrow", "unfold",
             "is_same_size", "sym_size", "sym_stride", "sym_numel", "lift_fresh", "_local_scalar_dense", "flatten", "view_as", "resolve_conj", "resolve_neg",
             "is_nonzero", "item", "contiguous", "_to_copy.noop")

_tls = threading.local()


class _Audit(TorchDispatchMode):
    """Collects ATen ops with device tensors that run inside a recorded step outside tape.host / tape.const."""

    def __init__(self, sink: list) -> None:
        super().__init__()
        self.sink = sink

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):  # noqa: ANN001, ANN204
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name not in _VIEW_OPS and not getattr(_tls, "allowed", 0):
            flat = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
            outs = [o for o in (out if isinstance(out, (tuple, list)) else [out]) if isinstance(o, torch.Tensor)]
            if any(t.is_cuda for t in flat + outs):
                import traceback
                frames = [f for f in traceback.extract_stack() if "/cinema_amd/" in f.filename and "replay.py" not in f.filename]
                where = f"{frames[-1].filename.split('/cinema_amd/')[-1]}:{frames[-1].lineno}" if frames else "?"
                self.sink.append(f"{func.__name__} at {where}")
        return out


class allowed_aten:  # noqa: N801
    """Context for torch ops that are accounted for (inside tape.host / tape.const, or one-time setup)."""

    def __enter__(self) -> None:
        _tls.allowed = getattr(_tls, "allowed", 0) + 1

    def __exit__(self, *exc) -> None:  # noqa: ANN002
        _tls.allowed -= 1


class RecordedStep:
    """forward + backward for one input signature.  ``run(image_dict)`` -> (loss, metrics): static tensors, overwritten by the next run."""

    def __init__(self, model: torch.nn.Module, image_dict: dict, enc_mask_ratio: float, audit: bool = False) -> None:
        self.model, self.ratio = model, enc_mask_ratio
        # static inputs in the layout the recorded launches read: CineMA.forward converts with .float().contiguous() (mae.py), which is a
        # no-op on these and an ATen copy OUTSIDE the launch list on anything else (the replays would train on the first batch for ever);
        # run() copies every new batch into these tensors, converting dtype / strides on the way
        self.images = {k: v.detach().float().contiguous().clone() for k, v in image_dict.items()}
        masks, self.n_masked = model.draw_masks(self.images, enc_mask_ratio)
        self.masks = {k: m.clone() for k, m in masks.items()}
        self.pool = torch.cuda.MemPool()
        self.unaccounted: list = []
        T.WEIGHTS.invalidate()  # every re-laid-out weight shadow is rebuilt inside the recording (its launch must be in the list)
        calls: list = []
        K.RECORD = calls
        try:
            with torch.cuda.use_mem_pool(self.pool):
                if audit:
                    with _Audit(self.unaccounted):
                        self._eager_step()
                else:
                    self._eager_step()
        finally:
            K.RECORD, T.REC_CALL = None, None
        self.calls = calls
        self.n_launches = sum(1 for fn, _ in calls if fn is not None)

    def _draw_into_static(self) -> None:
        masks, _ = self.model.draw_masks(self.images, self.ratio)
        for k, m in masks.items():
            self.masks[k].copy_(m)

    def _eager_step(self) -> None:
        loss, _, _, metrics = self.model(self.images, self.ratio, enc_mask_dict=self.masks, n_masked=self.n_masked)
        seed = K.full((1,), 1.0, torch.float32, loss.device)  # d loss / d loss as a recorded fill
        T.REC_CALL.backward(seed)  # on this thread (inside the memory pool), not through the autograd engine
        self.loss, self.metrics = loss.detach(), {k: v.detach() for k, v in metrics.items()}

    def run(self, image_dict: dict):  # noqa: ANN201
        for k, v in image_dict.items():
            if v.data_ptr() != self.images[k].data_ptr():
                self.images[k].copy_(v, non_blocking=True)
        self._draw_into_static()
        for fn, args in self.calls:
            if fn is None:
                args()  # host entry: mask-dependent index tensors, gradient-exchange hooks
            else:
                rc = fn(*args)
                if rc != 0:
                    raise K.HipLibraryError(f"replayed launch {fn.__name__} failed: {rc}")
        return self.loss, self.metrics


class RecordedSegStep:
    """forward + CE / Dice loss + backward of the segmentation fine-tuning step for one input signature, as a launch list (see the module docstring).
    The model is entered through ``forward_rows`` and the loss kernels read / write channels-last rows, so nothing but library launches sits between
    the inputs and the flat gradient buffer; dropout / drop-path masks change per replay because the launch that advances the device RNG step is part
    of the list.  ``run(batch)`` -> (loss, metrics) with the keys of ``segmentation_loss`` (static tensors, overwritten by the next run)."""

    def __init__(self, model: torch.nn.Module, views: list, batch: dict, audit: bool = False) -> None:
        self.model, self.views = model, list(views)
        dev = next(model.parameters()).device
        self.images = {v: batch[f"{v}_image"].detach().to(dev).float().contiguous().clone() for v in self.views}
        self.labels = {v: batch[f"{v}_label"].detach().to(dev).reshape(-1).to(torch.int32).contiguous().clone() for v in self.views}
        self.batch = next(iter(self.images.values())).shape[0]
        self.pool = torch.cuda.MemPool()
        self.unaccounted: list = []
        T.WEIGHTS.invalidate()
        calls: list = []
        K.RECORD = calls
        try:
            with torch.cuda.use_mem_pool(self.pool):
                if audit:
                    with _Audit(self.unaccounted):
                        self._eager_step()
                else:
                    self._eager_step()
        finally:
            K.RECORD, T.REC_CALL = None, None
        self.calls = calls
        self.n_launches = sum(1 for fn, _ in calls if fn is not None)
        self.loss, self.metrics = self._collect()

    def _eager_step(self) -> None:
        rows = self.model.forward_rows(self.images)
        up = K.full((1,), 1.0 / len(self.views), torch.float32, self.labels[self.views[0]].device)  # d (mean over views) / d loss_v as a recorded fill
        self.out4, grads = {}, []
        for v in self.views:
            out4, coef = K.seg_loss_fwd(rows[v], self.labels[v], self.batch)
            grads.append(K.seg_loss_bwd(rows[v], self.labels[v], self.batch, coef, out4, up))
            self.out4[v] = out4
        T.REC_CALL.backward(*grads)  # on this thread (inside the memory pool), not through the autograd engine

    def _collect(self) -> tuple:
        """The metric dict of ``segmentation_loss_tensors`` from the loss kernels' result vectors (a few 0-d ops per step, outside the list)."""
        with allowed_aten():
            metrics, n = {}, len(self.views)
            for v in self.views:
                o = self.out4[v]
                metrics.update({f"{v}_cross_entropy": o[1], f"{v}_mean_dice_loss": o[2], f"{v}_loss": o[0], f"{v}_{v}_loss": o[0]})
            loss = sum(self.out4[v][0] for v in self.views) / n
            metrics["loss"] = loss
            for k in ("cross_entropy", "mean_dice_loss", "loss"):
                if k != "loss":
                    metrics[k] = sum(metrics[f"{v}_{k}"] for v in self.views) / n
            return loss, metrics

    def run(self, batch: dict):  # noqa: ANN201
        for v in self.views:
            self.images[v].copy_(batch[f"{v}_image"], non_blocking=True)
            self.labels[v].copy_(batch[f"{v}_label"].reshape(-1), non_blocking=True)
        for fn, args in self.calls:
            if fn is None:
                args()
            else:
                rc = fn(*args)
                if rc != 0:
                    raise K.HipLibraryError(f"replayed launch {fn.__name__} failed: {rc}")
        self.loss, self.metrics = self._collect()
        return self.loss, self.metrics
