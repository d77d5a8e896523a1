"""Loss and metric kernels (csrc/loss.hip): MAE patch MSE, segmentation CE + Dice, classification / regression heads, sliding-window accumulation, surface distances.

Part of the ctypes front of ``libcinema_hip.so`` (see ``cinema_amd/hip/__init__.py`` for the loader, the launch recorder, lane groups and the per-stream
workspaces); everything here is re-exported there, so callers keep writing ``hip.<name>``.  Module-level switches and the recorder state live in the package and
are read through it (``H.<NAME>``) so that an assignment ``hip.<NAME> = ...`` is seen by every family."""
from __future__ import annotations

import ctypes as C

import torch

from cinema_amd import hip as H
from cinema_amd.hip import (  # noqa: F401
    HipLibraryError, PatchGeom, _DT, _check, _dev, _empty, _empty_like, _p, _rowmajor, _stream, load,
)

__all__ = ['head_ce', 'head_mse', 'mask_edges', 'mean_finite', 'min_dist', 'mse_bwd', 'mse_fwd', 'patch_stats', 'seg_loss_bwd', 'seg_loss_fwd', 'seg_metric_counts', 'seg_window_accumulate', 'seg_window_finish']


def seg_loss_fwd(logits_rows: torch.Tensor, labels: torch.Tensor, batch: int):  # noqa: ANN201
    """-> (out4 = [loss, cross entropy, mean dice loss, 1/count], coef) for fp32 rows [batch*vox, c] and int32 labels [batch*vox]."""
    _dev(logits_rows, labels)
    if logits_rows.dtype != torch.float32 or labels.dtype != torch.int32 or not logits_rows.is_contiguous() or not labels.is_contiguous():
        raise HipLibraryError("seg_loss: logits fp32 rows and int32 labels, both contiguous")
    rows, c = logits_rows.shape
    acc = _empty(batch * c * 3 + 2, dtype=torch.float32, device=logits_rows.device)
    out4 = _empty(4, dtype=torch.float32, device=logits_rows.device)
    coef = _empty(batch * c * 2, dtype=torch.float32, device=logits_rows.device)
    _check(load().cinema_seg_loss_fwd(logits_rows.data_ptr(), labels.data_ptr(), batch, rows // batch, c, acc.data_ptr(), out4.data_ptr(), coef.data_ptr(),
                                      _stream()), "seg_loss_fwd")
    return out4, coef


def seg_loss_bwd(logits_rows: torch.Tensor, labels: torch.Tensor, batch: int, coef: torch.Tensor, out4: torch.Tensor, upstream: torch.Tensor | None):  # noqa: ANN201
    _dev(logits_rows, labels, coef, out4, upstream)
    rows, c = logits_rows.shape
    d = _empty_like(logits_rows)
    _check(load().cinema_seg_loss_bwd(logits_rows.data_ptr(), labels.data_ptr(), batch, rows // batch, c, coef.data_ptr(), out4.data_ptr(), _p(upstream),
                                      d.data_ptr(), _stream()), "seg_loss_bwd")
    return d


def head_ce(logits: torch.Tensor, labels: torch.Tensor, label_smoothing: float = 0.0) -> tuple:
    """Mean cross entropy with label smoothing of fp32 logits [b, c] against int32 labels [b] -> (loss [1], d loss / d logits [b, c])."""
    _dev(logits, labels)
    if logits.dtype != torch.float32 or labels.dtype != torch.int32 or logits.dim() != 2 or labels.numel() != logits.shape[0] or \
            not logits.is_contiguous() or not labels.is_contiguous():
        raise HipLibraryError("head_ce: contiguous fp32 logits [b, c] and int32 labels [b]")
    out, d = _empty(1, dtype=torch.float32, device=logits.device), _empty_like(logits)
    _check(load().cinema_head_ce(logits.data_ptr(), labels.data_ptr(), logits.shape[0], logits.shape[1], float(label_smoothing), out.data_ptr(), d.data_ptr(),
                                 _stream()), "head_ce")
    return out, d


def head_mse(pred: torch.Tensor, label: torch.Tensor) -> tuple:
    """-> (out6 = [mse, mae, max label, min label, max pred, min pred], d mse / d pred) for contiguous fp32 tensors of one shape."""
    _dev(pred, label)
    if pred.dtype != torch.float32 or label.dtype != torch.float32 or pred.shape != label.shape or not pred.is_contiguous() or not label.is_contiguous():
        raise HipLibraryError("head_mse: contiguous fp32 predictions and labels of one shape")
    out, d = _empty(6, dtype=torch.float32, device=pred.device), _empty_like(pred)
    _check(load().cinema_head_mse(pred.data_ptr(), label.data_ptr(), pred.numel(), out.data_ptr(), d.data_ptr(), _stream()), "head_mse")
    return out, d


def seg_window_accumulate(window_rows: torch.Tensor, patch: tuple, start: tuple, size: tuple, prob_sum: torch.Tensor, count: torch.Tensor) -> None:
    """Add softmax(window_rows) (fp32 [prod(patch), c], channels last) into prob_sum [prod(size), c] / count [prod(size)] at offset ``start``
    (2-D windows use a leading unit axis)."""
    _dev(window_rows, prob_sum, count)
    if window_rows.dtype != torch.float32 or not window_rows.is_contiguous() or prob_sum.dtype != torch.float32 or count.dtype != torch.float32:
        raise HipLibraryError("seg_window_accumulate: contiguous fp32 tensors")
    p3, z3 = [(1,) * (3 - len(t)) + tuple(int(v) for v in t) for t in (patch, size)]
    s3 = (0,) * (3 - len(start)) + tuple(int(v) for v in start)
    _check(load().cinema_seg_window_accumulate(window_rows.data_ptr(), window_rows.shape[1], *p3, *s3, *z3, prob_sum.data_ptr(), count.data_ptr(), _stream()),
           "seg_window_accumulate")


def seg_window_finish(prob_sum: torch.Tensor, count: torch.Tensor) -> torch.Tensor:
    """-> fp32 [c, n_voxels] = log(prob_sum / count) (channels first)."""
    _dev(prob_sum, count)
    n, c = prob_sum.shape
    out = _empty((c, n), dtype=torch.float32, device=prob_sum.device)
    _check(load().cinema_seg_window_finish(prob_sum.data_ptr(), count.data_ptr(), c, n, out.data_ptr(), _stream()), "seg_window_finish")
    return out


def seg_metric_counts(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """logits fp32 (b, c, *spatial) contiguous, labels int32 (b, *spatial) -> int32 [b, c, 6] voxel counts (see ``cinema_seg_metric_counts``)."""
    _dev(logits, labels)
    if logits.dtype != torch.float32 or labels.dtype != torch.int32 or not logits.is_contiguous() or not labels.is_contiguous():
        raise HipLibraryError("seg_metric_counts: contiguous fp32 logits (channels first) and int32 labels")
    b, c = logits.shape[0], logits.shape[1]
    vox = logits[0, 0].numel()
    counts = _empty((b, c, 6), dtype=torch.int32, device=logits.device)
    _check(load().cinema_seg_metric_counts(logits.data_ptr(), labels.data_ptr(), b, vox, c, counts.data_ptr(), _stream()), "seg_metric_counts")
    return counts


def mask_edges(label: torch.Tensor, n_classes: int) -> torch.Tensor:
    """label int32 (b, *spatial) with 2 or 3 spatial axes -> uint8 (b, n_classes, *spatial): surface voxels of every class (``cinema_mask_edges``)."""
    _dev(label)
    if label.dtype != torch.int32 or not label.is_contiguous() or label.dim() not in (3, 4):
        raise HipLibraryError("mask_edges: contiguous int32 label map (b, *spatial), 2 or 3 spatial axes")
    b, sp = label.shape[0], tuple(label.shape[1:])
    x, y, z = (1,) * (3 - len(sp)) + sp
    edges = _empty((b, n_classes, *sp), dtype=torch.uint8, device=label.device)
    _check(load().cinema_mask_edges(label.data_ptr(), b, x, y, z, n_classes, len(sp), edges.data_ptr(), _stream()), "mask_edges")
    return edges


def min_dist(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a fp32 [na, 3], b fp32 [nb, 3] (physical coordinates) -> fp32 [na]: distance of every a_i to the nearest point of b."""
    _dev(a, b)
    if a.dtype != torch.float32 or b.dtype != torch.float32 or not a.is_contiguous() or not b.is_contiguous() or a.shape[1:] != (3,) or b.shape[1:] != (3,):
        raise HipLibraryError("min_dist: contiguous fp32 [n, 3] point sets")
    out = _empty((a.shape[0],), dtype=torch.float32, device=a.device)
    _check(load().cinema_min_dist(a.data_ptr(), b.data_ptr(), a.shape[0], b.shape[0], out.data_ptr(), _stream()), "min_dist")
    return out


def mse_fwd(image: torch.Tensor, geom: PatchGeom, pred: torch.Tensor, norm_target: bool, eps: float, loss_out: torch.Tensor,
            max_out: torch.Tensor | None = None) -> None:
    _dev(image, pred, loss_out, max_out)
    feat = geom.px * geom.py * geom.pz * geom.c
    _check(load().cinema_mse_fwd(image.data_ptr(), C.byref(geom), pred.data_ptr(), _DT[pred.dtype], _rowmajor(pred, "pred"), int(norm_target), eps,
                                 1.0 / (geom.n_rows * feat), loss_out.data_ptr(), _p(max_out), _stream()), "mse_fwd")


def mse_bwd(image: torch.Tensor, geom: PatchGeom, pred: torch.Tensor, norm_target: bool, eps: float, upstream: torch.Tensor | None,
            host_scale: float) -> torch.Tensor:
    _dev(image, pred, upstream)
    dpred = _empty(pred.shape, dtype=torch.bfloat16, device=pred.device)
    _check(load().cinema_mse_bwd(image.data_ptr(), C.byref(geom), pred.data_ptr(), _DT[pred.dtype], _rowmajor(pred, "pred"), int(norm_target), eps,
                                 _p(upstream), host_scale, dpred.data_ptr(), dpred.stride(0), _stream()), "mse_bwd")
    return dpred


def patch_stats(image: torch.Tensor, geom_all: PatchGeom, out2: torch.Tensor) -> None:
    _dev(image, out2)
    _check(load().cinema_patch_stats(image.data_ptr(), C.byref(geom_all), out2.data_ptr(), _stream()), "patch_stats")


def mean_finite(vals: torch.Tensor, mean_out: torch.Tensor, coef_out: torch.Tensor | None) -> None:
    _dev(vals, mean_out, coef_out)
    _check(load().cinema_mean_finite(vals.data_ptr(), vals.numel(), mean_out.data_ptr(), _p(coef_out), _stream()), "mean_finite")
