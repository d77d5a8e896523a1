"""Tape ops that move rows: views, segment means, patch gathers, row splits, token assembly (position embedding + mask tokens), the MAE patch loss, the mean over the views' losses.

Part of the tape (``cinema_amd/tape/__init__.py`` holds :class:`Tape`, :class:`Var`, the weight caches, the weight-gradient streams and groups, the fp8 sites and the
autograd bridge); everything here is re-exported there, so callers keep writing ``tape.op_*``.  Module-level switches live in the package and are read through it
(``T.<NAME>``) so that an assignment ``tape.<NAME> = ...`` is seen here."""
from __future__ import annotations


import torch

from cinema_amd import hip as K
from cinema_amd import tape as T
from cinema_amd.tape import (  # noqa: F401
    BF16, F32, Tape, Var, _wgrad_launch, const, zeros,
)

__all__ = ['Segment', 'op_assemble', 'op_mean_finite', 'op_mse', 'op_patch_gather', 'op_scale', 'op_segment_mean', 'op_split_rows', 'op_view']


def op_view(tape: Tape, x: Var, shape: tuple) -> Var:
    """Zero-copy reshape of contiguous rows (e.g. [n*4, c] -> [n, 4*c]); the gradient is reshaped back."""
    y = Var(x.data.view(shape), needs_grad=x.needs_grad)

    def bwd() -> None:
        if y.grad is not None and x.needs_grad:
            x.add_grad(y.grad.reshape(x.data.shape), None if y.grad16 is None else y.grad16.reshape(x.data.shape))

    tape.record(bwd)
    return y


def op_segment_mean(tape: Tape, x: Var, n_seg: int) -> Var:
    """fp32 [n_seg * rows, c] -> [n_seg, c]: mean over each block of consecutive rows (token pooling, ``convvit.py:523-547``)."""
    seg_rows = x.data.shape[0] // n_seg
    y = Var(K.segment_mean(x.data, n_seg))

    def bwd() -> None:
        if y.grad is not None and x.needs_grad:
            x.add_grad(K.segment_mean_bwd(y.grad, seg_rows))

    tape.record(bwd)
    return y


def op_scale(tape: Tape, x: Var, alpha: float) -> Var:
    y = Var(K.scale(x.data, alpha))

    def bwd() -> None:
        if y.grad is not None and x.needs_grad:
            x.add_grad(K.scale(y.grad.contiguous(), alpha))

    tape.record(bwd)
    return y


def op_patch_gather(tape: Tape, x: Var, geom, dst_shape: tuple | None = None) -> Var:  # noqa: ANN001
    """rows[token, (patch, c)] (bf16) gathered from a volume described by ``geom``; backward scatters (zeros elsewhere)."""
    y = Var(K.patch_gather(x.data, geom, BF16))

    def bwd() -> None:
        if y.grad is None or not x.needs_grad:
            return
        subset = geom.token_idx is not None
        dx = zeros(x.data.shape, F32, x.data.device) if subset else K.empty(x.data.shape, dtype=F32, device=x.data.device)
        K.patch_scatter(y.grad, dx, geom)
        x.add_grad(dx)

    tape.record(bwd)
    return y


def op_split_rows(tape: Tape, x: Var, idx_list: list) -> list:
    """ys[i] = x[idx_list[i]] (int32 row indices, disjoint across the list).  One zeroed gradient buffer is shared; all the gathers (and,
    backward, all the scatters) go out as one multi-segment launch."""
    c = x.data.shape[1]
    outs = [K.empty((idx.numel(), c), dtype=x.data.dtype, device=x.data.device) for idx in idx_list]
    K.row_copy_multi([dict(dst=out, src=x.data, src_idx=idx) for out, idx in zip(outs, idx_list)])
    ys = [Var(out) for out in outs]

    def bwd() -> None:
        if not x.needs_grad or all(y.grad is None for y in ys):
            return
        dx = zeros(x.data.shape, x.data.dtype, x.data.device)
        K.row_copy_multi([dict(dst=dx, src=y.grad, dst_idx=idx) for y, idx in zip(ys, idx_list) if y.grad is not None])
        x.add_grad(dx)

    tape.record(bwd)
    return ys


class Segment:
    """One source of rows for :func:`op_assemble`: ``dst[dst_idx[i]] = src[i or src_idx[i]] + add[add_idx[i]]``.

    ``src`` is a :class:`Var` (rows [n, c]) or an ``nn.Parameter`` token of shape (1, 1, c) broadcast to every row.
    ``add`` is a constant table (frozen sin-cos positional embedding), indexed by ``add_idx``.
    """

    def __init__(self, dst_idx: torch.Tensor, src=None, add: torch.Tensor | None = None, add_idx: torch.Tensor | None = None,  # noqa: ANN001
                 grad_bf16: bool = False) -> None:
        self.dst_idx, self.src, self.add, self.add_idx = dst_idx, src, add, add_idx
        # the source's gradient is only ever read as a bf16 GEMM operand (an op_linear output with a single consumer): gather it as bf16, no cast pass later
        self.grad_bf16 = grad_bf16


def op_assemble(tape: Tape, n_rows: int, c: int, segments: list, device: torch.device) -> Var:
    """Build a token matrix [n_rows, c] (fp32) from row segments (replaces torch.cat / bool-mask selects / pos-embed adds:
    cinema/vit.py:672-674, cinema/mae/mae.py:98-104,580-585, cinema/convvit.py:205)."""
    out = K.empty((n_rows, c), dtype=F32, device=device)
    copies = []
    for s in segments:  # disjoint destination rows: one multi-segment launch
        n = s.dst_idx.numel()
        if isinstance(s.src, Var):
            copies.append(dict(dst=out, src=s.src.data, dst_idx=s.dst_idx, add=s.add, add_idx=s.add_idx))
        elif s.src is not None:  # broadcast token parameter
            z = const(("zero_idx", n, str(device)), lambda: torch.zeros(n, dtype=torch.int32, device=device))
            copies.append(dict(dst=out, src=s.src.detach().view(1, c), dst_idx=s.dst_idx, src_idx=z, add=s.add, add_idx=s.add_idx))
        else:
            copies.append(dict(dst=out, src=None, dst_idx=s.dst_idx, add=s.add, add_idx=s.add_idx))
    K.row_copy_multi(copies)
    y = Var(out)
    y.grad_any = True

    def bwd() -> None:
        if y.grad is None:
            return
        gathers, targets = [], []
        for s in segments:
            if isinstance(s.src, Var):
                if s.src.needs_grad:
                    g = K.empty((s.dst_idx.numel(), c), dtype=BF16 if (s.grad_bf16 and s.src.grad is None) else F32, device=device)
                    gathers.append(dict(dst=g, src=y.grad, src_idx=s.dst_idx))
                    targets.append((s.src, g))
            elif s.src is not None and s.src.requires_grad:
                buf, yg, idx = tape.pvar(s.src).grad_buffer((c,)), y.grad, s.dst_idx
                if yg.is_cuda:  # a leaf gradient: nothing in the backward chain waits for it
                    _wgrad_launch(lambda buf=buf, yg=yg, idx=idx: K.colsum(yg, buf, row_idx=idx), yg, keys=(buf.data_ptr(),))
                else:
                    K.colsum(yg, buf, row_idx=idx)
        K.row_copy_multi(gathers)
        for var, g in targets:
            var.add_grad(g)

    tape.record(bwd)
    return y


def op_mse(tape: Tape, pred: Var, image: torch.Tensor, geom_masked, norm_target: bool, eps: float = 1e-6) -> Var:  # noqa: ANN001
    """Scalar masked-patch MSE (cinema/mae/mae.py:140-143); the target patches are gathered from ``image`` on the fly."""
    loss = zeros(1, F32, pred.data.device)
    maxes = K.full((2,), float("-inf"), F32, pred.data.device) if norm_target else None
    K.mse_fwd(image, geom_masked, pred.data, norm_target, eps, loss, maxes)
    y = Var(loss)

    def bwd() -> None:
        if y.grad is None or not pred.needs_grad:
            return
        d = K.mse_bwd(image, geom_masked, pred.data, norm_target, eps, y.grad, 1.0 / pred.data.numel())
        # (the prediction head's backward reads the gradient as a bf16 GEMM operand only: no fp32 copy)
        pred.add_grad(d if (pred.data.dtype == BF16 or pred.grad is None) else K.cast(d, F32))

    tape.record(bwd)
    return y, maxes  # maxes: (normed_target_max, pred_max) metrics of the norm_target mode (mae.py:146-150), else None


def op_mean_finite(tape: Tape, losses: list) -> Var:
    """Mean over the finite per-view losses (cinema/mae/mae.py:604-608) without a host round trip."""
    dev = losses[0].data.device
    vals = K.empty(len(losses), dtype=F32, device=dev)
    K.row_copy_multi([dict(dst=vals[i:i + 1].view(1, 1), src=lv.data.view(1, 1)) for i, lv in enumerate(losses)])
    mean, coef = K.empty(1, dtype=F32, device=dev), K.empty(len(losses), dtype=F32, device=dev)
    K.mean_finite(vals, mean, coef)
    y = Var(mean)

    def bwd() -> None:
        if y.grad is None:
            return
        gs = K.mul_scalar(coef, y.grad.reshape(1))  # d loss / d loss_i = coef[i] * upstream, all views in one launch
        for i, lv in enumerate(losses):
            lv.add_grad(gs[i:i + 1])

    tape.record(bwd)
    return y
