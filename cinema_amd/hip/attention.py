"""Flash-style attention forward / backward (csrc/attention.hip).

Part of the ctypes front of ``libcinema_hip.so`` (see ``cinema_amd/hip/__init__.py`` for the loader, the launch recorder, lane groups and the per-stream
workspaces); everything here is re-exported there, so callers keep writing ``hip.<name>``.  Module-level switches and the recorder state live in the package and
are read through it (``H.<NAME>``) so that an assignment ``hip.<NAME> = ...`` is seen by every family."""
from __future__ import annotations

import ctypes as C
import os

import torch

from cinema_amd import hip as H
from cinema_amd.hip import (  # noqa: F401
    HipLibraryError, _check, _dev, _empty, _p, _stream, _workspace, load, persistent,
)

__all__ = ['ATTN_COUNTERS', '_ATTN_COUNTERS', '_attn_counters', '_attn_view', 'attention_bwd', 'attention_fwd']


def _attn_view(t: torch.Tensor, heads: int, hd: int, name: str):  # noqa: ANN202
    """t: [b, tokens, >= heads*hd] view with unit inner stride and batch stride = tokens*row stride."""
    if t.dim() != 3 or t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1) or t.shape[2] != heads * hd or t.dtype != torch.bfloat16:
        raise HipLibraryError(f"{name}: expected bf16 [b, t, heads*hd] view with packed batch stride, got {tuple(t.shape)} {t.stride()}")
    return t.data_ptr(), t.stride(1)


def attention_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float, force_generic: bool = False, want_lo: bool = False):  # noqa: ANN201
    """q: [b,tq,C], k/v: [b,tk,C] bf16 views (C = heads*hd) -> (o [b,tq,C] bf16, lse [b,heads,tq] fp32 log2-domain); ``want_lo``: -> (o, lse, o_lo) with
    o_lo = bf16(O - float(o)), the second half of the output that :func:`attention_bwd` adds when it forms delta = rowsum(dO O) (training)."""
    _dev(q, k, v)
    b, tq, cdim = q.shape
    tk, hd = k.shape[1], cdim // heads
    (qp, ldq), (kp, ldk), (vp, ldv) = _attn_view(q, heads, hd, "q"), _attn_view(k, heads, hd, "k"), _attn_view(v, heads, hd, "v")
    o = _empty((b, tq, cdim), dtype=torch.bfloat16, device=q.device)
    o_lo = _empty((b, tq, cdim), dtype=torch.bfloat16, device=q.device) if want_lo else None
    lse = _empty((b, heads, tq), dtype=torch.float32, device=q.device)
    _check(load().cinema_attention_fwd(qp, ldq, kp, ldk, vp, ldv, o.data_ptr(), _p(o_lo), cdim, lse.data_ptr(), b, heads, tq, tk, hd, scale,
                                       int(force_generic or H.FORCE_GENERIC), _stream()), "attention_fwd")
    return (o, lse, o_lo) if want_lo else (o, lse)


def attention_bwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor, d_o: torch.Tensor, lse: torch.Tensor, heads: int,
                  scale: float, dq: torch.Tensor, dk: torch.Tensor, dv: torch.Tensor, force_generic: bool = False, o_lo: torch.Tensor | None = None) -> None:
    """Writes dq/dk/dv (bf16 views with the same addressing rules as q/k/v).  ``o_lo``: the second half of the forward output (``attention_fwd(want_lo=True)``)."""
    _dev(q, k, v, o, d_o, lse, dq, dk, dv, o_lo)
    b, tq, cdim = q.shape
    tk, hd = k.shape[1], cdim // heads
    ptrs = [_attn_view(t, heads, hd, n) for t, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o"), (d_o, "d_o"), (dq, "dq"), (dk, "dk"), (dv, "dv"))]
    if o_lo is not None and (o_lo.shape != o.shape or o_lo.stride() != o.stride() or o_lo.dtype != torch.bfloat16):
        raise HipLibraryError("attention_bwd: o_lo must have the layout of o")
    delta = _empty((b, heads, tq), dtype=torch.float32, device=q.device)
    args = (ptrs[0][0], ptrs[0][1], ptrs[1][0], ptrs[1][1], ptrs[2][0], ptrs[2][1], ptrs[3][0], _p(o_lo), ptrs[3][1], ptrs[4][0], ptrs[4][1], lse.data_ptr(),
            delta.data_ptr(), ptrs[5][0], ptrs[5][1], ptrs[6][0], ptrs[6][1], ptrs[7][0], ptrs[7][1], b, heads, tq, tk, hd, scale, int(force_generic or H.FORCE_GENERIC))
    if not (hd == 64 and os.environ.get("CINEMA_ATTN_ONEPASS", "0") == "1"):  # the dQ + dK/dV kernel pair / the one-pass kernel at head_dim 32: no scratch, no
        # counters (the library reads the same variable per call; the head_dim 64 one-pass form is an option, off by default)
        _check(load().cinema_attention_bwd(*args, _stream()), "attention_bwd")
        return
    # scratch of the one-pass backward at head_dim 64 (csrc/attention.hip attn_bwd_onepass_mfma): running dQ sums across the passes over the keys / the sums of
    # the workgroups that share a (batch, head) pair, and their arrival tickets (zero at allocation, left zero by every launch)
    ws_bytes = load().cinema_attention_bwd_workspace_bytes(b, heads, tq, tk, hd)
    ws = _workspace("attn_bwd", ws_bytes // 4, q.device) if ws_bytes > 0 else None
    cnt = _attn_counters(q.device) if b * heads <= ATTN_COUNTERS else None
    _check(load().cinema_attention_bwd_ws(*args, None if ws is None else ws.data_ptr(), ws_bytes if ws is not None else 0,
                                          None if cnt is None else cnt.data_ptr(), 0 if cnt is None else cnt.numel(), _stream()), "attention_bwd")


ATTN_COUNTERS = 16384
_ATTN_COUNTERS: dict = {}


def _attn_counters(device: torch.device) -> torch.Tensor:
    """Arrival tickets of the key-split one-pass attention backward, one buffer per (device, stream, lane): zero at allocation, left zero by every launch,
    never handed back to the allocator (``persistent``)."""
    key = (device.index, _stream(), H.LANE)
    t = _ATTN_COUNTERS.get(key)
    if t is None:
        t = _ATTN_COUNTERS[key] = persistent(lambda: torch.zeros(ATTN_COUNTERS, dtype=torch.int32, device=device))
    return t
