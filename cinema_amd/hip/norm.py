"""LayerNorm kernels (csrc/norm.hip).

Part of the ctypes front of ``libcinema_hip.so`` (see ``cinema_amd/hip/__init__.py`` for the loader, the launch recorder, lane groups and the per-stream
workspaces); everything here is re-exported there, so callers keep writing ``hip.<name>``.  Module-level switches and the recorder state live in the package and
are read through it (``H.<NAME>``) so that an assignment ``hip.<NAME> = ...`` is seen by every family."""
from __future__ import annotations

import ctypes as C

import torch

from cinema_amd import hip as H
from cinema_amd.hip import (  # noqa: F401
    HipLibraryError, LnReduceItem, Q8Site, _check, _dev, _empty, _p, _rowmajor, _stream, _workspace, load,
)

__all__ = ['layernorm_bwd', 'layernorm_fwd', 'ln_param_reduce_batched']


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, *, act: int = 0, want_bf16: bool = True,
                  want_f32: bool = False, want_fp8: bool = False, q8: Q8Site | None = None):  # noqa: ANN201
    """x: [rows, c] fp32/bf16 -> (y_bf16 | None, y_f32 | None, mean, rstd) [+ (y_fp8 uint8 [rows, c], row_scale fp32 [rows]) with ``want_fp8``; with ``q8`` (a
    site with a scale) the copy uses the site's per-tensor delayed scale: (y_fp8, site.scale); a site without a scale yet only records the maximum and the
    per-row copy is returned]."""
    _dev(x, gamma, beta)
    rows, c = x.shape
    y16 = _empty((rows, c), dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    y32 = _empty((rows, c), dtype=torch.float32, device=x.device) if want_f32 else None
    mean = _empty(rows, dtype=torch.float32, device=x.device)
    rstd = _empty(rows, dtype=torch.float32, device=x.device)
    if q8 is not None and q8.ready:
        y8 = _empty((rows, c), dtype=torch.uint8, device=x.device)
        q = q8.out(y8)
        _check(load().cinema_layernorm_fwd_q8(x.data_ptr(), int(x.dtype == torch.bfloat16), _rowmajor(x, "x"), gamma.data_ptr(), beta.data_ptr(), rows, c, eps, act,
                                              _p(y16), _p(y32), c, mean.data_ptr(), rstd.data_ptr(), C.byref(q), _stream()), "layernorm_fwd_q8")
        return y16, y32, mean, rstd, (y8, q8.scale)
    if q8 is not None:  # calibration: the maximum of y16 through the stand-alone recorder (one extra pass, first step only)
        out = layernorm_fwd(x, gamma, beta, eps, act=act, want_bf16=True, want_f32=want_f32, want_fp8=want_fp8)
        H.quantize_fp8_site(out[0], q8)
        return out
    if want_fp8:
        y8 = _empty((rows, c), dtype=torch.uint8, device=x.device)
        rscale = _empty(rows, dtype=torch.float32, device=x.device)
        _check(load().cinema_layernorm_fwd_fp8(x.data_ptr(), int(x.dtype == torch.bfloat16), _rowmajor(x, "x"), gamma.data_ptr(), beta.data_ptr(), rows, c, eps, act,
                                               _p(y16), _p(y32), c, mean.data_ptr(), rstd.data_ptr(), y8.data_ptr(), rscale.data_ptr(), _stream()), "layernorm_fwd_fp8")
        return y16, y32, mean, rstd, (y8, rscale)
    _check(load().cinema_layernorm_fwd(x.data_ptr(), int(x.dtype == torch.bfloat16), _rowmajor(x, "x"), gamma.data_ptr(), beta.data_ptr(),
                                       rows, c, eps, act, _p(y16), _p(y32), c, mean.data_ptr(), rstd.data_ptr(), _stream()), "layernorm_fwd")
    return y16, y32, mean, rstd


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor | None, mean: torch.Tensor, rstd: torch.Tensor, *,
                  act: int = 0, dx_residual: torch.Tensor | None = None, want_f32: bool = True, want_bf16: bool = False,
                  dgamma: torch.Tensor | None = None, dbeta: torch.Tensor | None = None, dx_f32_out: torch.Tensor | None = None,
                  deferred: list | None = None, q8: Q8Site | None = None, q8_colsum: torch.Tensor | None = None):  # noqa: ANN201
    """-> (dx_f32 | None, dx_bf16 | None); dgamma/dbeta (fp32 [c]) are accumulated in place when given - at once, or (``deferred`` list)
    by a later :func:`ln_param_reduce_batched` over the entries appended to that list.  ``q8``: -> (dx_f32, dx_bf16, (dx8, site.scale) | None, colsum_done), the
    8-bit copy of dx under the site's delayed scale; ``q8_colsum`` (fp32 [c]): the column sums of dx are accumulated there by the same deferred reduce (the bias
    gradient of the projection that produced x) - ``colsum_done`` says whether that form ran."""
    _dev(dy, x, gamma, mean, rstd, dx_residual, dgamma, dbeta)
    rows, c = x.shape
    if q8 is not None:
        if deferred is None or (dgamma is None and dbeta is None) or c % 4:
            r = layernorm_bwd(dy, x, gamma, beta, mean, rstd, act=act, dx_residual=dx_residual, want_f32=want_f32, want_bf16=True, dgamma=dgamma, dbeta=dbeta,
                              dx_f32_out=dx_f32_out, deferred=deferred)
            return r[0], r[1], H.quantize_fp8_site(r[1], q8), False
        dx32 = dx_f32_out if dx_f32_out is not None else (_empty((rows, c), dtype=torch.float32, device=x.device) if want_f32 else None)
        dx16 = _empty((rows, c), dtype=torch.bfloat16, device=x.device) if want_bf16 else None
        dx8 = _empty((rows, c), dtype=torch.uint8, device=x.device) if q8.ready else None
        if dx_residual is not None and (dx_residual.stride(0) != c or dx_residual.dtype != torch.float32):
            raise HipLibraryError("dx_residual must be dense fp32 [rows, c]")
        if q8_colsum is not None:
            _dev(q8_colsum)
            if q8_colsum.dtype != torch.float32 or q8_colsum.numel() != c or not q8_colsum.is_contiguous():
                raise HipLibraryError("q8_colsum must be dense fp32 [c]")
        ws = _empty(max(load().cinema_layernorm_bwd_workspace_bytes(rows, c) // 8 * (3 if q8_colsum is not None else 2), 4), dtype=torch.float32, device=x.device)
        n_part = C.c_int(0)
        q = q8.out(dx8, q8_colsum)
        _check(load().cinema_layernorm_bwd_deferred_q8(dy.data_ptr(), int(dy.dtype == torch.bfloat16), _rowmajor(dy, "dy"), x.data_ptr(),
                                                       int(x.dtype == torch.bfloat16), _rowmajor(x, "x"), gamma.data_ptr(), _p(beta), mean.data_ptr(),
                                                       rstd.data_ptr(), rows, c, act, _p(dx_residual), _p(dx32), _p(dx16), c, _p(dgamma), _p(dbeta),
                                                       ws.data_ptr(), ws.numel() * 4, C.byref(n_part), C.byref(q), _stream()), "layernorm_bwd_q8")
        if n_part.value > 0:
            deferred.append((ws, n_part.value, c, dgamma, dbeta, q8_colsum))
        return dx32, dx16, (None if dx8 is None else (dx8, q8.scale)), q8_colsum is not None  # (fewer than 64 workgroups: the kernel added the sums with atomics)
    dx32 = dx_f32_out if dx_f32_out is not None else (_empty((rows, c), dtype=torch.float32, device=x.device) if want_f32 else None)
    dx16 = _empty((rows, c), dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    if dx_residual is not None and (dx_residual.stride(0) != c or dx_residual.dtype != torch.float32):
        raise HipLibraryError("dx_residual must be dense fp32 [rows, c]")
    if deferred is not None and (dgamma is not None or dbeta is not None):
        # the per-block partial sums stay in a buffer of their own until ln_param_reduce_batched adds them up (end of the backward pass)
        # sized from the launch's actual grid (a fixed 2048-block buffer was 12.6 MB per LayerNorm at c = 768: ~1 GB held across a step)
        ws = _empty(max(load().cinema_layernorm_bwd_workspace_bytes(rows, c) // 4, 4), dtype=torch.float32, device=x.device)
        n_part = C.c_int(0)
        _check(load().cinema_layernorm_bwd_deferred(dy.data_ptr(), int(dy.dtype == torch.bfloat16), _rowmajor(dy, "dy"), x.data_ptr(),
                                                    int(x.dtype == torch.bfloat16), _rowmajor(x, "x"), gamma.data_ptr(), _p(beta), mean.data_ptr(),
                                                    rstd.data_ptr(), rows, c, act, _p(dx_residual), _p(dx32), _p(dx16), c, _p(dgamma), _p(dbeta),
                                                    ws.data_ptr(), ws.numel() * 4, C.byref(n_part), _stream()), "layernorm_bwd")
        if n_part.value > 0:
            deferred.append((ws, n_part.value, c, dgamma, dbeta))
        return dx32, dx16
    ws = _workspace("ln_bwd", 2048 * 2 * c, x.device) if (dgamma is not None or dbeta is not None) else None
    _check(load().cinema_layernorm_bwd(dy.data_ptr(), int(dy.dtype == torch.bfloat16), _rowmajor(dy, "dy"), x.data_ptr(),
                                       int(x.dtype == torch.bfloat16), _rowmajor(x, "x"), gamma.data_ptr(), _p(beta), mean.data_ptr(),
                                       rstd.data_ptr(), rows, c, act, _p(dx_residual), _p(dx32), _p(dx16), c, _p(dgamma), _p(dbeta),
                                       ws.data_ptr() if ws is not None else None, 0 if ws is None else ws.numel() * 4, _stream()), "layernorm_bwd")
    return dx32, dx16


def ln_param_reduce_batched(items: list) -> None:
    """items: (partials, n_partials, c, dgamma | None, dbeta | None) from layernorm_bwd(..., deferred=list): one launch per 48 LayerNorms."""
    if not items:
        return
    arr = (LnReduceItem * len(items))()
    for e, it in zip(arr, items):
        ws, n_part, c, dg, db = it[:5]
        e.partials, e.n_partials, e.c, e.dgamma, e.dbeta = ws.data_ptr(), n_part, c, _p(dg), _p(db)
        e.dcol = _p(it[5]) if len(it) > 5 else None  # third partial row: column sums of dx (layernorm_bwd(q8_colsum=...))
    _check(load().cinema_ln_param_reduce_batched(arr, len(items), _stream()), "ln_param_reduce_batched")
