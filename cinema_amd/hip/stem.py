"""Visible-voxel conv stem (csrc/sparse_conv.hip, csrc/stem_dw.hip, csrc/stem.hip): sparse geometry, depthwise convolutions on kept tokens, fused MaskedConvBlock halves.

Part of the ctypes front of ``libcinema_hip.so`` (see ``cinema_amd/hip/__init__.py`` for the loader, the launch recorder, lane groups and the per-stream
workspaces); everything here is re-exported there, so callers keep writing ``hip.<name>``.  Module-level switches and the recorder state live in the package and
are read through it (``H.<NAME>``) so that an assignment ``hip.<NAME> = ...`` is seen by every family."""
from __future__ import annotations

import ctypes as C

import torch

from cinema_amd import hip as H
from cinema_amd.hip import (  # noqa: F401
    HipLibraryError, SparseGeom, StemWgradProblem, _check, _dev, _empty, _empty_like, _p, _stream, _workspace, load, on_stream, stream_fork,
)

__all__ = ['_kernel3', '_sparse_nbr', '_stem_rows', 'sparse_dwconv', 'sparse_dwconv_bwd_weight', 'sparse_geom', 'sparse_nbr_prefetch', 'sparse_pair_form', 'stem_ln_linear', 'stem_ln_linear_bwd', 'stem_mlp_bwd', 'stem_mlp_fwd', 'stem_supported', 'stem_wgrad']


def sparse_geom(batch: int, tok_grid: tuple, block: tuple, keep: torch.Tensor, rank: torch.Tensor, pos: torch.Tensor) -> SparseGeom:
    """``tok_grid`` / ``block``: 2-D or 3-D (token grid per sample, voxels per token); 2-D maps use a leading axis of 1 like the dense kernels."""
    _dev(keep, rank, pos)
    tg = (1,) * (3 - len(tok_grid)) + tuple(int(v) for v in tok_grid)
    bl = (1,) * (3 - len(block)) + tuple(int(v) for v in block)
    for t in (keep, rank, pos):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise HipLibraryError("sparse_geom index tensors must be contiguous int32")
    g = SparseGeom()
    g.b, (g.tx, g.ty, g.tz), (g.bx, g.by, g.bz) = batch, tg, bl
    g.n_tok, g.keep, g.rank, g.pos = keep.numel(), keep.data_ptr(), rank.data_ptr(), pos.data_ptr()
    g.keepalive = (keep, rank, pos)
    g.nbr_lists = {}  # kernel extent -> (nbr, cnt), built on first use for this mask
    g.halo_idx = {}   # kernel extent -> halo source rows of every kept token (weight gradient)
    return g


def sparse_nbr_prefetch(items: list, device: torch.device, stream: int) -> None:
    """Build the neighbour lists of ``items`` = [(geom, kdims), ...] on ANOTHER stream, forked from the current one here and joined by the first consumer
    (:func:`_sparse_nbr`): the lists depend on the mask only, so they need not sit in the chain gather -> patch GEMM -> LayerNorm -> ... that precedes the first
    depthwise convolution (110 + 30 us at config 2)."""
    stream_fork(_stream(), stream)
    with on_stream(stream):
        for geom, kdims in items:
            _sparse_nbr(geom, kdims, device)
    for geom, _ in items:
        geom.nbr_wait = stream


def _sparse_nbr(geom: SparseGeom, kdims: tuple, device: torch.device) -> tuple:
    wait = getattr(geom, "nbr_wait", None)
    if wait is not None and wait != _stream():  # built by sparse_nbr_prefetch on another stream: this stream waits for it once
        stream_fork(wait, _stream())
        geom.nbr_wait = None
    hit = geom.nbr_lists.get(kdims)
    if hit is None:
        rows = geom.n_tok * geom.bx * geom.by * geom.bz
        buf = _empty(load().cinema_sparse_nbr_ints(rows), dtype=torch.int32, device=device)
        nbr, cnt = buf[:rows * 128], buf[rows * 128:]
        _check(load().cinema_sparse_nbr_build(C.byref(geom), *kdims, nbr.data_ptr(), cnt.data_ptr(), _stream()), "sparse_nbr_build")
        hit = geom.nbr_lists[kdims] = (nbr, cnt)
    return hit


def sparse_pair_form(geom: SparseGeom, c: int, kdims: tuple) -> bool:
    """Whether :func:`sparse_dwconv` / :func:`sparse_dwconv_bwd_weight` take the token-pair kernels for this geometry (then no neighbour list is ever built)."""
    return bool(H.STEM_DW_PAIR and load().cinema_stem_dw_supported(C.byref(geom), c, *kdims))


def _kernel3(w: torch.Tensor) -> tuple:
    ks = tuple(w.shape[2:])
    return (1,) * (3 - len(ks)) + ks


def sparse_dwconv(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None, geom: SparseGeom, flip: bool = False) -> torch.Tensor:
    """Depthwise conv on visible-voxel compact rows x bf16 [n_tok * block, c]; w fp32 [c, 1, *k]; flip=True: data gradient."""
    _dev(x, w, bias)
    if x.dtype != torch.bfloat16 or not x.is_contiguous() or w.dtype != torch.float32 or not w.is_contiguous():
        raise HipLibraryError("sparse_dwconv: x must be contiguous bf16 rows, w contiguous fp32")
    c = x.shape[1]
    kx, ky, kz = _kernel3(w)
    y = _empty_like(x)
    if H.STEM_DW_PAIR and load().cinema_stem_dw_supported(C.byref(geom), c, kx, ky, kz):  # token-pair form (csrc/stem_dw.hip): no neighbour lists
        _check(load().cinema_stem_dw_fwd(x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), C.byref(geom), c, kx, ky, kz, int(flip), _stream()), "stem_dw_fwd")
        return y
    nbr, cnt = _sparse_nbr(geom, (kx, ky, kz), x.device)
    _check(load().cinema_sparse_dwconv_fwd(x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), C.byref(geom), nbr.data_ptr(), cnt.data_ptr(), c, kx, ky, kz,
                                           int(flip), _stream()), "sparse_dwconv")
    return y


def sparse_dwconv_bwd_weight(x: torch.Tensor, dy: torch.Tensor, w_shape: tuple, dw: torch.Tensor, dbias: torch.Tensor | None, geom: SparseGeom) -> None:
    _dev(x, dy, dw, dbias)
    c = x.shape[1]
    ks = tuple(w_shape[2:])
    kx, ky, kz = (1,) * (3 - len(ks)) + ks
    if H.STEM_DW_PAIR and load().cinema_stem_dw_supported(C.byref(geom), c, kx, ky, kz):
        need = load().cinema_stem_dw_wgrad_workspace_bytes(geom.n_tok, c, kx, ky, kz)
        ws = _workspace("stem_dw_wgrad", (need + 3) // 4, x.device)
        _check(load().cinema_stem_dw_bwd_weight(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _p(dbias), ws.data_ptr(), need, C.byref(geom), c, kx, ky, kz, _stream()),
               "stem_dw_bwd_weight")
        return
    need = load().cinema_sparse_dwconv_wgrad_workspace_bytes(geom.n_tok, c, kx, ky, kz)
    ws = _workspace("sparse_wgrad", (need + 3) // 4, x.device)
    hidx = None
    if H.SPARSE_WGRAD_PIPE:
        hit = geom.halo_idx.get((kx, ky, kz))
        if hit is None:  # once per mask and kernel extent, on the stream of its first user; a user on another stream (two weight-gradient streams) waits for it
            hidx = _empty(max(load().cinema_sparse_halo_ints(C.byref(geom), kx, ky, kz), 1), dtype=torch.int32, device=x.device)
            _check(load().cinema_sparse_halo_index(C.byref(geom), kx, ky, kz, hidx.data_ptr(), _stream()), "sparse_halo_index")
            geom.halo_idx[(kx, ky, kz)] = (hidx, _stream())
        else:
            hidx, built_on = hit
            if built_on != _stream():
                stream_fork(built_on, _stream())
    _check(load().cinema_sparse_dwconv_bwd_weight(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _p(dbias), ws.data_ptr(), need, C.byref(geom), c, kx, ky, kz,
                                                  _p(hidx), _stream()), "sparse_dwconv_bwd_weight")


# ---- fused per-voxel halves of a MaskedConvBlock on compact rows (csrc/stem.hip) -------------------------------------------------------------------
def stem_supported(c: int, hidden: int) -> bool:
    return c in (64, 128) and hidden == 4 * c


def _stem_rows(t: torch.Tensor, dtype: torch.dtype, name: str) -> None:
    if t.dtype != dtype or t.dim() != 2 or not t.is_contiguous():
        raise HipLibraryError(f"stem kernels: {name} must be dense 2-D {dtype}, got {t.dtype} {tuple(t.shape)} strides {t.stride()}")


def stem_ln_linear(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, w16: torch.Tensor, bias: torch.Tensor | None, want_xn: bool = True) -> tuple:
    """-> (xn = LN(x) bf16 | None, h = xn w^T + bias bf16); x fp32 [rows, c], w16 bf16 [c, c]."""
    _dev(x, gamma, beta, w16, bias)
    _stem_rows(x, torch.float32, "x")
    _stem_rows(w16, torch.bfloat16, "w16")
    rows, c = x.shape
    xn = _empty((rows, c), dtype=torch.bfloat16, device=x.device) if want_xn else None
    h = _empty((rows, c), dtype=torch.bfloat16, device=x.device)
    _check(load().cinema_stem_ln_linear(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, w16.data_ptr(), _p(bias), _p(xn), h.data_ptr(), rows, c, _stream()), "stem_ln_linear")
    return xn, h


def stem_mlp_fwd(d: torch.Tensor, x: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, wf1: torch.Tensor,
                 bf1: torch.Tensor, wf2: torch.Tensor, bf2: torch.Tensor, want_x1: bool = True) -> tuple:
    """-> (x1 = x + d w2^T + b2 | None, x2 = x1 + fc2(GELU(fc1(LN(x1))))); d bf16, x fp32 [rows, c]; weights bf16 in nn.Linear layout."""
    _dev(d, x, w2, b2, gamma, beta, wf1, bf1, wf2, bf2)
    _stem_rows(d, torch.bfloat16, "d")
    _stem_rows(x, torch.float32, "x")
    for n, t in (("w2", w2), ("wf1", wf1), ("wf2", wf2)):
        _stem_rows(t, torch.bfloat16, n)
    rows, c = x.shape
    x1 = _empty((rows, c), dtype=torch.float32, device=x.device) if want_x1 else None
    x2 = _empty((rows, c), dtype=torch.float32, device=x.device)
    _check(load().cinema_stem_mlp_fwd(d.data_ptr(), x.data_ptr(), w2.data_ptr(), b2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, wf1.data_ptr(), bf1.data_ptr(),
                                      wf2.data_ptr(), bf2.data_ptr(), _p(x1), x2.data_ptr(), rows, c, _stream()), "stem_mlp_fwd")
    return x1, x2


def stem_mlp_bwd(g2: torch.Tensor, x1: torch.Tensor, w2: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, wf1: torch.Tensor, bf1: torch.Tensor,
                 wf2: torch.Tensor) -> dict:
    """Backward of :func:`stem_mlp_fwd` -> dict(dx1, dx1_16, dd, a, dz, xn2, g2_16, partials=(buffer, n_partials))."""
    _dev(g2, x1, w2, gamma, beta, wf1, bf1, wf2)
    _stem_rows(g2, torch.float32, "g2")
    _stem_rows(x1, torch.float32, "x1")
    rows, c = x1.shape
    dev = x1.device
    o = {"dx1": _empty((rows, c), dtype=torch.float32, device=dev)}
    for k in ("dx1_16", "dd", "xn2", "g2_16"):
        o[k] = _empty((rows, c), dtype=torch.bfloat16, device=dev)
    for k in ("a", "dz"):
        o[k] = _empty((rows, 4 * c), dtype=torch.bfloat16, device=dev)
    part = _empty((load().cinema_stem_partials(rows), 2 * c), dtype=torch.float32, device=dev)
    n_part = C.c_int(0)
    _check(load().cinema_stem_mlp_bwd(g2.data_ptr(), x1.data_ptr(), w2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, wf1.data_ptr(), bf1.data_ptr(), wf2.data_ptr(),
                                      o["dx1"].data_ptr(), o["dx1_16"].data_ptr(), o["dd"].data_ptr(), o["a"].data_ptr(), o["dz"].data_ptr(), o["xn2"].data_ptr(),
                                      o["g2_16"].data_ptr(), part.data_ptr(), rows, c, C.byref(n_part), _stream()), "stem_mlp_bwd")
    o["partials"] = (part, n_part.value)
    return o


def stem_ln_linear_bwd(dh: torch.Tensor, x: torch.Tensor, dres: torch.Tensor | None, gamma: torch.Tensor, eps: float, w16: torch.Tensor) -> tuple:
    """Backward of :func:`stem_ln_linear` -> (dx = dres + LN'(x)(dh w) fp32, (partials, n_partials))."""
    _dev(dh, x, dres, gamma, w16)
    _stem_rows(dh, torch.bfloat16, "dh")
    _stem_rows(x, torch.float32, "x")
    if dres is not None:
        _stem_rows(dres, torch.float32, "dres")
    rows, c = x.shape
    dx = _empty((rows, c), dtype=torch.float32, device=x.device)
    part = _empty((load().cinema_stem_partials(rows), 2 * c), dtype=torch.float32, device=x.device)
    n_part = C.c_int(0)
    _check(load().cinema_stem_ln_linear_bwd(dh.data_ptr(), x.data_ptr(), _p(dres), gamma.data_ptr(), eps, w16.data_ptr(), dx.data_ptr(), part.data_ptr(), rows, c,
                                            C.byref(n_part), _stream()), "stem_ln_linear_bwd")
    return dx, (part, n_part.value)


def stem_wgrad(problems: list) -> None:
    """problems: (dy bf16 [rows, n], x bf16 [rows, k], dw fp32 [n, k] (accumulated), db fp32 [n] | None), at most 6 with one row count: one launch + one reduce."""
    arr = (StemWgradProblem * len(problems))()
    for e, (dy, x, dw, db) in zip(arr, problems):
        _dev(dy, x, dw, db)
        _stem_rows(dy, torch.bfloat16, "dy")
        _stem_rows(x, torch.bfloat16, "x")
        if dw.dtype != torch.float32 or not dw.is_contiguous() or dw.numel() != dy.shape[1] * x.shape[1] or dy.shape[0] != x.shape[0]:
            raise HipLibraryError("stem_wgrad: dw must be dense fp32 [n, k] for dy [rows, n], x [rows, k]")
        e.dy, e.x, e.dw, e.db, e.rows, e.n, e.k = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), _p(db), dy.shape[0], dy.shape[1], x.shape[1]
    need = load().cinema_stem_wgrad_workspace_bytes(arr, len(problems))
    if need <= 0:
        raise HipLibraryError("stem_wgrad: unsupported problem list")
    ws = _workspace("stem_wgrad", (need + 3) // 4, problems[0][0].device)
    _check(load().cinema_stem_wgrad(arr, len(problems), ws.data_ptr(), need, _stream()), "stem_wgrad")
