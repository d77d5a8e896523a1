"""Row movers and element-wise kernels (csrc/rows.hip, csrc/select.hip, csrc/input.hip): gathers / scatters, casts, dropout / drop-path, masks, patch re-layouts, fills.

Part of the ctypes front of ``libcinema_hip.so`` (see ``cinema_amd/hip/__init__.py`` for the loader, the launch recorder, lane groups and the per-stream
workspaces); everything here is re-exported there, so callers keep writing ``hip.<name>``.  Module-level switches and the recorder state live in the package and
are read through it (``H.<NAME>``) so that an assignment ``hip.<NAME> = ...`` is seen by every family."""
from __future__ import annotations

import ctypes as C
import struct

import torch

from cinema_amd import hip as H
from cinema_amd.hip import (  # noqa: F401
    HipLibraryError, PatchGeom, RowCopyArgs, _DT, _check, _dev, _empty, _empty_like, _p, _rowmajor, _stream, _workspace, load, persistent,
)

__all__ = ['mul_rows', '_RNG_STATE', 'cast', 'colsum', 'convt_weight_grad_accumulate', 'convt_weight_rows', 'dropout', 'droppath_scale', 'full', 'gelu_bwd', 'gelu_fwd', 'mask_select', 'mul_scalar', 'patch_gather', 'patch_geom', 'patch_scatter', 'patch_weight_grad_accumulate', 'patch_weight_rows', 'random_mask', 'rng_advance', 'rng_seed', 'rng_state', 'rope_heads', 'row_copy', 'row_copy_multi', 'scale', 'scale_rows_add', 'scale_rows_bf16', 'segment_mean', 'segment_mean_bwd', 'transpose_cast', 'visible_index', 'zeros', 'zoom_scale_pad']


def segment_mean(x: torch.Tensor, n_seg: int, scale: float | None = None) -> torch.Tensor:
    """x fp32 [n_seg * seg_rows, c] -> [n_seg, c]: scale (default 1/seg_rows) times the sum over each block of consecutive rows."""
    _dev(x)
    if x.dtype != torch.float32:
        raise HipLibraryError("segment_mean: x must be fp32")
    rows, c = x.shape
    seg_rows = rows // n_seg
    out = _empty((n_seg, c), dtype=torch.float32, device=x.device)
    _check(load().cinema_segment_mean_fwd(x.data_ptr(), _rowmajor(x, "x"), n_seg, seg_rows, c, 1.0 / seg_rows if scale is None else scale, out.data_ptr(),
                                          _stream()), "segment_mean_fwd")
    return out


def segment_mean_bwd(dy: torch.Tensor, seg_rows: int, scale: float | None = None) -> torch.Tensor:
    _dev(dy)
    n_seg, c = dy.shape
    dx = _empty((n_seg * seg_rows, c), dtype=torch.float32, device=dy.device)
    _check(load().cinema_segment_mean_bwd(dy.data_ptr(), n_seg, seg_rows, c, 1.0 / seg_rows if scale is None else scale, dx.data_ptr(), c, 0, _stream()),
           "segment_mean_bwd")
    return dx


def scale(x: torch.Tensor, alpha: float) -> torch.Tensor:
    _dev(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise HipLibraryError("scale: contiguous fp32 only")
    y = _empty_like(x)
    _check(load().cinema_scale_f32(x.data_ptr(), alpha, y.data_ptr(), x.numel(), _stream()), "scale")
    return y


def mul_scalar(x: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """x * s[0] for contiguous fp32 x and a one-element fp32 device tensor s."""
    _dev(x, s)
    if x.dtype != torch.float32 or s.dtype != torch.float32 or not x.is_contiguous() or s.numel() != 1:
        raise HipLibraryError("mul_scalar: contiguous fp32 x, one-element fp32 s")
    y = _empty_like(x)
    _check(load().cinema_mul_scalar_f32(x.data_ptr(), s.data_ptr(), y.data_ptr(), x.numel(), _stream()), "mul_scalar")
    return y


def mul_rows(a: torch.Tensor, b: torch.Tensor, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """a [rows, c] * b, with b either [c] (one factor per channel: timm ``LayerScale``, ``cinema/vit.py:561,576``) or [rows, c]; fp32 / bf16 operands, contiguous."""
    _dev(a, b)
    rows, c = a.shape
    full = b.dim() == 2
    if not a.is_contiguous() or not b.is_contiguous() or (tuple(b.shape) != (rows, c) if full else tuple(b.shape) != (c,)):
        raise HipLibraryError(f"mul_rows: contiguous [rows, c] times [c] or [rows, c], got {tuple(a.shape)} and {tuple(b.shape)}")
    if any(t.dtype not in (torch.float32, torch.bfloat16) for t in (a, b)) or out_dtype not in (torch.float32, torch.bfloat16):
        raise HipLibraryError("mul_rows: fp32 / bf16 only")
    y = _empty((rows, c), dtype=out_dtype, device=a.device)
    _check(load().cinema_mul_rows(a.data_ptr(), int(a.dtype == torch.bfloat16), b.data_ptr(), int(b.dtype == torch.bfloat16), int(full), y.data_ptr(),
                                  int(out_dtype == torch.bfloat16), rows, c, _stream()), "mul_rows")
    return y


def rope_heads(x: torch.Tensor, n_slots: int, heads: int, head_dim: int, cos: torch.Tensor, sin: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    """In-place head-indexed rotary embedding on bf16 rows x [rows, >= n_slots*head_dim] (see ``cinema_rope_heads``); cos/sin fp32 [heads, rotary_dim/2]."""
    _dev(x, cos, sin)
    if x.dtype != torch.bfloat16 or cos.dtype != torch.float32 or sin.dtype != torch.float32 or not cos.is_contiguous() or not sin.is_contiguous():
        raise HipLibraryError("rope_heads: bf16 rows, contiguous fp32 tables")
    if cos.shape != sin.shape or cos.shape[0] != heads:
        raise HipLibraryError(f"rope_heads: tables must be [heads={heads}, rotary_dim/2], got {tuple(cos.shape)}")
    _check(load().cinema_rope_heads(x.data_ptr(), _rowmajor(x, "x"), x.shape[0], n_slots, heads, head_dim, 2 * cos.shape[1], cos.data_ptr(), sin.data_ptr(),
                                    int(inverse), _stream()), "rope_heads")
    return x


# ---- stochastic regularisers (dropout / drop-path): device RNG state = [step counter, seed] as 2 x int64
_RNG_STATE: dict = {}


def rng_state(device: torch.device) -> torch.Tensor:
    """The per-device Philox state tensor (int64 [2]: step counter, seed); the seed is drawn from torch's generator on first use, so
    ``torch.manual_seed`` makes the dropout / drop-path masks reproducible."""
    key = torch.device(device).index or 0
    st = _RNG_STATE.get(key)
    if st is None:
        seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
        st = _RNG_STATE[key] = persistent(lambda: torch.tensor([0, seed], dtype=torch.int64, device=device))
    return st


def rng_seed(device: torch.device, seed: int) -> None:
    st = rng_state(device)
    st.copy_(torch.tensor([0, int(seed)], dtype=torch.int64))


def rng_advance(device: torch.device) -> None:
    """New masks from here on (one launch; part of a recorded step's list)."""
    _check(load().cinema_rng_advance(rng_state(device).data_ptr(), _stream()), "rng_advance")


def dropout(x: torch.Tensor, p: float, salt: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """y = x * keep / (1 - p) on contiguous bf16 (``nn.Dropout`` in training mode); the same call on a gradient is the backward pass."""
    _dev(x, out)
    if x.dtype != torch.bfloat16 or not x.is_contiguous():
        raise HipLibraryError("dropout: contiguous bf16 only")
    y = _empty_like(x) if out is None else out
    _check(load().cinema_dropout_bf16(x.data_ptr(), y.data_ptr(), x.numel(), float(p), rng_state(x.device).data_ptr(), salt & 0xFFFFFFFF, _stream()), "dropout")
    return y


def droppath_scale(batch: int, p: float, salt: int, device: torch.device) -> torch.Tensor:
    """fp32 [batch]: 0 or 1 / (1 - p) per sample (timm ``DropPath``, ``scale_by_keep=True``)."""
    s = _empty(batch, dtype=torch.float32, device=device)
    _dev(s)
    _check(load().cinema_droppath_scale(s.data_ptr(), batch, float(p), rng_state(device).data_ptr(), salt & 0xFFFFFFFF, _stream()), "droppath_scale")
    return s


def scale_rows_add(h: torch.Tensor, scale: torch.Tensor, rows_per_sample: int, residual: torch.Tensor | None = None) -> torch.Tensor:
    """residual + scale[row // rows_per_sample] * h over fp32 rows [n, c]."""
    _dev(h, scale, residual)
    if h.dtype != torch.float32 or not h.is_contiguous() or (residual is not None and (residual.dtype != torch.float32 or not residual.is_contiguous())):
        raise HipLibraryError("scale_rows_add: contiguous fp32 rows")
    out = _empty_like(h)
    _check(load().cinema_scale_rows_add(h.data_ptr(), _p(residual), scale.data_ptr(), out.data_ptr(), h.shape[0], h.shape[1], rows_per_sample, _stream()),
           "scale_rows_add")
    return out


def scale_rows_bf16(h: torch.Tensor, scale: torch.Tensor, rows_per_sample: int) -> torch.Tensor:
    """bf16(scale[row // rows_per_sample] * h) over fp32 rows [n, c]: the gradient of a DropPath branch as the GEMM operand its readers take."""
    _dev(h, scale)
    if h.dtype != torch.float32 or not h.is_contiguous():
        raise HipLibraryError("scale_rows_bf16: contiguous fp32 rows")
    out = _empty(h.shape, dtype=torch.bfloat16, device=h.device)
    _check(load().cinema_scale_rows_bf16(h.data_ptr(), scale.data_ptr(), out.data_ptr(), h.shape[0], h.shape[1], rows_per_sample, _stream()), "scale_rows_bf16")
    return out


def full(shape, value: float, dtype: torch.dtype = torch.float32, device=None) -> torch.Tensor:  # noqa: ANN001
    """torch.full / torch.zeros as a launch of this library (so that it is part of a recorded step, see cinema_amd/replay.py): fp32 with
    any value, other dtypes with zero only; the tensor must span whole 32-bit words."""
    t = _empty(shape, dtype=dtype, device=device)
    _dev(t)
    nbytes = t.numel() * t.element_size()
    if value == 0:
        word = 0
    elif dtype == torch.float32:
        word = struct.unpack("<I", struct.pack("<f", value))[0]
    else:
        raise HipLibraryError("full: non-zero fills are fp32 only")
    if nbytes % 4:
        raise HipLibraryError("full: the tensor must span whole 32-bit words")
    _check(load().cinema_fill_u32(t.data_ptr(), word, nbytes // 4, _stream()), "fill")
    return t


def zeros(shape, dtype: torch.dtype = torch.float32, device=None) -> torch.Tensor:  # noqa: ANN001
    return full(shape, 0.0, dtype, device)


def patch_weight_rows(w: torch.Tensor, jmap: torch.Tensor | None = None, pad_to: int = 1) -> torch.Tensor:
    """Conv weight (out, c, *k) fp32 -> bf16 GEMM operand [out, ld], features ordered (*k, c); ld = kvol*c rounded up to ``pad_to``."""
    _dev(w, jmap)
    if w.dtype != torch.float32 or not w.is_contiguous() or (jmap is not None and (jmap.dtype != torch.int32 or jmap.numel() != w[0, 0].numel())):
        raise HipLibraryError("patch_weight_rows: contiguous fp32 weight, int32 jmap with one entry per kernel voxel")
    out, c = w.shape[0], w.shape[1]
    kvol = w[0, 0].numel()
    ld = (kvol * c + pad_to - 1) // pad_to * pad_to
    rows = _empty((out, ld), dtype=torch.bfloat16, device=w.device)
    _check(load().cinema_patch_weight_relayout(w.data_ptr(), rows.data_ptr(), 1, out, c, kvol, ld, _p(jmap), 0, _stream()), "patch_weight_relayout")
    return rows


def patch_weight_grad_accumulate(g_rows: torch.Tensor, w_grad: torch.Tensor, jmap: torch.Tensor | None = None) -> None:
    """w_grad (out, c, *k) fp32 += g_rows fp32 [out, ld] (features (*k, c), padding ignored)."""
    _dev(g_rows, w_grad, jmap)
    if (g_rows.dtype != torch.float32 or w_grad.dtype != torch.float32 or not w_grad.is_contiguous() or g_rows.shape[0] != w_grad.shape[0]
            or (jmap is not None and (jmap.dtype != torch.int32 or jmap.numel() != w_grad[0, 0].numel()))):
        raise HipLibraryError("patch_weight_grad_accumulate: fp32 tensors, contiguous destination")
    out, c = w_grad.shape[0], w_grad.shape[1]
    kvol = w_grad[0, 0].numel()
    _check(load().cinema_patch_weight_relayout(w_grad.data_ptr(), g_rows.data_ptr(), 0, out, c, kvol, _rowmajor(g_rows, "g_rows"), _p(jmap), 1, _stream()),
           "patch_weight_relayout")


def convt_weight_rows(w: torch.Tensor, bias: torch.Tensor | None = None) -> tuple:
    """Transposed-conv weight fp32 (c_in, c_out, *k) -> (bf16 GEMM rows [(kv, c_out), c_in], bias repeated per kernel voxel or None)."""
    _dev(w, bias)
    if w.dtype != torch.float32 or not w.is_contiguous() or (bias is not None and (bias.dtype != torch.float32 or bias.numel() != w.shape[1])):
        raise HipLibraryError("convt_weight_rows: contiguous fp32 weight (c_in, c_out, *k), fp32 bias [c_out]")
    c_in, c_out = w.shape[0], w.shape[1]
    kvol = w[0, 0].numel()
    rows = _empty((kvol * c_out, c_in), dtype=torch.bfloat16, device=w.device)
    bias_t = None if bias is None else _empty(kvol * c_out, dtype=torch.float32, device=w.device)
    _check(load().cinema_convt_weight_relayout(w.data_ptr(), rows.data_ptr(), c_in, c_out, kvol, 0, _p(bias), _p(bias_t), _stream()), "convt_weight_relayout")
    return rows, bias_t


def convt_weight_grad_accumulate(g_rows: torch.Tensor, w_grad: torch.Tensor) -> None:
    """w_grad (c_in, c_out, *k) fp32 += g_rows fp32 [(kv, c_out), c_in]."""
    _dev(g_rows, w_grad)
    c_in, c_out = w_grad.shape[0], w_grad.shape[1]
    kvol = w_grad[0, 0].numel()
    if g_rows.dtype != torch.float32 or w_grad.dtype != torch.float32 or not w_grad.is_contiguous() or not g_rows.is_contiguous() or tuple(g_rows.shape) != (kvol * c_out, c_in):
        raise HipLibraryError("convt_weight_grad_accumulate: fp32 rows [(kv, c_out), c_in], contiguous fp32 destination")
    _check(load().cinema_convt_weight_relayout(w_grad.data_ptr(), g_rows.data_ptr(), c_in, c_out, kvol, 1, None, None, _stream()), "convt_weight_relayout")


def colsum(x: torch.Tensor, out: torch.Tensor, row_idx: torch.Tensor | None = None) -> torch.Tensor:
    """out[n] += sum_i x[row(i), n] (x bf16/fp32 2-D, out fp32; ``row_idx`` int32 selects rows)."""
    _dev(x, out, row_idx)
    m = x.shape[0] if row_idx is None else row_idx.numel()
    n = x.shape[1]
    if row_idx is None and n < 8 and 64 % n == 0 and x.is_contiguous() and (m * n) % 64 == 0 and m * n >= (1 << 16):
        # a narrow, tall matrix (bias gradient of a 4-class head over millions of voxels): the vectorised kernels want >= 8 / 4 columns per thread
        # group, so fold 64 / n rows into one 64-wide row, sum those columns, then sum the 64 / n groups of n
        tmp = zeros((64,), torch.float32, x.device)
        colsum(x.view(-1, 64), tmp)
        return colsum(tmp.view(64 // n, n), out)
    _check(load().cinema_colsum(x.data_ptr(), _DT[x.dtype], _p(row_idx), m, x.shape[1], _rowmajor(x, "x"), out.data_ptr(), _stream()), "colsum")
    return out


def random_mask(noise: torch.Tensor, n_keep: int) -> torch.Tensor:
    """bool [b, n], True = removed: the n - n_keep largest of each row of ``noise`` (ties by index), i.e. ``argsort(argsort(noise)) >= n_keep``."""
    _dev(noise)
    if noise.dtype != torch.float32 or noise.dim() != 2 or not noise.is_contiguous():
        raise HipLibraryError("random_mask: contiguous fp32 [batch, n] noise")
    b, n = noise.shape
    mask = _empty((b, n), dtype=torch.bool, device=noise.device)
    _check(load().cinema_mask_select(noise.data_ptr(), mask.data_ptr(), b, n, n_keep, None, None, None, None, _stream()), "mask_select")
    return mask


def mask_select(mask: torch.Tensor, n_keep: int) -> tuple:
    """bool [b, n] with n_keep False per row -> (keep_pos, drop_pos, keep, drop): int32 raster-ordered positions / flat ids b*n + i."""
    _dev(mask)
    if mask.dtype != torch.bool or mask.dim() != 2 or not mask.is_contiguous():
        raise HipLibraryError("mask_select: contiguous bool [batch, n] mask")
    b, n = mask.shape
    outs = [_empty(b * k, dtype=torch.int32, device=mask.device) for k in (n_keep, n - n_keep, n_keep, n - n_keep)]
    _check(load().cinema_mask_select(None, mask.data_ptr(), b, n, n_keep, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(),
                                     _stream()), "mask_select")
    return tuple(outs)


def visible_index(keep: torch.Tensor, batch: int, grid: tuple, block: tuple, inv1: torch.Tensor) -> tuple:
    """-> (rank int32 [batch * prod(grid)]: compact index of a kept token or -1, idx1 int32 [n_kept * prod(block)]: stage-1 voxel ids in row order)."""
    _dev(keep, inv1)
    if keep.dtype != torch.int32 or inv1.dtype != torch.int32 or not keep.is_contiguous() or not inv1.is_contiguous():
        raise HipLibraryError("visible_index: contiguous int32 index tensors")
    nd = len(grid)
    n_all, vol = batch, 1
    for g_ in grid:
        n_all *= int(g_)
    for b_ in block:
        vol *= int(b_)
    if inv1.numel() != vol:
        raise HipLibraryError("visible_index: inv1 must have one entry per voxel of a token block")
    rank = _empty(n_all, dtype=torch.int32, device=keep.device)
    _check(load().cinema_fill_u32(rank.data_ptr(), 0xFFFFFFFF, n_all, _stream()), "fill")  # -1
    idx1 = _empty(keep.numel() * vol, dtype=torch.int32, device=keep.device)
    garr, barr = (C.c_int * nd)(*[int(v) for v in grid]), (C.c_int * nd)(*[int(v) for v in block])
    _check(load().cinema_visible_index(keep.data_ptr(), keep.numel(), nd, garr, barr, inv1.data_ptr(), rank.data_ptr(), idx1.data_ptr(), _stream()), "visible_index")
    return rank, idx1


def patch_geom(batch: int, chans: int, grid: tuple, patch: tuple, strides: tuple, token_idx: torch.Tensor | None = None,
               n_rows: int | None = None) -> PatchGeom:
    """``strides`` = element strides (batch, channel, *spatial) of the volume; 2-D is padded with a unit z axis."""
    g = PatchGeom()
    grid3 = tuple(grid) + (1,) * (3 - len(grid))
    patch3 = tuple(patch) + (1,) * (3 - len(patch))
    sp = tuple(strides[2:]) + (0,) * (3 - len(grid))
    g.b, g.c = batch, chans
    g.gx, g.gy, g.gz = grid3
    g.px, g.py, g.pz = patch3
    g.sb, g.sc, g.sx, g.sy, g.sz = strides[0], strides[1], sp[0], sp[1], sp[2]
    n_tok = batch * grid3[0] * grid3[1] * grid3[2]
    if token_idx is not None:
        if token_idx.dtype != torch.int32:
            raise HipLibraryError("token_idx must be int32")
        g.token_idx = token_idx.data_ptr()
        g.keepalive = token_idx  # the struct only holds the raw pointer; backward closures outlive the caller's locals
        g.n_rows = token_idx.numel() if n_rows is None else n_rows
    else:
        g.n_rows = n_tok
    return g


def patch_gather(src: torch.Tensor, geom: PatchGeom, out_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    _dev(src)
    feat = geom.px * geom.py * geom.pz * geom.c
    out = _empty((geom.n_rows, feat), dtype=out_dtype, device=src.device)
    _check(load().cinema_patch_gather(src.data_ptr(), _DT[src.dtype], out.data_ptr(), _DT[out_dtype], feat, C.byref(geom), _stream()), "patch_gather")
    return out


def patch_scatter(rows: torch.Tensor, dst: torch.Tensor, geom: PatchGeom, accumulate: bool = False) -> torch.Tensor:
    _dev(rows, dst)
    _check(load().cinema_patch_scatter(rows.data_ptr(), _DT[rows.dtype], _rowmajor(rows, "rows"), dst.data_ptr(), _DT[dst.dtype], int(accumulate),
                                       C.byref(geom), _stream()), "patch_scatter")
    return dst


def row_copy(dst: torch.Tensor, src: torch.Tensor | None = None, *, dst_idx: torch.Tensor | None = None, src_idx: torch.Tensor | None = None,
             add: torch.Tensor | None = None, add_idx: torch.Tensor | None = None, n_rows: int | None = None, accumulate: bool = False) -> torch.Tensor:
    """dst[di(i)] (+)= src[si(i)] + add[ai(i)] over 2-D row-major tensors (bf16/fp32), indices int32."""
    _dev(dst, src, add, dst_idx, src_idx, add_idx)
    for idx in (dst_idx, src_idx, add_idx):
        if idx is not None and idx.dtype != torch.int32:
            raise HipLibraryError("row_copy indices must be int32")
    if n_rows is None:
        n_rows = next((i.numel() for i in (dst_idx, src_idx, add_idx) if i is not None), dst.shape[0])
    c = dst.shape[1]
    _check(load().cinema_row_copy(dst.data_ptr(), _DT[dst.dtype], _rowmajor(dst, "dst"), _p(dst_idx), _p(src), _DT[src.dtype] if src is not None else 0,
                                  _rowmajor(src, "src") if src is not None else 0, _p(src_idx), _p(add), _DT[add.dtype] if add is not None else 0,
                                  _rowmajor(add, "add") if add is not None else 0, _p(add_idx), n_rows, c, int(accumulate), _stream()), "row_copy")
    return dst


def row_copy_multi(copies: list) -> None:
    """Several independent :func:`row_copy` calls in one launch: ``copies`` = list of dicts with row_copy's arguments (dst, src, dst_idx, ...)."""
    if not copies:
        return
    if len(copies) == 1 or not H.ROW_COPY_MULTI:
        for kw in copies:
            row_copy(**kw)
        return
    arr = (RowCopyArgs * len(copies))()
    for a, kw in zip(arr, copies):
        dst, src, add = kw["dst"], kw.get("src"), kw.get("add")
        dst_idx, src_idx, add_idx = kw.get("dst_idx"), kw.get("src_idx"), kw.get("add_idx")
        _dev(dst, src, add, dst_idx, src_idx, add_idx)
        for idx in (dst_idx, src_idx, add_idx):
            if idx is not None and idx.dtype != torch.int32:
                raise HipLibraryError("row_copy indices must be int32")
        n_rows = kw.get("n_rows")
        if n_rows is None:
            n_rows = next((i.numel() for i in (dst_idx, src_idx, add_idx) if i is not None), dst.shape[0])
        a.dst, a.dst_dtype, a.ld_dst, a.dst_idx = dst.data_ptr(), _DT[dst.dtype], _rowmajor(dst, "dst"), _p(dst_idx)
        a.src, a.src_dtype, a.ld_src, a.src_idx = _p(src), _DT[src.dtype] if src is not None else 0, _rowmajor(src, "src") if src is not None else 0, _p(src_idx)
        a.add, a.add_dtype, a.ld_add, a.add_idx = _p(add), _DT[add.dtype] if add is not None else 0, _rowmajor(add, "add") if add is not None else 0, _p(add_idx)
        a.n_rows, a.c, a.accumulate = n_rows, dst.shape[1], int(bool(kw.get("accumulate", False)))
    _check(load().cinema_row_copy_multi(arr, len(copies), _stream()), "row_copy_multi")


def cast(src: torch.Tensor, dtype: torch.dtype, out: torch.Tensor | None = None) -> torch.Tensor:
    _dev(src, out)
    if not src.is_contiguous():
        raise HipLibraryError("cast needs a contiguous source")
    if out is None:
        out = _empty(src.shape, dtype=dtype, device=src.device)
    _check(load().cinema_cast(src.data_ptr(), _DT[src.dtype], out.data_ptr(), _DT[out.dtype], src.numel(), _stream()), "cast")
    return out


def transpose_cast(src: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """[r, c] fp32/bf16 contiguous -> [c, r] bf16."""
    _dev(src, out)
    r, c = src.shape
    if out is None:
        out = _empty((c, r), dtype=torch.bfloat16, device=src.device)
    _check(load().cinema_transpose_cast(src.data_ptr(), _DT[src.dtype], r, c, out.data_ptr(), _stream()), "transpose_cast")
    return out


def gelu_fwd(x: torch.Tensor) -> torch.Tensor:
    _dev(x)
    y = _empty_like(x)
    _check(load().cinema_gelu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "gelu_fwd")
    return y


def gelu_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    _dev(x, dy)
    dx = _empty_like(x)
    _check(load().cinema_gelu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _stream()), "gelu_bwd")
    return dx


def zoom_scale_pad(src: torch.Tensor, zoom: tuple, dst: torch.Tensor, cubic: bool = False) -> None:
    """One sample of the input pipeline: fp32 image / volume ``src`` (*size) -> zoom (keep size) -> ScaleIntensity to [0, 1] -> written into the
    zero-padded slot ``dst`` (*padded_size).  Two launches + a one-thread init; scratch from the per-stream workspace."""
    _dev(src, dst)
    if src.dtype != torch.float32 or dst.dtype != torch.float32 or not src.is_contiguous() or not dst.is_contiguous() or src.dim() != dst.dim() or src.dim() not in (2, 3):
        raise HipLibraryError("zoom_scale_pad: contiguous fp32 2-D / 3-D tensors")
    if any(d < s for d, s in zip(dst.shape, src.shape)):
        raise HipLibraryError("zoom_scale_pad: the destination must be at least as large as the source")
    s3 = tuple(src.shape) + (1,) * (3 - src.dim())
    d3 = tuple(dst.shape) + (1,) * (3 - dst.dim())
    z3 = tuple(float(z) for z in zoom) + (1.0,) * (3 - len(zoom))
    ws = _workspace("zoom", src.numel() + 4, src.device)
    tmp, mm = ws[4:4 + src.numel()], ws[:2]
    _check(load().cinema_zoom_resample(src.data_ptr(), *s3, *z3, int(cubic), tmp.data_ptr(), mm.data_ptr(), _stream()), "zoom_resample")
    _check(load().cinema_scale_intensity_pad(tmp.data_ptr(), *s3, mm.data_ptr(), dst.data_ptr(), *d3, _stream()), "scale_intensity_pad")
