"""Dense convolutions (csrc/conv.hip): depthwise, implicit-GEMM 3x3(x3), im2col / col2im, single-channel stem convolution.

Part of the ctypes front of ``libcinema_hip.so`` (see ``cinema_amd/hip/__init__.py`` for the loader, the launch recorder, lane groups and the per-stream
workspaces); everything here is re-exported there, so callers keep writing ``hip.<name>``.  Module-level switches and the recorder state live in the package and
are read through it (``H.<NAME>``) so that an assignment ``hip.<NAME> = ...`` is seen by every family."""
from __future__ import annotations

import ctypes as C

import torch

from cinema_amd import hip as H
from cinema_amd.hip import (  # noqa: F401
    GemmArgs, HipLibraryError, _check, _dev, _empty, _empty_like, _p, _rowmajor, _stream, _workspace, load,
)

__all__ = ['_conv1ch_geom', '_dw_dims', '_vol_dims', 'col2im', 'conv1ch_bwd', 'conv1ch_fwd', 'conv_coord_table', 'conv_gemm', 'conv_tap_table', 'conv_weight_dgrad', 'conv_weight_zblock', 'conv_wgrad', 'conv_wgrad_zfold', 'dwconv_bwd_data', 'dwconv_bwd_weight', 'dwconv_fwd', 'im2col']


def _conv1ch_geom(x: torch.Tensor, w: torch.Tensor) -> tuple:
    if x.dtype != torch.bfloat16 or w.dtype != torch.float32 or not x.is_contiguous() or not w.is_contiguous() or w.dim() != x.dim() + 1 or w.shape[1] != 1:
        raise HipLibraryError("conv1ch: contiguous bf16 volume [b, *spatial] and fp32 weight (n, 1, *k)")
    sp = (1,) * (3 - (x.dim() - 1)) + tuple(x.shape[1:])
    ks = (1,) * (3 - (w.dim() - 2)) + tuple(w.shape[2:])
    return (x.shape[0], *sp, *ks, w.shape[0])


def conv1ch_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """"Same" convolution of a one-channel volume x bf16 [b, *spatial] with the fp32 weight (n, 1, *k) (extents 1 or 3, n in {4, 8, 16, 32, 64}) + bias ->
    fp32 rows [b * prod(spatial), n]; direct stencil kernel."""
    _dev(x, w, bias)
    geom = _conv1ch_geom(x, w)
    y = _empty((x.numel(), w.shape[0]), dtype=torch.float32, device=x.device)
    _check(load().cinema_conv1ch_fwd(x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), *geom, _stream()), "conv1ch_fwd")
    return y


def conv1ch_bwd(x: torch.Tensor, w: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor | None, db: torch.Tensor | None, want_dx: bool) -> torch.Tensor | None:
    """Backward of :func:`conv1ch_fwd`: dy fp32 [rows, n]; dw (n, 1, *k) / db (n) fp32 accumulated in place; returns dx bf16 [rows, 1] when asked."""
    _dev(x, w, dy, dw, db)
    geom = _conv1ch_geom(x, w)
    if dy.dtype != torch.float32 or not dy.is_contiguous() or tuple(dy.shape) != (x.numel(), w.shape[0]) or (dw is not None and (dw.dtype != torch.float32 or not dw.is_contiguous() or dw.numel() != w.numel())):
        raise HipLibraryError("conv1ch_bwd: contiguous fp32 dy [rows, n], dw of the weight's size")
    dx = _empty((x.numel(), 1), dtype=torch.bfloat16, device=x.device) if want_dx else None
    _check(load().cinema_conv1ch_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), _p(dx), _p(dw), _p(db), *geom, _stream()), "conv1ch_bwd")
    return dx


# --------------------------------------------------------------------------------------------------------
def _dw_dims(x: torch.Tensor, ksize: tuple) -> tuple:
    """x: channels-last [b, *spatial, c]; 2-D maps are walked as (1, H, W) so the sliding-window axis is W."""
    b, *sp, c = x.shape
    if len(sp) == 2:
        return b, 1, sp[0], sp[1], c, 1, ksize[0], ksize[1]
    return b, sp[0], sp[1], sp[2], c, ksize[0], ksize[1], ksize[2]


def dwconv_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """x: bf16 channels-last [b, *spatial, c]; w: fp32 torch layout (c, 1, *k)."""
    _dev(x, w, bias)
    if not x.is_contiguous() or x.dtype != torch.bfloat16 or w.dtype != torch.float32 or not w.is_contiguous():
        raise HipLibraryError("dwconv: x must be contiguous bf16 channels-last, w contiguous fp32")
    y = _empty_like(x)
    b, X, Y, Z, c, kx, ky, kz = _dw_dims(x, tuple(w.shape[2:]))  # noqa: N806
    _check(load().cinema_dwconv_fwd(x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), b, X, Y, Z, c, kx, ky, kz, _stream()), "dwconv_fwd")
    return y


def dwconv_bwd_data(dy: torch.Tensor, w: torch.Tensor, out_mask: torch.Tensor | None = None) -> torch.Tensor:
    _dev(dy, w, out_mask)
    dx = _empty_like(dy)
    b, X, Y, Z, c, kx, ky, kz = _dw_dims(dy, tuple(w.shape[2:]))  # noqa: N806
    _check(load().cinema_dwconv_bwd_data(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), _p(out_mask), b, X, Y, Z, c, kx, ky, kz, _stream()), "dwconv_bwd_data")
    return dx


def dwconv_bwd_weight(x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, dbias: torch.Tensor | None) -> None:
    """dw (fp32, torch layout (c,1,*k)) and dbias (fp32 [c]) are accumulated in place."""
    _dev(x, dy, dw, dbias)
    b, X, Y, Z, c, kx, ky, kz = _dw_dims(x, tuple(dw.shape[2:]))  # noqa: N806
    ws = _workspace("dwconv_wgrad", 1024 * c * (kx * ky * kz + 1), x.device)  # per-block partial slabs (deterministic two-pass)
    _check(load().cinema_dwconv_bwd_weight(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _p(dbias), ws.data_ptr(), ws.numel() * 4, b, X, Y, Z, c, kx, ky,
                                           kz, _stream()), "dwconv_bwd_weight")


def _vol_dims(shape: tuple, ks: tuple) -> tuple:
    """(b, *spatial, c) with 2 or 3 spatial dims -> (b, X, Y, Z, c, kx, ky, kz); 2-D maps get a leading axis of 1."""
    b, c = shape[0], shape[-1]
    sp = (1,) * (3 - len(shape[1:-1])) + tuple(shape[1:-1])
    k3 = (1,) * (3 - len(ks)) + tuple(ks)
    return (b, *sp, c, *k3)


def conv_tap_table(c: int, ks: tuple, spatial: tuple, ld: int, transpose: bool, device: torch.device, zb: int = 1) -> torch.Tensor:
    """int32 [ld / 8, 4] table for :func:`conv_gemm`: per 16-byte k-chunk (8 channels of one tap) {row delta of the neighbour voxel, packed
    (dx+1, dy+1, dz+1), first channel, valid}.  ``transpose``: the offsets of the data gradient (the neighbour is at MINUS the tap offset).
    ``zb`` > 1: the z-blocked form (3x3x3 kernels): taps over (3, 3, zb + 2) offsets, dz counted from the first voxel of the row's z group."""
    k3 = (1,) * (3 - len(ks)) + tuple(int(v) for v in ks)
    sp = (1,) * (3 - len(spatial)) + tuple(int(v) for v in spatial)
    if any(k not in (1, 3) for k in k3):
        raise HipLibraryError("conv_gemm: kernel extents 1 or 3 only")
    rows = []
    if zb > 1:
        if k3 != (3, 3, 3) or sp[2] % zb or ld != 9 * (zb + 2) * c:
            raise HipLibraryError("conv_gemm: the z-blocked form needs a 3x3x3 kernel, Z % zb == 0 and ld = 9 * (zb + 2) * c")
        for j in range(ld // 8):
            tapz, ci = (j * 8) // c, (j * 8) % c
            txy, dzz = tapz // (zb + 2), tapz % (zb + 2) - 1
            d = [txy // 3 - 1, txy % 3 - 1]
            if transpose:
                d = [-v for v in d]
            rows.append((d[0] * sp[1] * sp[2] + d[1] * sp[2] + dzz, (d[0] + 1) | ((d[1] + 1) << 2) | ((dzz + 1) << 4), ci, 1))
        while len(rows) % 8:
            rows.append((0, 21, 0, 0))
        return torch.tensor(rows, dtype=torch.int32).to(device)
    taps = k3[0] * k3[1] * k3[2]
    for j in range(ld // 8):
        kk = j * 8
        tap, ci = kk // c, kk % c
        if tap >= taps:
            rows.append((0, 21, 0, 0))
            continue
        tz, ty, tx = tap % k3[2], (tap // k3[2]) % k3[1], tap // (k3[2] * k3[1])
        d = [tx - k3[0] // 2, ty - k3[1] // 2, tz - k3[2] // 2]
        if transpose:
            d = [-v for v in d]
        rows.append((d[0] * sp[1] * sp[2] + d[1] * sp[2] + d[2], (d[0] + 1) | ((d[1] + 1) << 2) | ((d[2] + 1) << 4), ci, 1))
    while len(rows) % 8:  # the kernel reads one entry per 16-byte chunk of whole 64-wide k-tiles
        rows.append((0, 21, 0, 0))
    return torch.tensor(rows, dtype=torch.int32).to(device)


def conv_gemm(x: torch.Tensor, w: torch.Tensor, taps: torch.Tensor, *, out_dtype: torch.dtype = torch.bfloat16, bias: torch.Tensor | None = None,
              residual: torch.Tensor | None = None, zb: int = 1) -> torch.Tensor:
    """Implicit-GEMM "same" convolution: x bf16 channels-last [b, *spatial, c] (c % 8 == 0), w bf16 [n, ld] with features (tap, channel), ``taps``
    from :func:`conv_tap_table` -> rows [b * prod(spatial), n] (+ bias, + fp32 residual); the im2col matrix is never materialised."""
    _dev(x, w, taps, bias, residual)
    if x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or not x.is_contiguous() or taps.dtype != torch.int32 or taps.dim() != 2 or taps.shape[1] != 4 or taps.shape[0] < (w.shape[1] + 63) // 64 * 8:
        raise HipLibraryError("conv_gemm: contiguous bf16 volume, bf16 weights [n, ld], int32 [ld / 8, 4] tap table")
    b, c = x.shape[0], x.shape[-1]
    sp = (1,) * (3 - (x.dim() - 2)) + tuple(x.shape[1:-1])
    m, n = b * sp[0] * sp[1] * sp[2] // zb, w.shape[0]  # zb > 1: one row per group of zb z-voxels, n = zb * c_out (w, bias: conv_weight_zblock)
    out = _empty((m, n), dtype=torch.float32 if residual is not None else out_dtype, device=x.device)
    g = GemmArgs()
    g.a, g.b, g.d = x.data_ptr(), w.data_ptr(), out.data_ptr()
    g.m, g.n, g.k, g.lda, g.ldb, g.ldd = m, n, w.shape[1], 0, _rowmajor(w, "w"), n
    g.a_kmajor, g.b_kmajor, g.alpha, g.split_k = 1, 1, 1.0, 1
    g.conv_taps, g.conv_x, g.conv_y, g.conv_z, g.conv_c, g.conv_zb = taps.data_ptr(), sp[0], sp[1], sp[2], c, zb
    if bias is not None:
        g.bias = bias.data_ptr()
    if residual is not None:
        if residual.dtype != torch.float32 or not residual.is_contiguous() or residual.numel() != m * n:
            raise HipLibraryError("conv_gemm: contiguous fp32 residual of the output's size")
        g.residual_f32, g.ld_res = residual.data_ptr(), n
    g.out_f32 = int(out.dtype == torch.float32)
    _check(load().cinema_conv_gemm_bf16(C.byref(g), _stream()), "conv_gemm")
    return out


def conv_coord_table(batch: int, spatial: tuple, device: torch.device, zb: int = 1) -> torch.Tensor:
    """int32 [batch * prod(spatial) / zb]: x | y << 10 | z << 20 of every voxel row of a channels-last volume (2-D: leading unit axis); ``zb`` > 1: of the
    first voxel of every group of zb consecutive z voxels."""
    sp = (1,) * (3 - len(spatial)) + tuple(int(v) for v in spatial)
    if max(sp) > 1023 or sp[2] % zb:
        raise HipLibraryError("conv_coord_table: extents up to 1023, Z % zb == 0")
    x = torch.arange(sp[0], dtype=torch.int32)[:, None, None]
    y = torch.arange(sp[1], dtype=torch.int32)[None, :, None]
    z = torch.arange(0, sp[2], zb, dtype=torch.int32)[None, None, :]
    return (x | (y << 10) | (z << 20)).reshape(-1).repeat(batch).contiguous().to(device)


def conv_wgrad(dy: torch.Tensor, x: torch.Tensor, taps: torch.Tensor, coords: torch.Tensor, out: torch.Tensor, split_k: int, a_rowsum: torch.Tensor | None = None,
               zb: int = 1, accumulate: bool = True) -> None:
    """out [c_out, ld] fp32 += dy^T im2col(x) without materialising im2col(x): dy bf16 [rows, c_out], x bf16 channels-last [b, *spatial, c],
    ``taps`` the forward tap table of the weight layout, ``coords`` from :func:`conv_coord_table`; a_rowsum [c_out] += column sums of dy."""
    _dev(dy, x, taps, coords, out, a_rowsum)
    if dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or out.dtype != torch.float32 or not x.is_contiguous() or coords.dtype != torch.int32:
        raise HipLibraryError("conv_wgrad: bf16 operands, fp32 destination, int32 coordinates")
    rows, c_out = dy.shape  # zb > 1: dy is the [voxel rows / zb, zb * c_out] view of the gradient rows, coords / taps the z-blocked tables, out [zb * c_out, 9 (zb + 2) c]
    c = x.shape[-1]
    sp = (1,) * (3 - (x.dim() - 2)) + tuple(x.shape[1:-1])
    if coords.numel() != rows or x.numel() // c != rows * zb or out.shape[0] != c_out:
        raise HipLibraryError("conv_wgrad: shape mismatch")
    g = GemmArgs()
    g.a, g.b, g.d = dy.data_ptr(), x.data_ptr(), out.data_ptr()
    g.m, g.n, g.k, g.lda, g.ldb, g.ldd = c_out, out.shape[1], rows, _rowmajor(dy, "dy"), 0, _rowmajor(out, "out")
    g.a_kmajor, g.b_kmajor, g.alpha, g.out_f32, g.accumulate, g.split_k = 0, 0, 1.0, 1, int(accumulate), split_k
    g.conv_taps, g.conv_x, g.conv_y, g.conv_z, g.conv_c, g.conv_coords, g.conv_zb = taps.data_ptr(), sp[0], sp[1], sp[2], c, coords.data_ptr(), zb
    ws = _workspace("splitk", split_k * c_out * out.shape[1], dy.device)
    g.workspace, g.workspace_bytes = ws.data_ptr(), split_k * c_out * out.shape[1] * 4
    if a_rowsum is not None:
        g.a_rowsum = a_rowsum.data_ptr()
    _check(load().cinema_conv_wgrad_bf16(C.byref(g), _stream()), "conv_wgrad")


def conv_weight_zblock(w: torch.Tensor, c: int, zb: int, transpose: bool, bias: torch.Tensor | None = None) -> tuple:
    """Block-banded operand of the z-blocked convolution: w bf16 [n, ld >= 27 c] (features (tap, channel)) -> (bf16 [zb n, 9 (zb + 2) c], bias repeated zb
    times or None); ``transpose``: for the data-gradient operand (taps pointing the other way)."""
    _dev(w, bias)
    if w.dtype != torch.bfloat16 or w.dim() != 2 or w.stride(1) != 1 or w.shape[1] < 27 * c or (bias is not None and (bias.dtype != torch.float32 or bias.numel() != w.shape[0])):
        raise HipLibraryError("conv_weight_zblock: bf16 rows [n, >= 27 c], fp32 bias [n]")
    n = w.shape[0]
    out = _empty((zb * n, 9 * (zb + 2) * c), dtype=torch.bfloat16, device=w.device)
    b_out = None if bias is None else _empty(zb * n, dtype=torch.float32, device=w.device)
    _check(load().cinema_conv_weight_zblock(w.data_ptr(), n, c, w.stride(0), zb, int(transpose), out.data_ptr(), _p(bias), _p(b_out), _stream()), "conv_weight_zblock")
    return out, b_out


def conv_wgrad_zfold(r: torch.Tensor, n: int, c: int, zb: int, dst: torch.Tensor, rowsum_zb: torch.Tensor | None = None, db: torch.Tensor | None = None) -> None:
    """dst fp32 [n, ld >= 27 c] += the zb bands of the z-blocked weight gradient r fp32 [zb n, 9 (zb + 2) c]; db [n] += the zb segments of rowsum_zb."""
    _dev(r, dst, rowsum_zb, db)
    if r.dtype != torch.float32 or dst.dtype != torch.float32 or tuple(r.shape) != (zb * n, 9 * (zb + 2) * c) or not r.is_contiguous() or dst.shape[0] != n or dst.stride(1) != 1:
        raise HipLibraryError("conv_wgrad_zfold: fp32 r [zb n, 9 (zb + 2) c] and dst [n, ld]")
    _check(load().cinema_conv_wgrad_zfold(r.data_ptr(), n, c, zb, dst.data_ptr(), dst.stride(0), _p(rowsum_zb), _p(db), _stream()), "conv_wgrad_zfold")


def conv_weight_dgrad(w: torch.Tensor) -> torch.Tensor:
    """Conv weight fp32 (c_out, c_in, *k) -> bf16 [c_in, ld] with features (tap, c_out), ld = taps * c_out rounded up to 8 (data-gradient operand)."""
    _dev(w)
    if w.dtype != torch.float32 or not w.is_contiguous():
        raise HipLibraryError("conv_weight_dgrad: contiguous fp32 weight")
    c_out, c_in = w.shape[0], w.shape[1]
    kvol = w[0, 0].numel()
    ld = (kvol * c_out + 7) // 8 * 8
    rows = _empty((c_in, ld), dtype=torch.bfloat16, device=w.device)
    _check(load().cinema_conv_weight_dgrad(w.data_ptr(), rows.data_ptr(), c_out, c_in, kvol, ld, _stream()), "conv_weight_dgrad")
    return rows


def im2col(x: torch.Tensor, ks: tuple) -> torch.Tensor:
    """x bf16 channels-last [b, *spatial, c] -> cols bf16 [b*prod(spatial), ld], ld = taps*c rounded up to 8 (zero tail)."""
    _dev(x)
    if x.dtype != torch.bfloat16 or not x.is_contiguous():
        raise HipLibraryError("im2col: x must be contiguous bf16 channels-last")
    b, X, Y, Z, c, kx, ky, kz = _vol_dims(tuple(x.shape), ks)  # noqa: N806
    ld = (kx * ky * kz * c + 7) // 8 * 8
    cols = _empty((b * X * Y * Z, ld), dtype=torch.bfloat16, device=x.device)
    _check(load().cinema_im2col(x.data_ptr(), cols.data_ptr(), ld, b, X, Y, Z, c, kx, ky, kz, _stream()), "im2col")
    return cols


def col2im(dcols: torch.Tensor, shape: tuple, ks: tuple) -> torch.Tensor:
    """Data gradient of :func:`im2col`: dcols bf16 [b*prod(spatial), ld] -> dx bf16 [b, *spatial, c]."""
    _dev(dcols)
    b, X, Y, Z, c, kx, ky, kz = _vol_dims(tuple(shape), ks)  # noqa: N806
    dx = _empty(shape, dtype=torch.bfloat16, device=dcols.device)
    _check(load().cinema_col2im(dcols.data_ptr(), _rowmajor(dcols, "dcols"), dx.data_ptr(), b, X, Y, Z, c, kx, ky, kz, _stream()), "col2im")
    return dx
