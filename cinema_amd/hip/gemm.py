"""GEMM family (csrc/gemm.hip, csrc/gemm256.hip): bf16 / e4m3 GEMMs with fused epilogues, grouped weight gradients, e4m3 quantisation, thin / fan-out linears.

Part of the ctypes front of ``libcinema_hip.so`` (see ``cinema_amd/hip/__init__.py`` for the loader, the launch recorder, lane groups and the per-stream
workspaces); everything here is re-exported there, so callers keep writing ``hip.<name>``.  Module-level switches and the recorder state live in the package and
are read through it (``H.<NAME>``) so that an assignment ``hip.<NAME> = ...`` is seen by every family."""
from __future__ import annotations

import ctypes as C

import torch

from cinema_amd import hip as H
from cinema_amd.hip import (  # noqa: F401
    GemmArgs, HipLibraryError, Q8Site, _check, _dev, _empty, _p, _p256_workspace, _rowmajor, _stream, _tail_counters, _tail_workspace, _workspace,
    load,
)

__all__ = ['_WARNED_GENERIC', '_gelu_deriv_mode', '_p256_call', '_set_colsum_partials', '_set_out8', '_warn_generic', 'dequantize_fp8', 'fanout_linear_bwd', 'fanout_linear_fwd', 'fanout_ok', 'fp8_sites_update', 'gemm', 'gemm_fp8', 'gemm_fp8_wgrad_grouped', 'gemm_wgrad_grouped', 'quantize_fp8', 'quantize_fp8_rows', 'quantize_fp8_segments', 'quantize_fp8_segments_t', 'quantize_fp8_site', 'quantize_fp8_site_colsum', 'thin_linear_bwd', 'thin_linear_fwd']


def _p256_call(arr, count: int, schedule: int, device: torch.device) -> None:  # noqa: ANN001
    ws = _p256_workspace(device)
    _check(load().cinema_gemm_bf16_p256(arr, count, schedule, ws.data_ptr(), ws.numel() * 4, _stream()), "gemm_p256")


# --------------------------------------------------------------------------------------------------------
def _set_out8(g: GemmArgs, out8: tuple, m: int, n: int) -> None:
    site, data = out8
    if data is not None:
        _dev(data)
        if data.dtype != torch.uint8 or tuple(data.shape) != (m, n):
            raise HipLibraryError("out8 must be uint8 [M, N]")
        g.out8, g.ld_out8 = data.data_ptr(), _rowmajor(data, "out8")
    g.out8_inv_scale, g.out8_amax = site.inv.data_ptr(), site.amax.data_ptr()


def _set_colsum_partials(g: GemmArgs, ws: torch.Tensor, m: int, n: int) -> None:
    _dev(ws)
    if ws.dtype != torch.float32 or not ws.is_contiguous() or tuple(ws.shape) != ((m + 31) // 32, n):
        raise HipLibraryError("colsum_partials must be dense fp32 [ceil(M / 32), N]")
    g.colsum_partials = ws.data_ptr()


def gemm(a: torch.Tensor, b: torch.Tensor, *, a_kmajor: bool = True, b_kmajor: bool = True, out: torch.Tensor | None = None,
         out_dtype: torch.dtype = torch.bfloat16, bias: torch.Tensor | None = None, residual: torch.Tensor | None = None,
         gelu_in: torch.Tensor | None = None, row_mask: torch.Tensor | None = None, aux_out: torch.Tensor | None = None,
         act: int = 0, accumulate: bool = False, split_k: int = 1, alpha: float = 1.0, force_generic: bool = False,
         a_rowsum: torch.Tensor | None = None, p256: int | None = None, gelu_deriv: bool = False, out8: tuple | None = None,
         colsum_partials: torch.Tensor | None = None) -> torch.Tensor:
    """``out8`` = (Q8Site, uint8 [M, N] | None): 8-bit copy of a bf16 result with the site's delayed scale (None: record the maximum only).
    D = epilogue(alpha * A @ B).  ``a``: [M,K] if a_kmajor else [K,M];  ``b``: [N,K] if b_kmajor else [K,N].
    ``gelu_deriv``: the auxiliary GELU tensor holds GELU'(pre-activation) - written to ``aux_out`` by an ``act=1`` launch, multiplied in from ``gelu_in``.
    ``p256`` = 0 / 1: the persistent 256x256 kernel with its split / stream schedule (``split_k`` = 1 then means whole-K tiles, 0 balanced slices)."""
    lib = load()
    _dev(a, b, out, bias, residual, gelu_in, row_mask, aux_out)
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16:
        raise HipLibraryError("gemm operands must be bf16")
    lda, ldb = _rowmajor(a, "a"), _rowmajor(b, "b")
    m, k = (a.shape[0], a.shape[1]) if a_kmajor else (a.shape[1], a.shape[0])
    n, kb = (b.shape[0], b.shape[1]) if b_kmajor else (b.shape[1], b.shape[0])
    if k != kb:
        raise HipLibraryError(f"gemm reduction mismatch: {k} vs {kb}")
    if out is None:
        out = _empty((m, n), dtype=out_dtype, device=a.device)
    elif tuple(out.shape) != (m, n):
        raise HipLibraryError(f"gemm out shape {tuple(out.shape)} != {(m, n)}")
    g = GemmArgs()
    g.a, g.b, g.d = a.data_ptr(), b.data_ptr(), out.data_ptr()
    g.m, g.n, g.k, g.lda, g.ldb, g.ldd = m, n, k, lda, ldb, _rowmajor(out, "out")
    g.a_kmajor, g.b_kmajor, g.alpha = int(a_kmajor), int(b_kmajor), alpha
    if bias is not None:
        if bias.dtype != torch.float32 or bias.numel() != n:
            raise HipLibraryError("gemm bias must be fp32 [n]")
        g.bias = bias.data_ptr()
    if residual is not None:
        g.ld_res = _rowmajor(residual, "residual")
        if residual.dtype == torch.float32:
            g.residual_f32 = residual.data_ptr()
        elif residual.dtype == torch.bfloat16:
            g.residual_bf16 = residual.data_ptr()
        else:
            raise HipLibraryError("gemm residual must be fp32 or bf16")
    if gelu_in is not None:
        g.gelu_in, g.ld_gelu = gelu_in.data_ptr(), _rowmajor(gelu_in, "gelu_in")
    if row_mask is not None:
        if row_mask.dtype not in (torch.uint8, torch.bool) or row_mask.numel() != m:
            raise HipLibraryError("gemm row_mask must be uint8/bool [m]")
        g.row_mask = row_mask.data_ptr()
    if aux_out is not None:
        g.aux_out, g.ld_aux = aux_out.data_ptr(), _rowmajor(aux_out, "aux_out")
    g.act, g.out_f32, g.accumulate = act, int(out.dtype == torch.float32), int(accumulate)
    g.gelu_deriv = _gelu_deriv_mode(gelu_deriv, aux_out, gelu_in)
    g.split_k, g.force_generic = split_k, int(force_generic or H.FORCE_GENERIC)
    if a_rowsum is not None:  # fp32 [m], accumulated: sum_k A[m, k] (bias gradient of a weight-gradient GEMM)
        _dev(a_rowsum)
        g.a_rowsum = a_rowsum.data_ptr()
    if out8 is not None:
        _set_out8(g, out8, m, n)
    if colsum_partials is not None:
        _set_colsum_partials(g, colsum_partials, m, n)
    if p256 is not None:
        if H.GEMM_PROFILE is None or H.LANE is not None:
            _p256_call(C.byref(g), 1, p256, a.device)
            return out
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        _p256_call(C.byref(g), 1, p256, a.device)
        ev1.record()
        extra = sum(t.numel() * t.element_size() for t in (residual, gelu_in, aux_out) if t is not None)
        alg_bytes = 2.0 * (m * k + k * n) + out.element_size() * m * n * (2 if accumulate else 1) + extra
        H.GEMM_PROFILE.append((g.kernel_used, 2.0 * m * n * k, ev0, ev1, (m, n, k, int(a_kmajor), int(b_kmajor), 0, alg_bytes), ((m, n, k, int(a_kmajor), int(b_kmajor)),)))
        return out
    ws = None
    if split_k > 1 and out.dtype == torch.float32:  # deterministic two-pass split-K: per-split fp32 slabs + one reduce kernel
        ws = _workspace("splitk", split_k * m * n, a.device)
        g.workspace, g.workspace_bytes = ws.data_ptr(), split_k * m * n * 4
    elif split_k == 1 and k >= H.TAIL_MIN_K:  # split-tail scratch (k-slices of the tiles left over after the last full round of workgroup slots)
        ws = _tail_workspace(a.device)
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        if H.TAIL_IN_LAUNCH:
            g.tail_counters = _tail_counters(a.device).data_ptr()
    if H.GEMM_PROFILE is None or H.LANE is not None:
        _check(lib.cinema_gemm_bf16(C.byref(g), _stream()), "gemm")
        if g.kernel_used == 0 and not g.force_generic and 2.0 * m * n * k > 1e9:
            _warn_generic(m, n, k, a, b, out)
        return out
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    _check(lib.cinema_gemm_bf16(C.byref(g), _stream()), "gemm")
    ev1.record()
    ob = out.element_size()
    extra = sum(t.numel() * t.element_size() for t in (residual, gelu_in, aux_out) if t is not None)
    alg_bytes = 2.0 * (m * k + k * n) + ob * m * n * (2 if accumulate else 1) + extra  # every operand read once, the result written once
    H.GEMM_PROFILE.append((g.kernel_used, 2.0 * m * n * k, ev0, ev1, (m, n, k, int(a_kmajor), int(b_kmajor), split_k, alg_bytes), ((m, n, k, int(a_kmajor), int(b_kmajor)),)))
    return out


_WARNED_GENERIC: set = set()


def _gelu_deriv_mode(gelu_deriv: bool, aux_out: torch.Tensor | None, gelu_in: torch.Tensor | None) -> int:
    """``cinema_gemm_args.gelu_deriv``: 0 = the auxiliary GELU tensor is the bf16 pre-activation, 1 = bf16 GELU'(pre-activation), 2 = GELU' as the 8-bit affine
    code of csrc/common.cuh (uint8 tensors; only with ``gelu_deriv``)."""
    t = aux_out if aux_out is not None else gelu_in
    if t is not None and t.dtype == torch.uint8:
        if not gelu_deriv:
            raise HipLibraryError("a uint8 auxiliary GELU tensor holds the 8-bit code of GELU': pass gelu_deriv=True")
        return 2
    return int(gelu_deriv)


def quantize_fp8(x: torch.Tensor) -> tuple:
    """Per-tensor e4m3 quantisation of a contiguous bf16 matrix: -> (uint8 tensor of the same shape, fp32 [1] dequantisation scale = amax / 448)."""
    _dev(x)
    if x.dtype != torch.bfloat16 or not x.is_contiguous() or x.numel() % 8:
        raise HipLibraryError("quantize_fp8: contiguous bf16 with a multiple of 8 elements")
    y = _empty(x.shape, dtype=torch.uint8, device=x.device)
    scale = _empty(1, dtype=torch.float32, device=x.device)
    ws = _workspace("fp8_amax", 4, x.device)
    _check(load().cinema_quantize_fp8(x.data_ptr(), x.numel(), y.data_ptr(), scale.data_ptr(), ws.data_ptr(), _stream()), "quantize_fp8")
    return y, scale


def quantize_fp8_site(x: torch.Tensor, site: Q8Site) -> tuple | None:
    """Stand-alone producer of an 8-bit copy under the site's delayed per-tensor scale: -> (uint8 tensor of x's shape, site.scale), or None while the site has
    no scale yet (this launch then only records max|x|).  One pass, no maximum pre-pass."""
    _dev(x)
    if x.dtype != torch.bfloat16 or not x.is_contiguous() or x.numel() % 8:
        raise HipLibraryError("quantize_fp8_site: contiguous bf16 with a multiple of 8 elements")
    y = _empty(x.shape, dtype=torch.uint8, device=x.device) if site.ready else None
    q = site.out(y)
    _check(load().cinema_quantize_fp8_site(x.data_ptr(), x.numel(), C.byref(q), _stream()), "quantize_fp8_site")
    return None if y is None else (y, site.scale)


def dequantize_fp8(q8: tuple) -> torch.Tensor:
    """bf16 tensor of an (e4m3 bytes, per-tensor scale [1]) pair (an 8-bit-only output that a consumer outside the e4m3 GEMMs asks for)."""
    y8, sc = q8
    _dev(y8, sc)
    if y8.dtype != torch.uint8 or not y8.is_contiguous() or y8.numel() % 8 or sc.numel() != 1:
        raise HipLibraryError("dequantize_fp8: contiguous uint8 with a multiple of 8 elements, one fp32 scale")
    y = _empty(y8.shape, dtype=torch.bfloat16, device=y8.device)
    _check(load().cinema_dequantize_fp8(y8.data_ptr(), y8.numel(), sc.data_ptr(), y.data_ptr(), _stream()), "dequantize_fp8")
    return y


def quantize_fp8_site_colsum(x: torch.Tensor, site: Q8Site, colsum_out: torch.Tensor) -> tuple | None:
    """:func:`quantize_fp8_site` of a bf16 matrix [rows, c] plus ``colsum_out[c] += column sums of x`` from the same pass (a gradient tensor: its 8-bit copy is
    the dY operand of the weight gradient, its column sums are the bias gradient)."""
    _dev(x, colsum_out)
    if x.dtype != torch.bfloat16 or x.dim() != 2 or x.shape[1] % 8 or colsum_out.dtype != torch.float32 or colsum_out.numel() != x.shape[1] or not colsum_out.is_contiguous():
        raise HipLibraryError("quantize_fp8_site_colsum: bf16 [rows, c] with c % 8 == 0, fp32 [c] sums")
    y = _empty(x.shape, dtype=torch.uint8, device=x.device) if site.ready else None
    q = site.out(y)
    _check(load().cinema_quantize_fp8_site_colsum(x.data_ptr(), x.shape[0], x.shape[1], _rowmajor(x, "x"), C.byref(q), colsum_out.data_ptr(), _stream()),
           "quantize_fp8_site_colsum")
    return None if y is None else (y, site.scale)


def fp8_sites_update(amax: torch.Tensor, scale: torch.Tensor, inv: torch.Tensor, n_sites: int, margin: float) -> None:
    _dev(amax, scale, inv)
    _check(load().cinema_fp8_sites_update(amax.data_ptr(), scale.data_ptr(), inv.data_ptr(), n_sites, margin, _stream()), "fp8_sites_update")


def quantize_fp8_rows(x: torch.Tensor) -> tuple:
    """Per-row e4m3 quantisation of a contiguous bf16 matrix [rows, c]: -> (uint8 [rows, c], fp32 [rows] scales); one launch."""
    _dev(x)
    if x.dtype != torch.bfloat16 or x.dim() != 2 or not x.is_contiguous() or x.shape[1] % 8:
        raise HipLibraryError("quantize_fp8_rows: contiguous bf16 [rows, c] with c % 8 == 0")
    y = _empty(x.shape, dtype=torch.uint8, device=x.device)
    scale = _empty(x.shape[0], dtype=torch.float32, device=x.device)
    _check(load().cinema_quantize_fp8_rows(x.data_ptr(), x.shape[0], x.shape[1], y.data_ptr(), scale.data_ptr(), _stream()), "quantize_fp8_rows")
    return y, scale


def quantize_fp8_segments(x: torch.Tensor, seg_bounds: torch.Tensor, y: torch.Tensor, scales: torch.Tensor) -> None:
    """Segments [seg_bounds[i, 0], seg_bounds[i, 1]) of the flat bf16 buffer ``x`` -> e4m3 in ``y`` (uint8, same layout), one scale per segment."""
    _dev(x, seg_bounds, y, scales)
    if x.dtype != torch.bfloat16 or y.dtype != torch.uint8 or seg_bounds.dtype != torch.int64 or scales.dtype != torch.float32 or not seg_bounds.is_contiguous():
        raise HipLibraryError("quantize_fp8_segments: bf16 source, uint8 destination, int64 [n, 2] bounds, fp32 scales")
    n = seg_bounds.shape[0]
    ws = _workspace("fp8_amax_seg", n, x.device)
    _check(load().cinema_quantize_fp8_segments(x.data_ptr(), seg_bounds.data_ptr(), n, y.data_ptr(), scales.data_ptr(), ws.data_ptr(), _stream()),
           "quantize_fp8_segments")


def quantize_fp8_segments_t(x: torch.Tensor, seg_desc: torch.Tensor, scales: torch.Tensor, yt: torch.Tensor) -> None:
    """Transposed e4m3 copies of the 2-D segments of the flat bf16 buffer ``x``: seg_desc int64 [n, 3] = (offset, rows, cols); yt uint8, [cols][rows] per segment
    at the same offsets, scaled with ``scales`` (from :func:`quantize_fp8_segments` on the same buffer)."""
    _dev(x, seg_desc, scales, yt)
    if x.dtype != torch.bfloat16 or yt.dtype != torch.uint8 or seg_desc.dtype != torch.int64 or not seg_desc.is_contiguous() or scales.dtype != torch.float32:
        raise HipLibraryError("quantize_fp8_segments_t: bf16 source, uint8 destination, int64 [n, 3] descriptors, fp32 scales")
    _check(load().cinema_quantize_fp8_segments_t(x.data_ptr(), seg_desc.data_ptr(), seg_desc.shape[0], scales.data_ptr(), yt.data_ptr(), _stream()),
           "quantize_fp8_segments_t")


def gemm_fp8(a8: torch.Tensor, scale_a: torch.Tensor, b8: torch.Tensor, scale_b: torch.Tensor, *, out_dtype: torch.dtype = torch.bfloat16,
             bias: torch.Tensor | None = None, residual: torch.Tensor | None = None, aux_out: torch.Tensor | None = None, act: int = 0,
             alpha: float = 1.0, out: torch.Tensor | None = None, gelu_in: torch.Tensor | None = None, gelu_deriv: bool = False,
             out8: tuple | None = None, colsum_partials: torch.Tensor | None = None, skip_d: bool = False) -> torch.Tensor | None:
    """D = epilogue(alpha * scale_a * scale_b * A8 @ B8^T): e4m3 operands a8 [M, K], b8 [N, K] (uint8 storage, k-major), per-tensor fp32 [1] scales;
    bias / exact GELU (+ bf16 pre-activation copy) / fp32 residual epilogue like :func:`gemm`."""
    _dev(a8, scale_a, b8, scale_b, bias, residual, aux_out)
    if a8.dtype != torch.uint8 or b8.dtype != torch.uint8 or a8.shape[1] != b8.shape[1]:
        raise HipLibraryError("gemm_fp8: uint8 (e4m3) operands [M, K] and [N, K]")
    m, k = a8.shape
    n = b8.shape[0]
    if skip_d:  # ``skip_d``: only the 8-bit copy of the (bf16) result is wanted (``out8`` with a buffer): no bf16 tensor is written or returned
        if out8 is None or out8[1] is None or residual is not None or out_dtype != torch.bfloat16:
            raise HipLibraryError("gemm_fp8(skip_d=True) needs out8 with a buffer and a bf16 result")
        out = None
    elif out is None:
        out = _empty((m, n), dtype=torch.float32 if residual is not None else out_dtype, device=a8.device)
    elif tuple(out.shape) != (m, n):
        raise HipLibraryError(f"gemm_fp8 out shape {tuple(out.shape)} != {(m, n)}")
    _dev(out)
    g = GemmArgs()
    g.a, g.b, g.d = a8.data_ptr(), b8.data_ptr(), (None if out is None else out.data_ptr())
    g.m, g.n, g.k, g.lda, g.ldb, g.ldd = m, n, k, _rowmajor(a8, "a8"), _rowmajor(b8, "b8"), (n if out is None else _rowmajor(out, "out"))
    g.a_kmajor, g.b_kmajor, g.alpha, g.split_k = 1, 1, alpha, 1
    g.scale_a, g.scale_b = scale_a.data_ptr(), scale_b.data_ptr()
    if scale_a.numel() not in (1, m) or scale_b.numel() != 1 or scale_a.dtype != torch.float32 or scale_b.dtype != torch.float32:
        raise HipLibraryError("gemm_fp8: scale_a fp32 [1] or [M] (per row), scale_b fp32 [1]")
    g.scale_a_rows = int(scale_a.numel() == m and m > 1)
    if bias is not None:
        g.bias = bias.data_ptr()
    if residual is not None:
        if residual.dtype != torch.float32:
            raise HipLibraryError("gemm_fp8: fp32 residual only")
        g.residual_f32, g.ld_res = residual.data_ptr(), _rowmajor(residual, "residual")
    if aux_out is not None:
        g.aux_out, g.ld_aux = aux_out.data_ptr(), _rowmajor(aux_out, "aux_out")
    if gelu_in is not None:  # D = (A8 B8^T) x GELU'(gelu_in): the data gradient through fc1's activation
        _dev(gelu_in)
        g.gelu_in, g.ld_gelu = gelu_in.data_ptr(), _rowmajor(gelu_in, "gelu_in")
    g.act, g.out_f32, g.gelu_deriv = act, int(out is not None and out.dtype == torch.float32), _gelu_deriv_mode(gelu_deriv, aux_out, gelu_in)
    if out8 is not None:
        _set_out8(g, out8, m, n)
    if colsum_partials is not None:
        _set_colsum_partials(g, colsum_partials, m, n)
    _check(load().cinema_gemm_fp8(C.byref(g), _stream()), "gemm_fp8")
    return out


def gemm_wgrad_grouped(problems: list, p256: bool = False, split_k: int = 0) -> None:
    """One launch for up to 8 weight gradients: each problem is (dy [rows, n_out] bf16, x [rows, k_out] bf16, dst fp32 [n_out, k_out] view,
    a_rowsum fp32 [n_out] | None); dst += dy^T x, a_rowsum += column sums of dy.  Whole-K 128x128 tiles, no split-K slabs (see the header);
    ``p256``: the persistent 256x256 kernel with balanced k-slices finished inside the launch (the problems may then differ in their row counts;
    ``split_k = 1`` keeps whole-K tiles there, for A/B measurements)."""
    arr = (GemmArgs * len(problems))()
    for g, (dy, x, dst, rowsum) in zip(arr, problems):
        _dev(dy, x, dst, rowsum)
        if dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or dst.dtype != torch.float32 or dy.shape[0] != x.shape[0]:
            raise HipLibraryError("gemm_wgrad_grouped: bf16 operands with a common row count, fp32 destination")
        g.a, g.b, g.d = dy.data_ptr(), x.data_ptr(), dst.data_ptr()
        g.m, g.n, g.k = dy.shape[1], x.shape[1], dy.shape[0]
        g.lda, g.ldb, g.ldd = _rowmajor(dy, "dy"), _rowmajor(x, "x"), _rowmajor(dst, "dst")
        g.a_kmajor, g.b_kmajor, g.alpha, g.out_f32, g.accumulate, g.split_k = 0, 0, 1.0, 1, 1, (split_k if p256 else 1)
        if rowsum is not None:
            g.a_rowsum = rowsum.data_ptr()
    if p256:
        if H.GEMM_PROFILE is None or H.LANE is not None:
            _p256_call(arr, len(problems), 0, problems[0][0].device)
            return
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        _p256_call(arr, len(problems), 0, problems[0][0].device)
        ev1.record()
        flops = sum(2.0 * g.m * g.n * g.k for g in arr)
        alg = sum(2.0 * (g.m * g.k + g.k * g.n) + 8.0 * g.m * g.n for g in arr)  # every operand once, the fp32 gradient read + written once
        H.GEMM_PROFILE.append((arr[0].kernel_used, flops, ev0, ev1, (sum(g.m for g in arr), arr[0].n, arr[0].k, 0, 0, len(problems), alg), tuple((g.m, g.n, g.k, 0, 0) for g in arr)))
        return
    if H.GEMM_PROFILE is None:
        _check(load().cinema_gemm_bf16_grouped(arr, len(problems), _stream()), "gemm_grouped")
        return
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    _check(load().cinema_gemm_bf16_grouped(arr, len(problems), _stream()), "gemm_grouped")
    ev1.record()
    flops = sum(2.0 * g.m * g.n * g.k for g in arr)
    alg = sum(2.0 * (g.m * g.k + g.k * g.n) + 8.0 * g.m * g.n for g in arr)
    H.GEMM_PROFILE.append((64, flops, ev0, ev1, (sum(g.m for g in arr), arr[0].n, arr[0].k, 0, 0, len(problems), alg), tuple((g.m, g.n, g.k, 0, 0) for g in arr)))


def gemm_fp8_wgrad_grouped(problems: list) -> None:
    """Weight gradients on 8-bit operands in ONE persistent launch (``cinema_gemm_fp8_wgrad_p256``): each problem is (dy8 uint8 [rows, n_out], scale_dy fp32 [1],
    x8 uint8 [rows, k_out], scale_x fp32 [1], dst fp32 [n_out, k_out] view); dst += scale_dy * scale_x * dy8^T x8 (e4m3 decode).  The operands are the row-major
    [token][feature] copies the producing kernels write - no transposed copies; bias gradients are not part of this launch (:func:`colsum`)."""
    arr = (GemmArgs * len(problems))()
    for g, (dy, sdy, x, sx, dst) in zip(arr, problems):
        _dev(dy, sdy, x, sx, dst)
        if dy.dtype != torch.uint8 or x.dtype != torch.uint8 or dst.dtype != torch.float32 or dy.shape[0] != x.shape[0] or sdy.dtype != torch.float32 or sx.dtype != torch.float32:
            raise HipLibraryError("gemm_fp8_wgrad_grouped: uint8 (e4m3) operands with a common row count, fp32 [1] scales, fp32 destination")
        g.a, g.b, g.d = dy.data_ptr(), x.data_ptr(), dst.data_ptr()
        g.m, g.n, g.k = dy.shape[1], x.shape[1], dy.shape[0]
        g.lda, g.ldb, g.ldd = _rowmajor(dy, "dy8"), _rowmajor(x, "x8"), _rowmajor(dst, "dst")
        g.a_kmajor, g.b_kmajor, g.alpha, g.out_f32, g.accumulate, g.split_k = 0, 0, 1.0, 1, 1, 0
        g.scale_a, g.scale_b = sdy.data_ptr(), sx.data_ptr()
    dev = problems[0][0].device
    ws = _p256_workspace(dev)
    if H.GEMM_PROFILE is None or H.LANE is not None:
        _check(load().cinema_gemm_fp8_wgrad_p256(arr, len(problems), ws.data_ptr(), ws.numel() * 4, _stream()), "gemm_fp8_wgrad_p256")
        return
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    _check(load().cinema_gemm_fp8_wgrad_p256(arr, len(problems), ws.data_ptr(), ws.numel() * 4, _stream()), "gemm_fp8_wgrad_p256")
    ev1.record()
    flops = sum(2.0 * g.m * g.n * g.k for g in arr)
    alg = sum(1.0 * (g.m * g.k + g.k * g.n) + 8.0 * g.m * g.n for g in arr)  # every 8-bit operand once, the fp32 gradient read + written once
    H.GEMM_PROFILE.append((arr[0].kernel_used, flops, ev0, ev1, (sum(g.m for g in arr), arr[0].n, arr[0].k, 0, 0, len(problems), alg), tuple((g.m, g.n, g.k, 0, 0) for g in arr)))


def _warn_generic(m: int, n: int, k: int, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor) -> None:
    """A large GEMM on the generic FMA kernel is ~20x slower than the MFMA kernel and almost always an alignment accident (16-byte
    pointers, leading dimensions / n / k multiples of 8): say so once per shape instead of being silently slow."""
    key = (m, n, k)
    if key not in _WARNED_GENERIC:
        _WARNED_GENERIC.add(key)
        import warnings

        warnings.warn(f"cinema_gemm_bf16 {m}x{n}x{k} ran on the generic (non-MFMA) kernel: operand pointers a/b/out % 16 = "
                      f"{a.data_ptr() % 16}/{b.data_ptr() % 16}/{out.data_ptr() % 16}, strides {a.stride()}/{b.stride()}/{out.stride()}", stacklevel=3)


def thin_linear_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """y fp32 [rows, n] = x (bf16 [rows, k]) @ w^T (fp32 [n, k]) + bias for n <= 8, k <= 64 (streaming kernel, no GEMM)."""
    _dev(x, w, bias)
    if x.dtype != torch.bfloat16 or w.dtype != torch.float32 or not x.is_contiguous() or not w.is_contiguous():
        raise HipLibraryError("thin_linear: contiguous bf16 rows, fp32 weight")
    y = _empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
    _check(load().cinema_thin_linear_fwd(x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), x.shape[0], w.shape[0], w.shape[1], _stream()), "thin_linear_fwd")
    return y


def thin_linear_bwd(x: torch.Tensor, w: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor | None, db: torch.Tensor | None, want_dx: bool) -> torch.Tensor | None:
    """Backward of :func:`thin_linear_fwd`: returns dx (bf16) when asked; dw (fp32 [n, k]) / db (fp32 [n]) are accumulated in place."""
    _dev(x, w, dy, dw, db)
    if dy.dtype != torch.float32 or not dy.is_contiguous() or tuple(dy.shape) != (x.shape[0], w.shape[0]):
        raise HipLibraryError("thin_linear_bwd: contiguous fp32 dy [rows, n]")
    dx = _empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_dx else None
    _check(load().cinema_thin_linear_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), _p(dx), _p(dw), _p(db), x.shape[0], w.shape[0], w.shape[1], _stream()),
           "thin_linear_bwd")
    return dx


def fanout_ok(n: int, k: int) -> bool:
    return 1 <= k <= 8 and n in (4, 8, 16, 32, 64)


def fanout_linear_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """y fp32 [rows, n] = x (bf16 [rows, k]) @ w^T (fp32 [n, k]) + bias for k <= 8, n in {4, 8, 16, 32, 64} (streaming kernel, no GEMM)."""
    _dev(x, w, bias)
    if x.dtype != torch.bfloat16 or w.dtype != torch.float32 or not x.is_contiguous() or not w.is_contiguous() or x.shape[1] != w.shape[1]:
        raise HipLibraryError("fanout_linear: contiguous bf16 rows [rows, k], fp32 weight [n, k]")
    y = _empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
    _check(load().cinema_fanout_linear_fwd(x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), x.shape[0], w.shape[0], w.shape[1], _stream()), "fanout_linear_fwd")
    return y


def fanout_linear_bwd(x: torch.Tensor, w: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor | None, db: torch.Tensor | None, want_dx: bool) -> torch.Tensor | None:
    """Backward of :func:`fanout_linear_fwd`: returns dx (bf16) when asked; dw (fp32 [n, k]) / db (fp32 [n]) are accumulated in place."""
    _dev(x, w, dy, dw, db)
    if dy.dtype != torch.float32 or not dy.is_contiguous() or tuple(dy.shape) != (x.shape[0], w.shape[0]):
        raise HipLibraryError("fanout_linear_bwd: contiguous fp32 dy [rows, n]")
    dx = _empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_dx else None
    _check(load().cinema_fanout_linear_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), _p(dx), _p(dw), _p(db), x.shape[0], w.shape[0], w.shape[1], _stream()),
           "fanout_linear_bwd")
    return dx
