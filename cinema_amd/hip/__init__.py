"""ctypes binding of ``libcinema_hip.so`` (C-ABI in ``include/cinema_hip.h``).

Thin, typed launchers: every function takes torch tensors that already live on the GPU, checks
dtype / contiguity / device, and enqueues one or two HIP kernels on torch's *current* stream.
There is deliberately no CPU or ATen fallback: without the library, or with a CPU tensor, the
call raises.
"""

from __future__ import annotations

import ctypes as C
import struct
import os
from pathlib import Path

import torch

_LIB_PATH = Path(__file__).resolve().parent.parent / "libcinema_hip.so"
_lib = None

BF16, F32 = 0, 1
_DT = {torch.bfloat16: BF16, torch.float32: F32}


class HipLibraryError(RuntimeError):
    """Raised when the HIP kernel library is missing or a kernel launch is rejected."""


class GemmArgs(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("b", C.c_void_p), ("d", C.c_void_p),
        ("m", C.c_int), ("n", C.c_int), ("k", C.c_int),
        ("lda", C.c_int), ("ldb", C.c_int), ("ldd", C.c_int),
        ("a_kmajor", C.c_int), ("b_kmajor", C.c_int),
        ("alpha", C.c_float),
        ("bias", C.c_void_p), ("residual_f32", C.c_void_p), ("residual_bf16", C.c_void_p), ("ld_res", C.c_int),
        ("gelu_in", C.c_void_p), ("ld_gelu", C.c_int),
        ("row_mask", C.c_void_p),
        ("aux_out", C.c_void_p), ("ld_aux", C.c_int),
        ("act", C.c_int), ("gelu_deriv", C.c_int), ("out_f32", C.c_int), ("accumulate", C.c_int), ("split_k", C.c_int), ("force_generic", C.c_int),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_longlong), ("a_rowsum", C.c_void_p),
        ("scale_a", C.c_void_p), ("scale_b", C.c_void_p), ("scale_a_rows", C.c_int),
        ("conv_taps", C.c_void_p), ("conv_x", C.c_int), ("conv_y", C.c_int), ("conv_z", C.c_int), ("conv_c", C.c_int), ("conv_coords", C.c_void_p), ("conv_zb", C.c_int),
        ("out8", C.c_void_p), ("ld_out8", C.c_int), ("out8_inv_scale", C.c_void_p), ("out8_amax", C.c_void_p),
        ("colsum_partials", C.c_void_p),
        ("tail_counters", C.c_void_p),
        ("kernel_used", C.c_int),
    ]


class Q8Out(C.Structure):
    """Mirror of ``cinema_q8_out``."""

    _fields_ = [("data", C.c_void_p), ("inv_scale", C.c_void_p), ("colsum", C.c_void_p), ("amax_slots", C.c_void_p)]


class Q8Site:
    """One tensor position of the model with an 8-bit copy under per-tensor DELAYED scaling (``cinema_q8_out``): views into the site arrays of
    ``cinema_amd.tape.Fp8Sites`` - ``scale`` fp32 [1] (dequantisation multiplier, read by the consuming GEMMs), ``inv`` fp32 [1], ``amax`` int32 [CINEMA_Q8_SLOTS] (this step's
    maximum, float bits).  ``ready``: a scale derived from a recorded maximum exists (one step after the site first ran); until then producers record only."""

    __slots__ = ("scale", "inv", "amax", "owner", "born")

    def __init__(self, scale: torch.Tensor, inv: torch.Tensor, amax: torch.Tensor, owner, born: int) -> None:  # noqa: ANN001
        self.scale, self.inv, self.amax, self.owner, self.born = scale, inv, amax, owner, born

    @property
    def ready(self) -> bool:
        return self.owner.updates > self.born

    def out(self, data: torch.Tensor | None, colsum: torch.Tensor | None = None) -> Q8Out:
        return Q8Out(None if data is None else data.data_ptr(), self.inv.data_ptr(), None if colsum is None else colsum.data_ptr(), self.amax.data_ptr())


# cinema_gemm_args.kernel_used -> kernel name as rocprofv3 prints it: 0 generic, otherwise
# operand layout (1: A,B k-major = forward; 2: B n-major = data gradient; 3: both strided = small weight gradients) + 8 x epilogue class
# (0 general, 1 bf16, 2 bf16 + GELU, 3 bf16 x GELU', 4 fp32 (+ residual)), see csrc/gemm.hip
_LAYOUTS = {1: "true, true", 2: "true, false", 3: "false, false"}
GEMM_KERNEL_NAMES = {0: "gemm_generic_kernel"}
GEMM_KERNEL_NAMES[64] = "gemm_mfma_grouped_kernel<false, false, 4>"
GEMM_KERNEL_NAMES.update({128 + lay + 8 * epi: f"gemm_mfma_k32_kernel<{txt}, {epi}, 3>" for lay, txt in _LAYOUTS.items() for epi in range(5)})
GEMM_KERNEL_NAMES.update({lay + 8 * epi: f"gemm_mfma_kernel<{txt}, {epi}>" for lay, txt in _LAYOUTS.items() for epi in range(5)})
# main-loop form of the persistent kernel (csrc/gemm256.hip reads the same variable per call): 2 = LDS-DMA issued by the reading wave (default), 1 = between the MFMAs, 0 = k-tile loop
_P256_LOOP = int(os.environ.get("CINEMA_P256_LOOP", "2"))
GEMM_KERNEL_NAMES.update({2048 + lay + 8 * epi: f"gemm_p256_kernel<{txt}, {epi}, {min(_P256_LOOP, 2)}>" for lay, txt in _LAYOUTS.items() for epi in range(5)})  # csrc/gemm256.hip
GEMM_KERNEL_NAMES[4096 + 3 + 8 * 4] = f"gemm_p256_kernel<false, false, 4, {10 if _P256_LOOP >= 2 else 3}>"  # weight gradients on e4m3 operands (cinema_gemm_fp8_wgrad_p256)
# bench.py sets this to a list to time every GEMM launch with HIP events on the launch stream: entries are (kernel_used, algorithmic_flops, start_event,
# end_event, (m, n, k, a_kmajor, b_kmajor, split_k | problems, algorithmic_bytes), the launch's problems as (m, n, k, a_kmajor, b_kmajor) each)
GEMM_PROFILE: list | None = None


class PatchGeom(C.Structure):
    _fields_ = [
        ("b", C.c_int), ("c", C.c_int), ("gx", C.c_int), ("gy", C.c_int), ("gz", C.c_int),
        ("px", C.c_int), ("py", C.c_int), ("pz", C.c_int),
        ("sb", C.c_longlong), ("sc", C.c_longlong), ("sx", C.c_longlong), ("sy", C.c_longlong), ("sz", C.c_longlong),
        ("n_rows", C.c_int), ("token_idx", C.c_void_p),
    ]


class RowCopyArgs(C.Structure):
    """Mirror of ``cinema_row_copy_args``."""

    _fields_ = [("dst", C.c_void_p), ("dst_dtype", C.c_int), ("ld_dst", C.c_int), ("dst_idx", C.c_void_p),
                ("src", C.c_void_p), ("src_dtype", C.c_int), ("ld_src", C.c_int), ("src_idx", C.c_void_p),
                ("add", C.c_void_p), ("add_dtype", C.c_int), ("ld_add", C.c_int), ("add_idx", C.c_void_p),
                ("n_rows", C.c_int), ("c", C.c_int), ("accumulate", C.c_int)]


class LnReduceItem(C.Structure):
    """Mirror of ``cinema_ln_reduce_item``."""

    _fields_ = [("partials", C.c_void_p), ("n_partials", C.c_int), ("c", C.c_int), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dcol", C.c_void_p)]


class SparseGeom(C.Structure):
    """Mirror of ``cinema_sparse_geom``: visible-voxel (token-major compact row) geometry of one stem stage."""

    _fields_ = [
        ("b", C.c_int), ("tx", C.c_int), ("ty", C.c_int), ("tz", C.c_int), ("bx", C.c_int), ("by", C.c_int), ("bz", C.c_int), ("n_tok", C.c_int),
        ("keep", C.c_void_p), ("rank", C.c_void_p), ("pos", C.c_void_p),
    ]


class StemWgradProblem(C.Structure):
    """Mirror of ``cinema_stem_wgrad_problem``."""

    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("rows", C.c_int), ("n", C.c_int), ("k", C.c_int)]


_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong
_PROTOS = {
    "cinema_hip_info": [C.POINTER(C.c_int)],
    "cinema_gemm_bf16": [C.POINTER(GemmArgs), _vp],
    "cinema_gemm_bf16_grouped": [C.POINTER(GemmArgs), _i, _vp],
    "cinema_gemm_bf16_p256": [C.POINTER(GemmArgs), _i, _i, _vp, _ll, _vp],
    "cinema_gemm_fp8_wgrad_p256": [C.POINTER(GemmArgs), _i, _vp, _ll, _vp],
    "cinema_gemm_p256_workspace_bytes": [],
    "cinema_gemm_fp8": [C.POINTER(GemmArgs), _vp],
    "cinema_conv_gemm_bf16": [C.POINTER(GemmArgs), _vp],
    "cinema_conv_wgrad_bf16": [C.POINTER(GemmArgs), _vp],
    "cinema_conv_weight_dgrad": [_vp, _vp, _i, _i, _i, _i, _vp],
    "cinema_quantize_fp8": [_vp, _ll, _vp, _vp, _vp, _vp],
    "cinema_quantize_fp8_rows": [_vp, _i, _i, _vp, _vp, _vp],
    "cinema_fp8_sites_update": [_vp, _vp, _vp, _i, _f, _vp],
    "cinema_quantize_fp8_site": [_vp, _ll, C.POINTER(Q8Out), _vp],
    "cinema_dequantize_fp8": [_vp, _ll, _vp, _vp, _vp],
    "cinema_quantize_fp8_site_colsum": [_vp, _i, _i, _i, C.POINTER(Q8Out), _vp, _vp],
    "cinema_layernorm_fwd_q8": [_vp, _i, _i, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _i, _vp, _vp, C.POINTER(Q8Out), _vp],
    "cinema_layernorm_bwd_deferred_q8": [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _ll, C.POINTER(C.c_int), C.POINTER(Q8Out), _vp],
    "cinema_layernorm_fwd_fp8": [_vp, _i, _i, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "cinema_quantize_fp8_segments": [_vp, _vp, _i, _vp, _vp, _vp, _vp],
    "cinema_quantize_fp8_segments_t": [_vp, _vp, _i, _vp, _vp, _vp],
    "cinema_colsum": [_vp, _i, _vp, _i, _i, _i, _vp, _vp],
    "cinema_layernorm_fwd": [_vp, _i, _i, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _i, _vp, _vp, _vp],
    "cinema_layernorm_bwd": [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _ll, _vp],
    "cinema_layernorm_bwd_deferred": [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _ll, C.POINTER(C.c_int), _vp],
    "cinema_layernorm_bwd_workspace_bytes": [_i, _i],
    "cinema_ln_param_reduce_batched": [_vp, _i, _vp],
    "cinema_row_copy_multi": [_vp, _i, _vp],
    "cinema_attention_fwd": [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "cinema_attention_bwd": [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "cinema_attention_bwd_workspace_bytes": [_i, _i, _i, _i, _i],
    "cinema_attention_bwd_ws": [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _ll, _vp, _i, _vp],
    "cinema_dwconv_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cinema_dwconv_bwd_data": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cinema_dwconv_bwd_weight": [_vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cinema_im2col": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cinema_col2im": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cinema_sparse_nbr_ints": [_i],
    "cinema_sparse_nbr_build": [C.POINTER(SparseGeom), _i, _i, _i, _vp, _vp, _vp],
    "cinema_sparse_dwconv_fwd": [_vp, _vp, _vp, _vp, C.POINTER(SparseGeom), _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "cinema_sparse_dwconv_bwd_weight": [_vp, _vp, _vp, _vp, _vp, _ll, C.POINTER(SparseGeom), _i, _i, _i, _i, _vp, _vp],
    "cinema_sparse_halo_ints": [C.POINTER(SparseGeom), _i, _i, _i],
    "cinema_sparse_halo_index": [C.POINTER(SparseGeom), _i, _i, _i, _vp, _vp],
    "cinema_sparse_dwconv_wgrad_workspace_bytes": [_i, _i, _i, _i, _i],
    "cinema_stem_dw_supported": [C.POINTER(SparseGeom), _i, _i, _i, _i],
    "cinema_stem_dw_fwd": [_vp, _vp, _vp, _vp, C.POINTER(SparseGeom), _i, _i, _i, _i, _i, _vp],
    "cinema_stem_dw_wgrad_workspace_bytes": [_i, _i, _i, _i, _i],
    "cinema_stem_dw_bwd_weight": [_vp, _vp, _vp, _vp, _vp, _ll, C.POINTER(SparseGeom), _i, _i, _i, _i, _vp],
    "cinema_stem_supported": [_i],
    "cinema_stem_partials": [_i],
    "cinema_stem_ln_linear": [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "cinema_stem_mlp_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "cinema_stem_mlp_bwd": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(C.c_int), _vp],
    "cinema_stem_ln_linear_bwd": [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, C.POINTER(C.c_int), _vp],
    "cinema_stem_wgrad_slices": [_i],
    "cinema_stem_wgrad_workspace_bytes": [C.POINTER(StemWgradProblem), _i],
    "cinema_stem_wgrad": [C.POINTER(StemWgradProblem), _i, _vp, _ll, _vp],
    "cinema_patch_gather": [_vp, _i, _vp, _i, _i, C.POINTER(PatchGeom), _vp],
    "cinema_patch_scatter": [_vp, _i, _i, _vp, _i, _i, C.POINTER(PatchGeom), _vp],
    "cinema_row_copy": [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp],
    "cinema_seg_loss_fwd": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "cinema_seg_loss_bwd": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "cinema_head_ce": [_vp, _vp, _i, _i, _f, _vp, _vp, _vp],
    "cinema_conv_weight_zblock": [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "cinema_conv_wgrad_zfold": [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp],
    "cinema_head_mse": [_vp, _vp, _i, _vp, _vp, _vp],
    "cinema_seg_window_accumulate": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "cinema_seg_window_finish": [_vp, _vp, _i, _ll, _vp, _vp],
    "cinema_seg_metric_counts": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "cinema_mask_edges": [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "cinema_min_dist": [_vp, _vp, _i, _i, _vp, _vp],
    "cinema_segment_mean_fwd": [_vp, _i, _i, _i, _i, _f, _vp, _vp],
    "cinema_segment_mean_bwd": [_vp, _i, _i, _i, _f, _vp, _i, _i, _vp],
    "cinema_scale_f32": [_vp, _f, _vp, _ll, _vp],
    "cinema_fill_u32": [_vp, C.c_uint, _ll, _vp],
    "cinema_thin_linear_fwd": [_vp, _vp, _vp, _vp, _ll, _i, _i, _vp],
    "cinema_thin_linear_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp],
    "cinema_fanout_linear_fwd": [_vp, _vp, _vp, _vp, _ll, _i, _i, _vp],
    "cinema_fanout_linear_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp],
    "cinema_convt_weight_relayout": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "cinema_conv1ch_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cinema_conv1ch_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cinema_rng_advance": [_vp, _vp],
    "cinema_dropout_bf16": [_vp, _vp, _ll, _f, _vp, C.c_uint, _vp],
    "cinema_droppath_scale": [_vp, _i, _f, _vp, C.c_uint, _vp],
    "cinema_scale_rows_add": [_vp, _vp, _vp, _vp, _ll, _i, _i, _vp],
    "cinema_scale_rows_bf16": [_vp, _vp, _vp, _ll, _i, _i, _vp],
    "cinema_rope_heads": [_vp, _i, _ll, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "cinema_mul_scalar_f32": [_vp, _vp, _vp, _ll, _vp],
    "cinema_mul_rows": [_vp, _i, _vp, _i, _i, _vp, _i, _ll, _i, _vp],
    "cinema_mask_select": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "cinema_visible_index": [_vp, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), _vp, _vp, _vp, _vp],
    "cinema_stream_fork": [_vp, _vp],
    "cinema_lanes_begin": [_i],
    "cinema_lanes_select": [_i],
    "cinema_lanes_end": [_vp, _vp],
    "cinema_lanes_abort": [],
    "cinema_marker_record": [_vp],
    "cinema_marker_done": [_ll],
    "cinema_launch_probe": [_i, _vp],
    "cinema_mfma_probe": [_i, _i, _vp, _vp],
    "cinema_patch_weight_relayout": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "cinema_cast": [_vp, _i, _vp, _i, _ll, _vp],
    "cinema_transpose_cast": [_vp, _i, _i, _i, _vp, _vp],
    "cinema_gelu_fwd": [_vp, _vp, _ll, _vp],
    "cinema_gelu_bwd": [_vp, _vp, _vp, _ll, _vp],
    "cinema_zoom_resample": [_vp, _i, _i, _i, _f, _f, _f, _i, _vp, _vp, _vp],
    "cinema_scale_intensity_pad": [_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp],
    "cinema_mse_fwd": [_vp, C.POINTER(PatchGeom), _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp],
    "cinema_mse_bwd": [_vp, C.POINTER(PatchGeom), _vp, _i, _i, _i, _f, _vp, _f, _vp, _i, _vp],
    "cinema_patch_stats": [_vp, C.POINTER(PatchGeom), _vp, _vp],
    "cinema_mean_finite": [_vp, _i, _vp, _vp, _vp],
    "cinema_sqnorm_f32": [_vp, _ll, _vp, _vp, _vp],
    "cinema_clip_coef": [_vp, _f, _vp, _vp, _vp, _vp],
    "cinema_adamw": [_vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp],
    "cinema_adamw_groups": [_vp, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _vp, _vp, _vp, _vp],
    "cinema_kernel_launch_count": [],
    "cinema_adamw_groups_grid": [_vp, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _vp, _vp, _vp, _i, _vp],
}
EXPORTED_SYMBOLS = tuple(_PROTOS)
FORCE_GENERIC = bool(int(os.environ.get("CINEMA_HIP_FORCE_GENERIC", "0")))


def library_path() -> Path:
    return _LIB_PATH


# Call recording (cinema_amd/replay.py): while RECORD is a list every launch through this module is appended to it as (cfunc, args) right
# after it ran.  The arguments are plain ints / floats / ctypes structs, so the same launch can be issued again verbatim; host-only queries
# (workspace sizes) and the completion markers are not part of a step's launch list.
RECORD: list | None = None
_NOT_REPLAYED = ("cinema_kernel_launch_count", "cinema_stem_dw_supported", "cinema_stem_supported", "cinema_stem_partials", "cinema_stem_wgrad_slices", "_workspace_bytes", "_nbr_ints", "_halo_ints", "cinema_marker_record", "cinema_marker_done", "cinema_launch_probe", "cinema_mfma_probe", "cinema_lanes_abort")


class _Entry:
    """One exported function of the C-ABI: call-through, plus the append to RECORD when a recording is active."""

    __slots__ = ("fn", "replayed")

    def __init__(self, fn, replayed: bool) -> None:  # noqa: ANN001
        self.fn, self.replayed = fn, replayed

    def __call__(self, *args):  # noqa: ANN002, ANN204
        rc = self.fn(*args)
        if RECORD is not None and self.replayed:
            RECORD.append((self.fn, args))
        return rc


class _Library:
    def __init__(self, cdll) -> None:  # noqa: ANN001
        self.cdll = cdll

    def __getattr__(self, name: str):  # noqa: ANN204
        return getattr(self.cdll, name)  # symbols outside _PROTOS (dev builds)


def load():  # noqa: ANN201
    """Load the shared library (once). Raises :class:`HipLibraryError` if it was not built."""
    global _lib  # noqa: PLW0603
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise HipLibraryError(
            f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). cinema_amd has no CPU/ATen fallback."
        )
    lib = _Library(C.CDLL(str(_LIB_PATH)))
    for name, argtypes in _PROTOS.items():
        fn = getattr(lib.cdll, name)
        fn.argtypes = argtypes
        fn.restype = C.c_longlong if name.endswith(("_workspace_bytes", "_nbr_ints", "_halo_ints", "_marker_record", "_launch_count")) else C.c_int
        setattr(lib, name, _Entry(fn, not name.endswith(_NOT_REPLAYED)))
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        kind = {-1: "bad argument", -2: "unsupported shape/alignment"}.get(rc, f"hipError {rc}")
        raise HipLibraryError(f"{what} failed: {kind}")


_raw_current_stream = torch._C._cuda_getCurrentRawStream  # (device index) -> hipStream_t as int; ~0.2 us, torch.cuda.current_stream() costs ~8 us
_current_device = torch._C._cuda_getDevice
_STREAM_OVERRIDE: int | None = None


def _stream() -> int:
    """The stream every launch goes to: torch's current stream of the current device, or the override set by :func:`on_stream`."""
    return _STREAM_OVERRIDE if _STREAM_OVERRIDE is not None else _raw_current_stream(_current_device())


class on_stream:  # noqa: N801
    """``with on_stream(raw_handle):`` sends the launches of this module to another stream without touching torch's current stream
    (used for the side-stream weight-gradient launches; their scratch comes from :func:`_workspace`, which is per stream, and their
    outputs are caller-owned, so nothing is allocated under the wrong stream).  One launching thread at a time."""

    def __init__(self, raw: int) -> None:
        self.raw = raw

    def __enter__(self) -> None:
        global _STREAM_OVERRIDE  # noqa: PLW0603
        self.prev, _STREAM_OVERRIDE = _STREAM_OVERRIDE, self.raw

    def __exit__(self, *exc) -> None:  # noqa: ANN002
        global _STREAM_OVERRIDE  # noqa: PLW0603
        _STREAM_OVERRIDE = self.prev


def _empty(*args, **kw) -> torch.Tensor:  # noqa: ANN002, ANN003
    """torch.empty; inside a lane group the buffer is held until the group's deferred launches have been issued (the caching allocator would
    otherwise hand a dropped temporary of one lane to the next lane while the first lane's kernels have not even been launched)."""
    t = torch.empty(*args, **kw)
    if LANE is not None:
        _LANE_KEEP.append(t)
    return t


def _empty_like(x: torch.Tensor, **kw) -> torch.Tensor:  # noqa: ANN003
    t = torch.empty_like(x, **kw)
    if LANE is not None:
        _LANE_KEEP.append(t)
    return t


empty, empty_like = _empty, _empty_like


def stream_fork(from_stream: int, to_stream: int) -> None:
    _check(load().cinema_stream_fork(from_stream, to_stream), "stream_fork")


def marker_record(stream: int) -> int:
    ticket = load().cinema_marker_record(stream)
    if ticket < 0:
        raise HipLibraryError(f"marker_record failed: {ticket}")
    return ticket


def marker_done(ticket: int) -> bool:
    rc = load().cinema_marker_done(ticket)
    if rc < 0:
        raise HipLibraryError(f"marker_done({ticket}) failed: {rc}")
    return rc == 1


def _p(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _dev(*ts: torch.Tensor | None) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise HipLibraryError("cinema_amd kernels need GPU (HIP) tensors; got a CPU tensor. There is no CPU fallback.")
    if LANE is not None:  # launches are deferred to the end of the lane group: nothing they touch may go back to the allocator before that
        _LANE_KEEP.extend(t for t in ts if t is not None)


# ---- lane groups (include/cinema_hip.h: cinema_lanes_*): independent, identically shaped launch sequences merged into wide launches -------------
LANE: int | None = None      # lane being recorded, None outside a group
LANES_ENABLED = True  # (tests/test_lanes_gpu.py compares against one launch per lane)
LANE_STATS = [0, 0]          # merged / single launches issued by the lane groups so far (diagnostics)
_LANE_KEEP: list = []
# when set, a closing lane group hands the buffers it held over to this list instead of dropping them: its launches went to a stream the caching allocator
# does not associate with those buffers, so they must outlive the JOIN of that stream, not merely the issue of the launches (tape.lane_group(stream=...))
LANE_KEEP_SINK: list | None = None


class lanes:  # noqa: N801
    """``with lanes(n) as g: g.select(0); <launches of lane 0>; g.select(1); ...``: the launches issued inside are recorded by the library and
    go out, zipped across the lanes, when the block ends.  The lanes must be independent.  Inactive (plain immediate launches) when
    ``CINEMA_LANES=0`` or when a group is already open.  (While ``GEMM_PROFILE`` times single launches the group stays active and the launches
    inside it are not timed: the timed set is then exactly the launches rocprofv3 reports under the single-launch kernel names.)"""

    def __init__(self, n: int) -> None:
        self.n = n
        self.active = LANES_ENABLED and LANE is None and 2 <= n <= 4

    def __enter__(self) -> "lanes":
        global LANE  # noqa: PLW0603
        if self.active:
            _check(load().cinema_lanes_begin(self.n), "lanes_begin")
            LANE = 0
        return self

    def select(self, lane: int) -> None:
        global LANE  # noqa: PLW0603
        if self.active:
            _check(load().cinema_lanes_select(lane), "lanes_select")
            LANE = lane

    def __exit__(self, *exc) -> None:  # noqa: ANN002
        global LANE  # noqa: PLW0603
        if self.active:
            LANE = None
            m, s1 = C.c_int(0), C.c_int(0)
            rc = load().cinema_lanes_end(C.byref(m), C.byref(s1))
            LANE_STATS[0] += m.value
            LANE_STATS[1] += s1.value
            if LANE_KEEP_SINK is not None:
                LANE_KEEP_SINK.extend(_LANE_KEEP)
            _LANE_KEEP.clear()
            if exc[0] is None:
                _check(rc, "lanes_end")


def lanes_abort() -> None:
    """Close whatever lane group is open without issuing its launches (error paths of callers that open and close a group in separate steps)."""
    global LANE, _STREAM_OVERRIDE, LANE_KEEP_SINK  # noqa: PLW0603
    LANE = None
    _STREAM_OVERRIDE, LANE_KEEP_SINK = None, None  # (a backward lane group on a stream of its own redirects the launches between its two closures)
    _LANE_KEEP.clear()
    if _lib is not None:
        _lib.cinema_lanes_abort()


def _rowmajor(t: torch.Tensor, name: str) -> int:
    if t.dim() == 2 and t.shape[1] == 1:  # a single column: the inner stride is meaningless (torch may report anything for it)
        return t.stride(0)
    if t.dim() != 2 or t.stride(1) != 1:
        raise HipLibraryError(f"{name} must be 2-D with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.stride(0)


_WORKSPACES: dict = {}
_RETIRED_WORKSPACES: list = []


def _workspace(tag: str, n_floats: int, device: torch.device) -> torch.Tensor:
    """fp32 scratch per (tag, device, stream), grown on demand and kept: every use is one stream-ordered kernel sequence (producer ->
    reduce / fix-up), so a single buffer per stream serves all launches on it, and a launch redirected by :func:`on_stream` never
    borrows memory the allocator believes to belong to torch's current stream.  Outgrown buffers are parked, not freed (a kernel on the
    other stream may still be reading them; there are only a handful of growth steps per process)."""
    key = (tag, device.index, _stream(), LANE)  # per lane inside a lane group: the merged launch runs the lanes' kernels side by side
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < n_floats:
        if ws is not None:
            _RETIRED_WORKSPACES.append(ws)
        ws = _WORKSPACES[key] = _empty(n_floats, dtype=torch.float32, device=device)
    return ws


def _tail_workspace(device: torch.device) -> torch.Tensor:
    """32 MiB: 512 workgroup slots x one 128x128 fp32 partial tile (split-tail GEMM -> fix-up kernel)."""
    return _workspace("tail", 512 * 128 * 128, device)


_TAIL_COUNTERS: dict = {}
# 1 (default since round 4): the split tail of the 128x128 GEMM is finished inside the launch - every k-slice of a tail tile publishes its partial tile and then
# sums and finishes ITS share of the tile (reduce-scatter over the slices, csrc/gemm.hip tail_finish_in_launch) - instead of the fix-up launch.  Bit-identical
# results.  Round 3's form (ONE workgroup, the last arriver, read up to 15 x 64 KiB) was slower than the fix-up launch; this one is time-neutral per shape and in
# the step (profiles/r04_b_*: 27.31 / 27.13 ms with the fix-up launch, 27.01 / 27.16 without) and removes 117 launches per step.  0: the fix-up launch.
TAIL_IN_LAUNCH = True
TAIL_MIN_K = 768  # shortest reduction that gets split-tail scratch (the library decides per shape)


def _tail_counters(device: torch.device) -> torch.Tensor:
    """Arrival / publish counters of the split-tail tiles (csrc/gemm.hip), one buffer per (device, stream, lane): zero at allocation, left zero by every
    launch, never handed back to the allocator (``persistent``)."""
    key = (device.index, _stream(), LANE)
    t = _TAIL_COUNTERS.get(key)
    if t is None:
        t = _TAIL_COUNTERS[key] = persistent(lambda: torch.zeros(2048, dtype=torch.int32, device=device))
    return t


_P256_WS: dict = {}


def _p256_workspace(device: torch.device) -> torch.Tensor:
    """Counters + fp32 partial slots of the persistent 256x256 GEMM (csrc/gemm256.hip), one per (device, stream, lane): the counters are zero at
    allocation and every launch leaves them zero, so the buffer must never be handed back to the allocator (``persistent``: outside a recording's pool)."""
    key = (device.index, _stream(), LANE)
    ws = _P256_WS.get(key)
    if ws is None:
        n = load().cinema_gemm_p256_workspace_bytes()
        ws = _P256_WS[key] = persistent(lambda: torch.zeros(n // 4, dtype=torch.float32, device=device))
    return ws


P256_ERROR_WORD = 65536 // 4 - 1  # csrc/gemm256.hip: last word of the 64 KiB counter head (P_COUNTER_BYTES)
TAIL_ERROR_WORD = 2047            # csrc/gemm_shared.cuh


def check_reduction_workspaces() -> None:
    """Read the error word of every in-launch-reduction workspace of this process (persistent 256x256 GEMM, split-tail counters): a workgroup that waited
    for a partial tile longer than its bounded spin (2^26 polls) gives up, finishes with what it has and sets the word - the results of that launch are then
    WRONG.  Never observed in a healthy run; a hung or reset neighbour queue is the scenario.  One small device->host read (a synchronisation): the training
    steps call this every ``check_every`` updates and before a checkpoint is written.  On a set word the counter regions are zeroed again (so that later
    launches start from a clean state) and :class:`HipLibraryError` is raised."""
    words, owners = [], []
    for key, ws in _P256_WS.items():
        words.append(ws[P256_ERROR_WORD:P256_ERROR_WORD + 1].view(torch.int32))
        owners.append(("persistent GEMM workspace", key, ws))
    for key, t in _TAIL_COUNTERS.items():
        words.append(t[TAIL_ERROR_WORD:TAIL_ERROR_WORD + 1])
        owners.append(("split-tail counters", key, t))
    if not words:
        return
    by_dev: dict = {}
    for w, o in zip(words, owners):
        by_dev.setdefault(w.device, []).append((w, o))
    bad = []
    for items in by_dev.values():
        vals = torch.cat([w for w, _ in items]).tolist()
        bad += [o for v, (_, o) in zip(vals, items) if v != 0]
    if bad:
        for what, _key, t in bad:
            (t[:P256_ERROR_WORD + 1] if what.startswith("persistent") else t).zero_()
        raise HipLibraryError("an in-launch split reduction gave up waiting for a partial tile (error word set): the gradients of that launch are wrong - "
                              + "; ".join(f"{what} of (device, stream, lane) {key}" for what, key, _ in bad) + ". The counters were reset; restart from the last checkpoint.")


def persistent(fn):  # noqa: ANN001, ANN201
    """Run ``fn()`` - which builds a device tensor whose CONTENT must survive for the life of the process (index tables, RNG state) - outside a recording's
    private memory pool.  While a step is being recorded (cinema_amd/replay.py) the caching allocator hands freed blocks of the pool out again: a table built
    at first use could land on the address of an earlier temporary, and every replay of the launches that wrote that temporary would overwrite the table
    (seen as a memory fault of the implicit convolution reading a clobbered tap table).  ``torch.cuda.use_mem_pool`` routes only the CURRENT thread's
    allocations, so the build runs on a helper thread."""
    if RECORD is None:
        out = fn()
        # built by torch ops on torch's CURRENT stream, but its first reader may be a launch that on_stream() redirected to another stream (the zeroed counters
        # of a weight-gradient stream's first split-K GEMM): finish the build before anyone is handed the tensor.  First use only - one host wait per table
        if torch.cuda.is_available() and not torch._C._cuda_isCurrentStreamCapturing():
            torch.cuda.current_stream().synchronize()
        return out
    import threading

    box: list = []
    dev = torch.cuda.current_device() if torch.cuda.is_available() else None

    def work() -> None:
        try:
            if dev is not None:
                torch.cuda.set_device(dev)
            box.append(fn())
            if dev is not None:
                torch.cuda.synchronize()  # built on this thread's default stream: complete before any consumer on the recording's streams
        except BaseException as e:  # noqa: BLE001
            box.append(e)

    th = threading.Thread(target=work)
    th.start()
    th.join()
    if isinstance(box[0], BaseException):
        raise box[0]
    return box[0]


SPARSE_WGRAD_PIPE = True  # False: the per-token index chase (the form the kernel tests compare against)
# the depthwise conv of the visible-voxel stem as a walk over neighbour TOKENS (csrc/stem_dw.hip; 64 / 128 channels, 4x4 / 2x2 token blocks); 0: the per-voxel neighbour
# lists of csrc/sparse_conv.hip, which stay the form for every other geometry
STEM_DW_PAIR = True  # False: the neighbour-list kernels for every geometry (the form the kernel tests compare against)


ROW_COPY_MULTI = True


def kernel_launch_count() -> int:
    """Kernels the library has launched since it was loaded (merged lane-group launches count once; stream forks are not kernels)."""
    return int(load().cinema_kernel_launch_count())


def info() -> dict:
    out = (C.c_int * 8)()
    _check(load().cinema_hip_info(out), "info")
    return {"abi_version": out[0], "n_cus": out[1], "lds_bytes_per_block": out[2], "wave_size": out[3]}


# ---- the kernel families (one module per csrc/ source group), re-exported: callers write hip.<name> ----------------------------------------------------
from cinema_amd.hip.gemm import *  # noqa: E402, F401, F403
from cinema_amd.hip.loss import *  # noqa: E402, F401, F403
from cinema_amd.hip.rows import *  # noqa: E402, F401, F403
from cinema_amd.hip.norm import *  # noqa: E402, F401, F403
from cinema_amd.hip.attention import *  # noqa: E402, F401, F403
from cinema_amd.hip.conv import *  # noqa: E402, F401, F403
from cinema_amd.hip.stem import *  # noqa: E402, F401, F403
from cinema_amd.hip.update import *  # noqa: E402, F401, F403
