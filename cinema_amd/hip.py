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

_LIB_PATH = Path(__file__).resolve().parent / "libcinema_hip.so"
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
    g.split_k, g.force_generic = split_k, int(force_generic or FORCE_GENERIC)
    if a_rowsum is not None:  # fp32 [m], accumulated: sum_k A[m, k] (bias gradient of a weight-gradient GEMM)
        _dev(a_rowsum)
        g.a_rowsum = a_rowsum.data_ptr()
    if out8 is not None:
        _set_out8(g, out8, m, n)
    if colsum_partials is not None:
        _set_colsum_partials(g, colsum_partials, m, n)
    if p256 is not None:
        if GEMM_PROFILE is None or LANE is not None:
            _p256_call(C.byref(g), 1, p256, a.device)
            return out
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        _p256_call(C.byref(g), 1, p256, a.device)
        ev1.record()
        extra = sum(t.numel() * t.element_size() for t in (residual, gelu_in, aux_out) if t is not None)
        alg_bytes = 2.0 * (m * k + k * n) + out.element_size() * m * n * (2 if accumulate else 1) + extra
        GEMM_PROFILE.append((g.kernel_used, 2.0 * m * n * k, ev0, ev1, (m, n, k, int(a_kmajor), int(b_kmajor), 0, alg_bytes), ((m, n, k, int(a_kmajor), int(b_kmajor)),)))
        return out
    ws = None
    if split_k > 1 and out.dtype == torch.float32:  # deterministic two-pass split-K: per-split fp32 slabs + one reduce kernel
        ws = _workspace("splitk", split_k * m * n, a.device)
        g.workspace, g.workspace_bytes = ws.data_ptr(), split_k * m * n * 4
    elif split_k == 1 and k >= TAIL_MIN_K:  # split-tail scratch (k-slices of the tiles left over after the last full round of workgroup slots)
        ws = _tail_workspace(a.device)
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        if TAIL_IN_LAUNCH:
            g.tail_counters = _tail_counters(a.device).data_ptr()
    if GEMM_PROFILE is None or LANE is not None:
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
    GEMM_PROFILE.append((g.kernel_used, 2.0 * m * n * k, ev0, ev1, (m, n, k, int(a_kmajor), int(b_kmajor), split_k, alg_bytes), ((m, n, k, int(a_kmajor), int(b_kmajor)),)))
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
        if GEMM_PROFILE is None or LANE is not None:
            _p256_call(arr, len(problems), 0, problems[0][0].device)
            return
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        _p256_call(arr, len(problems), 0, problems[0][0].device)
        ev1.record()
        flops = sum(2.0 * g.m * g.n * g.k for g in arr)
        alg = sum(2.0 * (g.m * g.k + g.k * g.n) + 8.0 * g.m * g.n for g in arr)  # every operand once, the fp32 gradient read + written once
        GEMM_PROFILE.append((arr[0].kernel_used, flops, ev0, ev1, (sum(g.m for g in arr), arr[0].n, arr[0].k, 0, 0, len(problems), alg), tuple((g.m, g.n, g.k, 0, 0) for g in arr)))
        return
    if GEMM_PROFILE is None:
        _check(load().cinema_gemm_bf16_grouped(arr, len(problems), _stream()), "gemm_grouped")
        return
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    _check(load().cinema_gemm_bf16_grouped(arr, len(problems), _stream()), "gemm_grouped")
    ev1.record()
    flops = sum(2.0 * g.m * g.n * g.k for g in arr)
    alg = sum(2.0 * (g.m * g.k + g.k * g.n) + 8.0 * g.m * g.n for g in arr)
    GEMM_PROFILE.append((64, flops, ev0, ev1, (sum(g.m for g in arr), arr[0].n, arr[0].k, 0, 0, len(problems), alg), tuple((g.m, g.n, g.k, 0, 0) for g in arr)))


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
    if GEMM_PROFILE is None or LANE is not None:
        _check(load().cinema_gemm_fp8_wgrad_p256(arr, len(problems), ws.data_ptr(), ws.numel() * 4, _stream()), "gemm_fp8_wgrad_p256")
        return
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    _check(load().cinema_gemm_fp8_wgrad_p256(arr, len(problems), ws.data_ptr(), ws.numel() * 4, _stream()), "gemm_fp8_wgrad_p256")
    ev1.record()
    flops = sum(2.0 * g.m * g.n * g.k for g in arr)
    alg = sum(1.0 * (g.m * g.k + g.k * g.n) + 8.0 * g.m * g.n for g in arr)  # every 8-bit operand once, the fp32 gradient read + written once
    GEMM_PROFILE.append((arr[0].kernel_used, flops, ev0, ev1, (sum(g.m for g in arr), arr[0].n, arr[0].k, 0, 0, len(problems), alg), tuple((g.m, g.n, g.k, 0, 0) for g in arr)))


def _warn_generic(m: int, n: int, k: int, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor) -> None:
    """A large GEMM on the generic FMA kernel is ~20x slower than the MFMA kernel and almost always an alignment accident (16-byte
    pointers, leading dimensions / n / k multiples of 8): say so once per shape instead of being silently slow."""
    key = (m, n, k)
    if key not in _WARNED_GENERIC:
        _WARNED_GENERIC.add(key)
        import warnings

        warnings.warn(f"cinema_gemm_bf16 {m}x{n}x{k} ran on the generic (non-MFMA) kernel: operand pointers a/b/out % 16 = "
                      f"{a.data_ptr() % 16}/{b.data_ptr() % 16}/{out.data_ptr() % 16}, strides {a.stride()}/{b.stride()}/{out.stride()}", stacklevel=3)


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
        quantize_fp8_site(out[0], q8)
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
            return r[0], r[1], quantize_fp8_site(r[1], q8), False
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
                                       int(force_generic or FORCE_GENERIC), _stream()), "attention_fwd")
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
            delta.data_ptr(), ptrs[5][0], ptrs[5][1], ptrs[6][0], ptrs[6][1], ptrs[7][0], ptrs[7][1], b, heads, tq, tk, hd, scale, int(force_generic or FORCE_GENERIC))
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
    key = (device.index, _stream(), LANE)
    t = _ATTN_COUNTERS.get(key)
    if t is None:
        t = _ATTN_COUNTERS[key] = persistent(lambda: torch.zeros(ATTN_COUNTERS, dtype=torch.int32, device=device))
    return t


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


SPARSE_WGRAD_PIPE = True  # False: the per-token index chase (the form the kernel tests compare against)
# the depthwise conv of the visible-voxel stem as a walk over neighbour TOKENS (csrc/stem_dw.hip; 64 / 128 channels, 4x4 / 2x2 token blocks); 0: the per-voxel neighbour
# lists of csrc/sparse_conv.hip, which stay the form for every other geometry
STEM_DW_PAIR = True  # False: the neighbour-list kernels for every geometry (the form the kernel tests compare against)


def sparse_pair_form(geom: SparseGeom, c: int, kdims: tuple) -> bool:
    """Whether :func:`sparse_dwconv` / :func:`sparse_dwconv_bwd_weight` take the token-pair kernels for this geometry (then no neighbour list is ever built)."""
    return bool(STEM_DW_PAIR and load().cinema_stem_dw_supported(C.byref(geom), c, *kdims))


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
    if STEM_DW_PAIR and load().cinema_stem_dw_supported(C.byref(geom), c, kx, ky, kz):  # token-pair form (csrc/stem_dw.hip): no neighbour lists
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
    if STEM_DW_PAIR and load().cinema_stem_dw_supported(C.byref(geom), c, kx, ky, kz):
        need = load().cinema_stem_dw_wgrad_workspace_bytes(geom.n_tok, c, kx, ky, kz)
        ws = _workspace("stem_dw_wgrad", (need + 3) // 4, x.device)
        _check(load().cinema_stem_dw_bwd_weight(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _p(dbias), ws.data_ptr(), need, C.byref(geom), c, kx, ky, kz, _stream()),
               "stem_dw_bwd_weight")
        return
    need = load().cinema_sparse_dwconv_wgrad_workspace_bytes(geom.n_tok, c, kx, ky, kz)
    ws = _workspace("sparse_wgrad", (need + 3) // 4, x.device)
    hidx = None
    if SPARSE_WGRAD_PIPE:
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


ROW_COPY_MULTI = True


def row_copy_multi(copies: list) -> None:
    """Several independent :func:`row_copy` calls in one launch: ``copies`` = list of dicts with row_copy's arguments (dst, src, dst_idx, ...)."""
    if not copies:
        return
    if len(copies) == 1 or not ROW_COPY_MULTI:
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


def sqnorm(g: torch.Tensor, out: torch.Tensor) -> None:
    _dev(g, out)
    ws = _workspace("sqnorm", 2048, g.device)
    _check(load().cinema_sqnorm_f32(g.data_ptr(), g.numel(), out.data_ptr(), ws.data_ptr(), _stream()), "sqnorm")


def clip_coef(sq: torch.Tensor, max_norm: float, coef_out: torch.Tensor | None, norm_out: torch.Tensor | None, step_state: torch.Tensor | None = None) -> None:
    """coef = min(1, max_norm / (sqrt(sq) + 1e-6)); a non-finite norm gives coef = 0 (= skip, see :func:`adamw`).  ``step_state`` int32 [2]:
    [0] counts applied updates, [1] skipped ones."""
    _dev(sq, coef_out, norm_out, step_state)
    if step_state is not None and (step_state.dtype != torch.int32 or step_state.numel() < 2):
        raise HipLibraryError("clip_coef: step_state must be int32 [2]")
    _check(load().cinema_clip_coef(sq.data_ptr(), max_norm, _p(coef_out), _p(norm_out), _p(step_state), _stream()), "clip_coef")


def adamw(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float,
          step: int, clip: torch.Tensor | None = None, shadow: torch.Tensor | None = None, step_state: torch.Tensor | None = None) -> None:
    """``step_state`` (int32 [2] written by :func:`clip_coef`): the update is skipped on the device when ``clip[0]`` is not > 0 and the Adam
    step of the bias corrections is ``step_state[0]`` (``step`` is then ignored)."""
    _dev(p, g, m, v, clip, shadow, step_state)
    step = max(int(step), 1)
    _check(load().cinema_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps, weight_decay,
                               1.0 - beta1**step, 1.0 - beta2**step, _p(clip), _p(shadow), _p(step_state), _stream()), "adamw")


class AdamWGroup(C.Structure):
    """Mirror of ``cinema_adamw_group``."""

    _fields_ = [("begin", C.c_longlong), ("end", C.c_longlong), ("lr", C.c_float), ("weight_decay", C.c_float)]


ADAMW_MAX_GROUPS = 64


def adamw_groups(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, groups: list, beta1: float, beta2: float, eps: float,
                 clip: torch.Tensor, shadow: torch.Tensor | None, step_state: torch.Tensor, max_blocks: int = 0) -> None:
    """AdamW over several ranges of ONE flat buffer in one launch: ``groups`` = [(begin, end, lr, weight_decay), ...], ascending element ranges (multiples of 4).
    ``max_blocks`` caps the number of workgroups (0: the library's default) for an update that shares the chip with another stream's work.
    Same arithmetic per element as :func:`adamw` with ``step_state`` (the layer-decay groups of a fine-tuning step: one launch instead of one per group)."""
    _dev(p, g, m, v, clip, shadow, step_state)
    arr = (AdamWGroup * len(groups))()
    for a, (b, e, lr, wd) in zip(arr, groups):
        a.begin, a.end, a.lr, a.weight_decay = int(b), int(e), float(lr), float(wd)
    if max_blocks:
        _check(load().cinema_adamw_groups_grid(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), arr, len(groups), beta1, beta2, eps, clip.data_ptr(),
                                               _p(shadow), step_state.data_ptr(), int(max_blocks), _stream()), "adamw_groups_grid")
        return
    _check(load().cinema_adamw_groups(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), arr, len(groups), beta1, beta2, eps, clip.data_ptr(), _p(shadow),
                                      step_state.data_ptr(), _stream()), "adamw_groups")


def kernel_launch_count() -> int:
    """Kernels the library has launched since it was loaded (merged lane-group launches count once; stream forks are not kernels)."""
    return int(load().cinema_kernel_launch_count())


def info() -> dict:
    out = (C.c_int * 8)()
    _check(load().cinema_hip_info(out), "info")
    return {"abi_version": out[0], "n_cus": out[1], "lds_bytes_per_block": out[2], "wave_size": out[3]}
