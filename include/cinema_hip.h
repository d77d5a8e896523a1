/* C-ABI of libcinema_hip.so -- the MI355X (gfx950) compute path behind cinema_amd's nn.Module layer.
 *
 * The upstream reference (mathpluscode/CineMA) has no FFI/plugin interface: every "kernel" is a PyTorch
 * ATen call inside cinema/vit.py, cinema/conv.py, cinema/convvit.py and cinema/mae/mae.py.  Each entry
 * point below therefore cites the ATen call site(s) of the reference it replaces (ids A1-A20).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; nothing is allocated or freed here;
 *   - `stream` is a hipStream_t (NULL = default stream); all work is stream-ordered, no host sync;
 *   - bf16 = raw bfloat16 bits (uint16_t); matrices are row-major with explicit leading dimensions (elements);
 *   - return value: 0 on success, CINEMA_ERR_* (<0) for rejected arguments, hipError_t (>0) if the launch failed.
 */
#ifndef CINEMA_HIP_H
#define CINEMA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CINEMA_ERR_BAD_ARG (-1)
#define CINEMA_ERR_UNSUPPORTED (-2)

/* library / device info: fills out[0..7] = {abi_version, n_CUs, lds_bytes_per_block, wave_size, 0...} */
int cinema_hip_info(int* out_host);

/* 8-bit copy of a kernel's output with per-tensor DELAYED scaling (the "fp8 MFMA path" of BASELINE config 5: the copies are the operands of the e4m3 forward,
 * data-gradient and weight-gradient GEMMs; the reference computes those matmuls in bf16 autocast, cinema/vit.py:472-477,565-575).  The producing kernel
 * quantises with the scale derived from the PREVIOUS step's maximum and records this step's maximum; cinema_fp8_sites_update turns maxima into scales. */
#define CINEMA_Q8_SLOTS 4096
typedef struct cinema_q8_out {
  uint8_t* data;              /* [rows][c] OCP e4m3 bytes, dense: q = e4m3(sat(value * *inv_scale)); NULL = record the maximum only */
  const float* inv_scale;     /* device scalar */
  float* colsum;              /* cinema_layernorm_bwd_deferred_q8 only, else NULL: fp32 [c], accumulated by the LATER cinema_ln_param_reduce_batched (item.dcol): the column sums of the
                                 values = the bias gradient of the projection whose output this gradient belongs to; the partials then carry THREE rows per workgroup */
  unsigned int* amax_slots;   /* device [CINEMA_Q8_SLOTS]: float bits of max|value| over this launch, every wave raises slot (wave id mod CINEMA_Q8_SLOTS): thousands of
                                 device-scope atomics on a handful of addresses serialise (measured: 64 slots cost 20-40 us per launch) */
} cinema_q8_out;
/* sites: amax_slots [n_sites][CINEMA_Q8_SLOTS], scale / inv_scale [n_sites].  For every site with a finite positive maximum: scale = margin * amax / 448 (the dequantisation
 * multiplier the consuming GEMM reads), inv_scale = 1 / scale; the slots are reset to 0.  Sites that recorded nothing keep their scales. */
int cinema_fp8_sites_update(unsigned int* amax_slots, float* scale, float* inv_scale, int n_sites, float margin, void* stream);
/* y8 = e4m3(sat(x * *inv_scale)) for a dense bf16 tensor of n elements (n % 8 == 0, 16-byte aligned) + the launch's max|x| into amax_slots: the stand-alone
 * producer for tensors whose kernels do not write an 8-bit copy themselves (attention outputs / gradients). */
int cinema_quantize_fp8_site(const uint16_t* x, long long n, const cinema_q8_out* q8, void* stream);
/* y (bf16) = e4m3 value * *scale for n elements (n % 8 == 0): the bf16 tensor of an 8-bit-only output for a consumer outside the e4m3 GEMMs */
int cinema_dequantize_fp8(const uint8_t* x8, long long n, const float* scale, uint16_t* y, void* stream);
/* the same over a bf16 matrix x [rows][ldx] (c columns, c % 8 == 0; the copy is dense [rows][c]) with its column sums on the side: colsum[c] += sum_r x[r][:] (fp32
 * atomics) - the 8-bit dY copy of a weight-gradient GEMM and that layer's bias gradient (the autograd of nn.Linear's bias, cinema/vit.py:472-477) in ONE pass */
int cinema_quantize_fp8_site_colsum(const uint16_t* x, int rows, int c, int ldx, const cinema_q8_out* q8, float* colsum, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * GEMM  (reference: nn.Linear / F.linear at cinema/vit.py:472-477,498-499,520, timm Mlp fc1/fc2 vit.py:570-575,
 * 1x1 ConvNd of MaskedConvBlock/ConvMlp cinema/conv.py:383-389, k==s patch convs cinema/convvit.py:94-102,252,
 * PatchEmbed.proj vit.py:294-298, dec_linear / pred heads cinema/mae/mae.py:395,435-440 and their autograd
 * backward.)   D[M,N] (+)= epilogue( alpha * sum_k A[m,k] * B[k,n] )
 *   a_kmajor=1: A stored [M][lda], k contiguous.     a_kmajor=0: A stored [K][lda], m contiguous.
 *   b_kmajor=1: B stored [N][ldb], k contiguous (the nn.Linear weight layout).  b_kmajor=0: B stored [K][ldb].
 * Epilogue, in order:  v = alpha*acc + bias[n];  aux_out[m,n] = bf16(v);  v = gelu(v) if act==1;
 *   v *= gelu'(gelu_in[m,n]);  v *= row_mask[m];  v += residual[m,n];  store (bf16 or fp32) or atomicAdd (fp32).
 * split_k > 1 requires accumulate=1 (fp32 atomics).  force_generic=1 selects the plain FMA kernel (any shape).
 */
typedef struct {
  const void* a; const void* b; void* d;
  int m, n, k;
  int lda, ldb, ldd;
  int a_kmajor, b_kmajor;
  float alpha;
  const float* bias;          /* [n] fp32 or NULL */
  const float* residual_f32;  /* [m][ld_res] or NULL */
  const uint16_t* residual_bf16;
  int ld_res;
  const uint16_t* gelu_in;    /* [m][ld_gelu] bf16 pre-activations or NULL */
  int ld_gelu;
  const uint8_t* row_mask;    /* [m] 0/1 or NULL */
  uint16_t* aux_out;          /* [m][ld_aux] bf16 pre-activation copy or NULL */
  int ld_aux;
  int act;                    /* 0 none, 1 exact GELU */
  int gelu_deriv;             /* 1: the auxiliary GELU tensor holds GELU'(pre-activation) instead of the pre-activation: with act = 1 aux_out receives the
                                 derivative (computed from the same erf terms as the activation), and gelu_in is multiplied in as it is - the data gradient
                                 through the activation then costs one multiply per element instead of an erf evaluation */
  int out_f32;                /* 1: D is fp32, 0: D is bf16 */
  int accumulate;             /* 1: D += result with fp32 atomics (out_f32 must be 1) */
  int split_k;                /* >=1 */
  int force_generic;
  void* workspace;            /* optional fp32 scratch. split_k > 1: >= split_k*m*n*4 bytes -> deterministic two-pass reduction instead of atomics.
                                 split_k == 1: >= 32 MiB lets the library cut the tiles left over after the last full round of workgroup slots
                                 into k-slices (partial tiles here; a fix-up launch applies the epilogue, or - with tail_counters - the last k-slice
                                 to arrive at a tile sums its partners' partial tiles and runs the epilogue inside the launch); contents are scratch,
                                 use is stream-ordered */
  long long workspace_bytes;
  float* a_rowsum;            /* optional (a_kmajor=0 only): a_rowsum[m] += sum_k A[m][k], i.e. the bias gradient of a weight-gradient GEMM, fused */
  const float* scale_a;       /* cinema_gemm_fp8 only: per-tensor dequantisation scales of the e4m3 operands (device scalars); NULL for cinema_gemm_bf16 */
  const float* scale_b;
  int scale_a_rows;           /* cinema_gemm_fp8: 1 = scale_a holds one scale per row of A (per-token activation scaling), 0 = one scalar */
  const void* conv_taps;      /* cinema_conv_gemm_bf16 only: device int4 [k / 8] tap table; conv_x/y/z/c = volume size and channels */
  int conv_x, conv_y, conv_z, conv_c;
  const void* conv_coords;    /* cinema_conv_wgrad_bf16 only: device int [rows], x | y << 10 | z << 20 of every voxel row */
  int conv_zb;                /* implicit convolution: 0 / 1 = one row per voxel; ZB > 1 = one row per group of ZB consecutive z voxels (see cinema_conv_gemm_bf16) */
  uint8_t* out8;              /* optional (bf16-output classes of cinema_gemm_bf16 / cinema_gemm_fp8): 8-bit copy of D with per-tensor DELAYED scaling (cinema_q8_out
                                 semantics): out8[m][ld_out8] = e4m3(sat(D[m][n] * *out8_inv_scale)); max|D| of this launch is atomic-maxed into out8_amax[CINEMA_Q8_SLOTS] */
  int ld_out8;                /* with out8 given, `d` may be NULL (bf16 classes of the MFMA kernels): only the 8-bit copy is written */
  const float* out8_inv_scale;
  unsigned int* out8_amax;    /* may be given without out8: records the maximum only (calibration step) */
  float* colsum_partials;     /* optional (bf16-output classes, MFMA kernels): fp32 [ceil(m / 32)][n], 16-byte aligned: column sums of D over every strip of 32 rows, each element
                                 written once (no zeroing, no atomics); cinema_colsum over the strips = the bias gradient of the layer whose dY this GEMM produces */
  void* tail_counters;        /* optional, with `workspace` at split_k == 1: >= 8 KiB of device memory (16-byte aligned), ZERO before the first use (its last word is set if a bounded wait ever gives up) and left zero by every launch,
                                 one per stream: the split tail is finished inside the GEMM launch (no fix-up launch); NULL = fix-up launch */
  int kernel_used;            /* OUT: 0 generic FMA kernel; otherwise the 128x128 MFMA kernel: operand layout (1 fwd, 2 dgrad, 3 wgrad)
                                 + 8 x epilogue class (0 general, 1 bf16, 2 bf16+GELU, 3 bf16 x GELU', 4 fp32); 64: cinema_gemm_bf16_grouped;
                                 + 128: the BK = 32 instance; 2048 + layout + 8 x class: cinema_gemm_bf16_p256 */
} cinema_gemm_args;
int cinema_gemm_bf16(cinema_gemm_args* args_host, void* stream);
/* fp8 forward GEMM (BASELINE config 5: "fp8 MFMA path"; the reference picks its autocast dtype at cinema/device.py:58-66): the same call with OCP e4m3
 * operands a [M][lda] / b [N][ldb] (bytes, k-major: a_kmajor = b_kmajor = 1), per-tensor scales scale_a / scale_b, K % 16 == 0:
 * D = epilogue(alpha * scale_a * scale_b * A8 B8^T) on v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales).  Epilogue: bias, exact GELU (+ bf16
 * pre-activation copy), fp32 residual, or x GELU'(gelu_in) (the data gradient through fc1's activation); bf16 or fp32 D.  Data gradients run through the
 * same call with A = per-row quantised dY and B = the transposed weight shadow (cinema_quantize_fp8_segments_t); weight gradients stay bf16. */
int cinema_gemm_fp8(cinema_gemm_args* args_host, void* stream);
/* Implicit-GEMM "same" convolution (dense 3^n convs of ConvResBlock, cinema/conv.py:320-345, forward and data gradient): args->a = channels-last bf16
 * volume x [batch*X*Y*Z][C] (C % 8 == 0), m = batch*X*Y*Z, b = weights [n][ldb] with features (tap, channel) zero-padded to k = ldb (k % 8 == 0),
 * conv_taps[j] (one per 16-byte k-chunk = 8 channels of one tap) = {row delta of the neighbour, (dx+1) | (dy+1)<<2 | (dz+1)<<4, first channel, valid}.
 * D[r][n] = sum over the taps / channels of x at the neighbour voxel (zeros outside the volume) + bias (+ fp32 residual); bf16 or fp32 D.  The im2col
 * matrix (27 x the activation) is never written: the MFMA kernel's A tiles are gathered from the volume by the LDS-DMA. */
int cinema_conv_gemm_bf16(cinema_gemm_args* args_host, void* stream);
/* z-blocked form of the same convolution for narrow layers (conv_zb = ZB in {2, 4}, kernel 3x3x3, Z % ZB == 0): m = batch*X*Y*Z/ZB rows, n = ZB*c_out - the
 * output rows are byte-for-byte the plain [voxel][c_out] rows -, k = 9*(ZB+2)*C with taps over (3, 3, ZB+2) offsets (dz field: 3 bits, offset from the
 * group's first voxel + 1).  cinema_conv_weight_zblock builds the block-banded weights [ZB*n][9*(ZB+2)*c] from the plain [n][27*c] rows (features
 * (tap, channel)): w_zb[zo*n + co][((tx*3+ty)*(ZB+2) + dzi)*c + ci] = w[co][((tx*3+ty)*3 + tz)*c + ci] with tz = dzi - zo (transpose = 1, the data-gradient
 * operand whose taps point the other way: tz = zo + 2 - dzi), zero when tz is outside 0..2; bias_zb = bias repeated ZB times (NULL bias: skipped).
 * The weight gradient of the z-blocked problem is R [ZB*n][9*(ZB+2)*c] (cinema_conv_wgrad_bf16 with conv_zb and group coordinates);
 * cinema_conv_wgrad_zfold adds its ZB bands into dst [n][ld_dst] (features (tap, channel)) and the ZB row-sum segments into db [n]. */
int cinema_conv_weight_zblock(const uint16_t* w, int n, int c, int ld_w, int zb, int transpose, uint16_t* w_zb, const float* bias, float* bias_zb, void* stream);
int cinema_conv_wgrad_zfold(const float* r, int n, int c, int zb, float* dst, int ld_dst, const float* rowsum_zb, float* db, void* stream);
/* Weight gradient of the same convolution, again without an im2col matrix: a = dy [rows][lda] bf16 (k = rows, m = c_out), b = the volume x [rows][C],
 * n = weight row length (taps * C padded to 8), conv_taps = the FORWARD table, conv_coords[r] = voxel coordinates of row r: D[co][(tap, ci)] fp32 (+)=
 * sum_r dy[r][co] * x[nbr_tap(r)][ci]; a_rowsum[co] += sum_r dy[r][co] (bias gradient).  Deterministic split-K through `workspace` (>= split_k*m*n*4 B). */
int cinema_conv_wgrad_bf16(cinema_gemm_args* args_host, void* stream);
/* Weights of the DATA GRADIENT of that convolution: w fp32 (c_out, c_in, kvol) contiguous -> rows [c_in][ld] bf16 with rows[ci][tap * c_out + co] = w[co][ci][tap],
 * zero-padded to ld (a multiple of 8 >= kvol * c_out). */
int cinema_conv_weight_dgrad(const float* w, uint16_t* rows, int c_out, int c_in, int kvol, int ld, void* stream);
/* Per-tensor e4m3 quantisation of a bf16 buffer (current scaling): amax = max|x|, scale = amax / 448 (1 when amax == 0), y = e4m3(x / scale)
 * (v_cvt_pk_fp8_f32, OCP format on gfx950); *scale_out = scale.  amax_ws: 4 bytes of scratch.  n % 8 == 0, 16-byte aligned x, 8-byte aligned y. */
int cinema_quantize_fp8(const uint16_t* x, long long n, uint8_t* y, float* scale_out, unsigned int* amax_ws, void* stream);
/* The same for n_seg segments [seg_bounds[2i], seg_bounds[2i+1]) (element offsets, multiples of 8, device int64) of ONE flat bf16 buffer - the weight
 * shadows of a whole model in three launches; y has the layout of x, scales [n_seg], amax_ws n_seg words of scratch. */
/* Per-row form: x dense bf16 [rows][c] (c % 8 == 0) -> y e4m3 [rows][c], row_scale[rows] = amax(row) / 448; one launch (the row maximum is local). */
int cinema_quantize_fp8_rows(const uint16_t* x, int rows, int c, uint8_t* y, float* row_scale, void* stream);
int cinema_quantize_fp8_segments(const uint16_t* x, const long long* seg_bounds, int n_seg, uint8_t* y, float* scales, unsigned int* amax_ws, void* stream);
/* Transposed e4m3 copies of the 2-D segments (weights [rows][cols], rows and cols multiples of 8, offsets multiples of 8): seg_desc[3 s] = element offset,
 * [3 s + 1] = rows, [3 s + 2] = cols (device int64); yt holds [cols][rows] bytes at the same offset, scaled with scales[s] of the call above.  The operand
 * of the fp8 DATA-GRADIENT GEMM dX = dY W (reduction over the weight's output features, which must be contiguous for the e4m3 MFMA). */
int cinema_quantize_fp8_segments_t(const uint16_t* x, const long long* seg_desc, int n_seg, const float* scales, uint8_t* yt, void* stream);
/* The tiles of up to 8 independent weight-gradient GEMMs (a_kmajor = b_kmajor = 0, fp32 D, optional accumulate and a_rowsum, no other
 * epilogue term, split_k ignored) in ONE launch with whole-K tiles: the four dW of a transformer block (cinema/vit.py:525-609) have
 * 36-144 output tiles each and would otherwise be cut into k-slices with fp32 slabs and a reduce launch each. */
int cinema_gemm_bf16_grouped(cinema_gemm_args* args_host_array, int count, void* stream);
/* Persistent 256x256x64 form of the same GEMMs (csrc/gemm256.hip): one 8-wave workgroup per CU walks a list of (problem, tile, k-range) pieces and a
 * tile cut into several pieces is finished INSIDE the launch by its last-arriving piece (fp32 partial slots + two counters per tile; no reduce launch,
 * no fix-up launch).  Up to 12 problems of one operand layout (a_kmajor / b_kmajor as cinema_gemm_bf16; (0, 1) unsupported) and one epilogue class
 * (bf16 | bf16 + GELU (+ aux_out) | bf16 x GELU'(gelu_in) | fp32 (+ bias, residual_f32 or accumulate)); a_rowsum for reduction-strided A.
 *   schedule 0: every tile of problem i in balanced k-slices (the linear-layer weight gradients of a transformer block, cinema/vit.py:565-575, in one
 *               launch; split_k == 1 in args[0] keeps whole-K tiles), schedule 1 (count == 1): equal contiguous (tile, k-tile) ranges per workgroup.
 * workspace: >= cinema_gemm_p256_workspace_bytes(), 256-byte aligned, first 64 KiB ZERO before the first use (left zero by every launch), one per stream. */
long long cinema_gemm_p256_workspace_bytes(void);
int cinema_gemm_bf16_p256(cinema_gemm_args* args_host_array, int count, int schedule, void* workspace, long long workspace_bytes, void* stream);
/* The same persistent launch for WEIGHT GRADIENTS ON 8-BIT OPERANDS (BASELINE config 5, "fp8 MFMA path"; replaces the bf16 autocast matmuls behind the backward of
 * the transformer MLP's nn.Linear layers, cinema/vit.py:565-575 via timm Mlp): a = dY8 [rows][lda] and b = X8 [rows][ldb] OCP-e4m3 bytes, row-major
 * [token][feature] as the producing kernels write them (a_kmajor = b_kmajor = 0), k = rows, m / n = feature counts (multiples of 16; lda, ldb multiples of 16
 * bytes; 16-byte aligned pointers), scale_a / scale_b = per-tensor dequantisation scales (device scalars), D fp32 [m][ldd], accumulate as cinema_gemm_bf16_p256:
 *   D (+)= scale_a * scale_b * dY8^T X8   on v_mfma_scale_f32_32x32x64_f8f6f4 with ds_read_b64_tr_b8 fragment reads (no transposed copies of the operands).
 * No a_rowsum / bias / activation terms (bias gradients: cinema_colsum).  Up to 12 problems, split schedule, workspace as cinema_gemm_bf16_p256.
 * kernel_used OUT: 4096 + 3 + 8 x 4. */
int cinema_gemm_fp8_wgrad_p256(cinema_gemm_args* args_host_array, int count, void* workspace, long long workspace_bytes, void* stream);

/* column sums: out[n] += sum_{i<m} x[row(i), n] with row(i) = row_idx ? row_idx[i] : i  (bias / token-parameter gradients).
 * x bf16 (x_dtype 0) or fp32 (1), row-major [.][ldx]; out fp32 [n], accumulated atomically */
int cinema_colsum(const void* x, int x_dtype, const int* row_idx, int m, int n, int ldx, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * LayerNorm (reference: nn.LayerNorm norm1/norm2/encoder.norm/fusion.norm/decoder.norm cinema/vit.py:549,564,650,738,
 * cinema/convvit.py:254; ConvLayerNorm + nn.GELU cinema/conv.py:169-187,271-272).
 * fwd: y = LN(x)*gamma+beta over the last dim (c), optional exact GELU; x fp32 or bf16, y bf16 and/or fp32;
 *      mean/rstd (fp32 [rows]) saved for the backward.
 * bwd: dx = LN'(dy) (+ dx_residual);  writes dx_f32 and/or dx_bf16; dgamma/dbeta accumulated (fp32): with a workspace of
 *      >= 2048*2*c*4 bytes the blocks store partial sums and a second small kernel adds them up, without one every block adds atomically.
 *      If act==1 the incoming dy is w.r.t. gelu(LN(x)) and is chained through gelu' (recomputed from x, stats).
 */
int cinema_layernorm_fwd(const void* x, int x_is_bf16, int ldx, const float* gamma, const float* beta, int rows, int c, float eps,
                         int act, uint16_t* y_bf16, float* y_f32, int ldy, float* mean, float* rstd, void* stream);
/* cinema_layernorm_fwd that also writes an e4m3 copy of y ([rows][c] bytes, dense) with one dequantisation scale per row - the quantisation of the
 * GEMM input costs no pass of its own (the row maximum is a reduction over the lanes that already hold the row). */
int cinema_layernorm_fwd_fp8(const void* x, int x_is_bf16, int ldx, const float* gamma, const float* beta, int rows, int c, float eps, int act, uint16_t* y_bf16,
                             float* y_f32, int ldy, float* mean, float* rstd, uint8_t* y_fp8, float* row_scale, void* stream);
/* the same with a per-TENSOR delayed scale (cinema_q8_out) instead of per-row scales: the copy then also serves the weight-gradient GEMM, whose reduction
 * runs over the rows */
int cinema_layernorm_fwd_q8(const void* x, int x_is_bf16, int ldx, const float* gamma, const float* beta, int rows, int c, float eps, int act, uint16_t* y_bf16,
                            float* y_f32, int ldy, float* mean, float* rstd, const cinema_q8_out* q8, void* stream);
int cinema_layernorm_bwd(const void* dy, int dy_is_bf16, int lddy, const void* x, int x_is_bf16, int ldx, const float* gamma,
                         const float* beta, const float* mean, const float* rstd, int rows, int c, int act,
                         const float* dx_residual, float* dx_f32, uint16_t* dx_bf16, int lddx, float* dgamma, float* dbeta,
                         float* workspace, long long workspace_bytes, void* stream);

/* The same with the parameter-gradient reduction left to the caller: the per-block partial sums stay in `workspace` ([n_partials][2*c] fp32,
 * n_partials returned; 0 = the gradients were added directly) and cinema_ln_param_reduce_batched adds the partials of many LayerNorms to their
 * dgamma / dbeta in one launch (a training step has 86 LayerNorm backward passes: 86 tiny reduce launches otherwise). */
typedef struct {
  const float* partials;  /* the workspace of one cinema_layernorm_bwd_deferred call */
  int n_partials, c;
  float* dgamma;          /* accumulated; either may be NULL */
  float* dbeta;
  float* dcol;            /* non-NULL: the partials carry a third row per workgroup (column sums of dx, cinema_q8_out.colsum), accumulated here */
} cinema_ln_reduce_item;
/* bytes of per-block partial sums the deferred backward of a (rows, c) LayerNorm writes (host-only query; 0: no workspace needed) */
long long cinema_layernorm_bwd_workspace_bytes(int rows, int c);
int cinema_layernorm_bwd_deferred(const void* dy, int dy_is_bf16, int lddy, const void* x, int x_is_bf16, int ldx, const float* gamma,
                                  const float* beta, const float* mean, const float* rstd, int rows, int c, int act,
                                  const float* dx_residual, float* dx_f32, uint16_t* dx_bf16, int lddx, float* dgamma, float* dbeta,
                                  float* workspace, long long workspace_bytes, int* n_partials_out, void* stream);
/* cinema_layernorm_bwd_deferred that also writes an 8-bit copy of dx (cinema_q8_out; dense [rows][c]): the gradient of the residual stream is the dY operand of
 * the preceding projection's e4m3 data- and weight-gradient GEMMs */
int cinema_layernorm_bwd_deferred_q8(const void* dy, int dy_is_bf16, int lddy, const void* x, int x_is_bf16, int ldx, const float* gamma, const float* beta,
                                     const float* mean, const float* rstd, int rows, int c, int act, const float* dx_residual, float* dx_f32, uint16_t* dx_bf16,
                                     int lddx, float* dgamma, float* dbeta, float* workspace, long long workspace_bytes, int* n_partials_out,
                                     const cinema_q8_out* q8, void* stream);
int cinema_ln_param_reduce_batched(const cinema_ln_reduce_item* items_host, int count, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Attention (reference: F.scaled_dot_product_attention / matmul-softmax-matmul cinema/vit.py:505-517, no mask, no
 * dropout).  q:[b,tq,h,hd] k,v:[b,tk,h,hd] addressed as base + (b*t + t_i)*ld + h*hd + d (bf16) so that the fused
 * qkv / kv GEMM outputs are consumed in place.  o:[b,tq,h*hd] bf16.  lse:[b,h,tq] fp32 = log2-domain
 * log-sum-exp of scale*log2(e)*q.k.  hd in {32,64} -> MFMA flash kernel; any hd<=128 -> generic kernel.
 * o_lo (optional, NULL = not wanted; same addressing and ld as o): bf16(O - float(bf16(O))), the part of the fp32 output the bf16 rounding removed.  Training keeps
 * it for the backward pass: delta = rowsum(dO (o + o_lo)) is then exact to 2^-17.  Formed from o alone its 2^-9 error shifts every dS of a query and is multiplied
 * by the keys' common component: 13 % of dQ / 5 % of attn.q.weight's gradient in the late blocks of ViT-Large (profiles/r06_i_attn_dq_error.txt).
 */
int cinema_attention_fwd(const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v, int ldv, uint16_t* o, uint16_t* o_lo, int ldo,
                         float* lse, int b, int h, int tq, int tk, int hd, float scale, int force_generic, void* stream);
/* delta:[b,h,tq] fp32 scratch (rowsum(dO*(o + o_lo))) is written by the call; o_lo may be NULL (delta from o alone). dq/dk/dv use the same addressing as q/k/v. */
int cinema_attention_bwd(const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v, int ldv, const uint16_t* o, const uint16_t* o_lo,
                         int ldo, const uint16_t* d_o, int lddo, const float* lse, float* delta, uint16_t* dq, int lddq,
                         uint16_t* dk, int lddk, uint16_t* dv, int lddv, int b, int h, int tq, int tk, int hd, float scale,
                         int force_generic, void* stream);
/* The same with scratch for the ONE-PASS backward at head_dim 64 (P and dS evaluated once per score; passes of 256 keys per workgroup, the keys of a
 * (batch, head) split over several workgroups when there are fewer pairs than compute units): workspace >= cinema_attention_bwd_workspace_bytes(...) bytes
 * (16-byte aligned, contents irrelevant), counters = b*h zero-initialised words that every launch leaves zero.  Without sufficient scratch the call falls back
 * to the dQ + dK/dV kernel pair of cinema_attention_bwd (same results up to summation order).  Reference: cinema/vit.py:505-517 (backward of SDPA). */
long long cinema_attention_bwd_workspace_bytes(int b, int h, int tq, int tk, int hd);
int cinema_attention_bwd_ws(const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v, int ldv, const uint16_t* o, const uint16_t* o_lo,
                            int ldo, const uint16_t* d_o, int lddo, const float* lse, float* delta, uint16_t* dq, int lddq,
                            uint16_t* dk, int lddk, uint16_t* dv, int lddv, int b, int h, int tq, int tk, int hd, float scale,
                            int force_generic, float* workspace, long long workspace_bytes, unsigned* counters, int n_counters, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Depthwise 5^n convolution, channels-last (reference: MaskedConvBlock.dw_conv, kernel 5 in every axis, "same"
 * zero padding, cinema/conv.py:385,410-413).  x,y: [b, X, Y, Z, c] bf16 (2-D views use Z=1,kz=1).
 * w: fp32 [c][kx*ky*kz] (the torch (c,1,kx,ky,kz) layout flattened), bias fp32 [c].
 */
int cinema_dwconv_fwd(const uint16_t* x, const float* w, const float* bias, uint16_t* y, int b, int X, int Y, int Z, int c, int kx, int ky,
                      int kz, void* stream);
/* dx = correlate(dy, flipped w); out_mask (uint8 [b*X*Y*Z] or NULL) zeroes masked voxels of dx (the `mask *` of conv.py:411) */
int cinema_dwconv_bwd_data(const uint16_t* dy, const float* w, uint16_t* dx, const uint8_t* out_mask, int b, int X, int Y, int Z, int c, int kx,
                           int ky, int kz, void* stream);
/* dw[c][taps] += sum x*dy ; dbias[c] += sum dy.  workspace (fp32, >= 1024 * c * (taps+1) * 4 bytes) makes the reduction a deterministic
 * two-pass (per-block slabs + reduce kernel); without it the kernel falls back to fp32 atomics. */
int cinema_dwconv_bwd_weight(const uint16_t* x, const uint16_t* dy, float* dw, float* dbias, float* workspace, long long workspace_bytes, int b,
                             int X, int Y, int Z, int c, int kx, int ky, int kz, void* stream);

/* Dense k^n "same"-padded convolution of the segmentation decoder (reference ConvResBlock conv1/conv2, cinema/conv.py:276-348, called from
 * cinema/segmentation/convunetr.py:320-345,455-485) = im2col + cinema_gemm_bf16:
 *   cols[v][(tap, c)] = x[v + tap - r][c], zero outside the volume; rows are ld_cols (multiple of 8, >= taps*c) wide, the tail is zero-filled.
 *   col2im is the data-gradient gather dx[v][c] = sum_tap dcols[v - (tap - r)][(tap, c)].  x / dx: bf16 channels-last [b][X][Y][Z][c]. */
int cinema_im2col(const uint16_t* x, uint16_t* cols, int ld_cols, int b, int X, int Y, int Z, int c, int kx, int ky, int kz, void* stream);
int cinema_col2im(const uint16_t* dcols, int ld_cols, uint16_t* dx, int b, int X, int Y, int Z, int c, int kx, int ky, int kz, void* stream);

/* ---- the same depthwise conv restricted to the VISIBLE voxels of an MAE step (exactly the reference result at those voxels: every other op of
 * cinema/conv.py:349-415 is per-voxel and the conv input is zero at masked voxels, so visible outputs depend on visible voxels only; the
 * reference computes all voxels and discards 75 % of them, cinema/mae/mae.py:548-550).
 * Activations are token-major compact rows: row = token_row * (bx*by*bz) + pos[voxel within token, raster (x,y,z)].
 *   keep[n_tok]       flat token id  sample * (tx*ty*tz) + token  of every kept token, in row order
 *   rank[b*tx*ty*tz]  row of that token, or -1 when it is masked
 *   pos[bx*by*bz]     raster voxel-in-token -> row offset inside the token's block (lets the caller keep coarser-stage children contiguous) */
typedef struct {
  int b, tx, ty, tz;  /* samples, token grid per sample */
  int bx, by, bz;     /* voxels per token along each axis at this stage */
  int n_tok;          /* kept tokens over all samples */
  const int* keep;
  const int* rank;
  const int* pos;
} cinema_sparse_geom;
/* neighbour lists of a stage for one kernel extent: nbr[rows][128] packed (tap << 24 | source row) of the visible stencil neighbours in tap
 * order, cnt[rows]; rows = n_tok*bx*by*bz; both together need cinema_sparse_nbr_ints(rows) ints.  Built once per mask, reused by every conv. */
long long cinema_sparse_nbr_ints(int n_rows);
int cinema_sparse_nbr_build(const cinema_sparse_geom* geom, int kx, int ky, int kz, int* nbr, int* cnt, void* stream);
/* y = bias + dwconv(x) on compact rows; flip = 1 evaluates the data gradient (taps reversed, pass bias = NULL) */
int cinema_sparse_dwconv_fwd(const uint16_t* x, const float* w, const float* bias, uint16_t* y, const cinema_sparse_geom* geom, const int* nbr,
                             const int* cnt, int c, int kx, int ky, int kz, int flip, void* stream);
/* dw[c][taps] += sum x*dy, dbias[c] += sum dy over the visible voxels; workspace >= cinema_sparse_dwconv_wgrad_workspace_bytes(...) */
int cinema_sparse_dwconv_bwd_weight(const uint16_t* x, const uint16_t* dy, float* dw, float* dbias, float* workspace, long long workspace_bytes,
                                    const cinema_sparse_geom* geom, int c, int kx, int ky, int kz, const int* halo_idx, void* stream);
/* halo_idx (optional, NULL = the per-token index chase): [n_tok][(bx+kx-1)(by+ky-1)(bz+kz-1)] compact row of every halo voxel of every kept token (-1 =
 * masked / outside), cinema_sparse_halo_ints(...) ints, built once per mask and kernel extent; with it the weight gradient runs as a pipeline over the
 * tokens of a workgroup (rows of the next token in flight while the current one is accumulated). */
long long cinema_sparse_halo_ints(const cinema_sparse_geom* geom, int kx, int ky, int kz);
int cinema_sparse_halo_index(const cinema_sparse_geom* geom, int kx, int ky, int kz, int* halo_idx, void* stream);
long long cinema_sparse_dwconv_wgrad_workspace_bytes(int n_tok, int c, int kx, int ky, int kz);

/* ---------------------------------------------------------------------------------------------------------
 * Fused per-voxel halves of a MaskedConvBlock on compact rows [rows][c] (reference: cinema/conv.py:405-413
 *   x = x + conv2(dw_conv(conv1(norm1(x)))) ; x = x + mlp(norm2(x)),  ConvLayerNorm cinema/conv.py:169-187 eps 1e-6,
 *   ConvMlp cinema/conv.py:111-166 = fc1 -> exact GELU -> fc2 on 1x1 convolutions).  c in {64, 128} (cinema_stem_supported), hidden = 4 c.
 * Weights are the bf16 operand shadows in nn.Linear layout [out][in]; biases / LayerNorm parameters fp32.  The depthwise conv between conv1 and conv2 stays
 * cinema_sparse_dwconv_fwd.  In the forward pass the 4c-wide hidden activation of the MLP never leaves the registers (the unfused form wrote it and its
 * GELU derivative to HBM); the backward pass recomputes it from the saved x1.
 */
/* The depthwise conv of cinema_sparse_dwconv_* as a walk over the 3 x 3 x 5 neighbour TOKENS of every kept token (no neighbour lists): supported for c in {64, 128},
 * token blocks 4x4x1 / 2x2x1 with a 5x5x5 kernel and 1x4x4 / 1x2x2 (2-D views, leading axis 1) with a 1x5x5 kernel; same arguments and results as the list form
 * (fp32 taps, other summation order).  workspace of the weight gradient >= cinema_stem_dw_wgrad_workspace_bytes(...). */
int cinema_stem_dw_supported(const cinema_sparse_geom* geom, int c, int kx, int ky, int kz);
int cinema_stem_dw_fwd(const uint16_t* x, const float* w, const float* bias, uint16_t* y, const cinema_sparse_geom* geom, int c, int kx, int ky, int kz, int flip,
                       void* stream);
long long cinema_stem_dw_wgrad_workspace_bytes(int n_tok, int c, int kx, int ky, int kz);
int cinema_stem_dw_bwd_weight(const uint16_t* x, const uint16_t* dy, float* dw, float* dbias, float* workspace, long long workspace_bytes,
                              const cinema_sparse_geom* geom, int c, int kx, int ky, int kz, void* stream);
int cinema_stem_supported(int c);
/* xn = LN(x; gamma, beta) (bf16, may be NULL) and h = xn w^T + bias (bf16) */
int cinema_stem_ln_linear(const float* x, const float* gamma, const float* beta, float eps, const uint16_t* w, const float* bias, uint16_t* xn, uint16_t* h,
                          int rows, int c, void* stream);
/* x1 = x + d w2^T + b2 (fp32, may be NULL: not kept) ; x2 = x1 + fc2(GELU(fc1(LN(x1; gamma, beta)))) */
int cinema_stem_mlp_fwd(const uint16_t* d, const float* x, const uint16_t* w2, const float* b2, const float* gamma, const float* beta, float eps, const uint16_t* wf1,
                        const float* bf1, const uint16_t* wf2, const float* bf2, float* x1, float* x2, int rows, int c, void* stream);
/* rows of LayerNorm parameter-gradient partials ([d gamma (c) | d beta (c)] each, the layout cinema_ln_param_reduce_batched sums) a backward call may write */
int cinema_stem_partials(int rows);
/* backward of cinema_stem_mlp_fwd from g2 = dL/dx2 and the saved x1: dx1 = dL/dx1 (fp32 and bf16), dd = dx1 w2 = dL/dd (bf16), and the operand pairs of the
 * weight gradients - a = GELU(fc1 output) and dz = dL/d(fc1 output) ([rows][4c] bf16), xn2 = LN(x1), g2 in bf16; *n_partials_out rows of LN partials */
int cinema_stem_mlp_bwd(const float* g2, const float* x1, const uint16_t* w2, const float* gamma, const float* beta, float eps, const uint16_t* wf1, const float* bf1,
                        const uint16_t* wf2, float* dx1, uint16_t* dx1_16, uint16_t* dd, uint16_t* a, uint16_t* dz, uint16_t* xn2, uint16_t* g2_16, float* partials,
                        int rows, int c, int* n_partials_out, void* stream);
/* backward of cinema_stem_ln_linear: dx = dres + LN'(x)(dh w)  (dres = the residual gradient dL/dx1, may be NULL); LN partials as above */
int cinema_stem_ln_linear_bwd(const uint16_t* dh, const float* x, const float* dres, const float* gamma, float eps, const uint16_t* w, float* dx, float* partials,
                              int rows, int c, int* n_partials_out, void* stream);
/* Weight gradients with small outputs over many rows (the 1x1 convolutions of the stem: reference autograd of cinema/conv.py:405-413):
 *   dw[n][k] += sum_r dy[r][n] x[r][k],  db[n] += sum_r dy[r][n]   for up to 6 problems with the same row count in one launch + one ordered reduce.
 * n, k powers of two in 32 .. 512, n + k <= 640.  workspace >= cinema_stem_wgrad_workspace_bytes(...) bytes, contents irrelevant. */
typedef struct {
  const uint16_t* dy; const uint16_t* x;  /* bf16 [rows][n], [rows][k], dense */
  float* dw; float* db;                   /* fp32 [n][k] dense, [n] or NULL */
  int rows, n, k;
} cinema_stem_wgrad_problem;
int cinema_stem_wgrad_slices(int rows);
long long cinema_stem_wgrad_workspace_bytes(const cinema_stem_wgrad_problem* probs, int count);
int cinema_stem_wgrad(const cinema_stem_wgrad_problem* probs, int count, float* workspace, long long workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Non-overlapping patch gather / scatter (reference: patchify cinema/vit.py:67-161 and the im2col of the k==s
 * ConvNd layers cinema/convvit.py:94-102,252).  Source is addressed by element strides (sb, sc, sx, sy, sz) so both
 * channels-first images and channels-last feature maps work.  Output row = token (b, gx, gy, gz) raster order,
 * feature order (px, py, pz, c).  token_idx (int32, [n_rows]) optionally selects flat token ids b*G + g (kept tokens).
 * gather: out[row, f] = src[...]  (out bf16 or fp32).  scatter (backward): dst[...] (+)= rows[row, f].
 */
typedef struct {
  int b, c, gx, gy, gz, px, py, pz;     /* grid and patch extents */
  long long sb, sc, sx, sy, sz;         /* element strides of the volume */
  int n_rows;                           /* rows produced/consumed (= b*gx*gy*gz when token_idx is NULL) */
  const int* token_idx;
} cinema_patch_geom;
int cinema_patch_gather(const void* src, int src_dtype, void* out, int out_dtype, int ld_out, const cinema_patch_geom* geom_host,
                        void* stream);
int cinema_patch_scatter(const void* rows, int rows_dtype, int ld_rows, void* dst, int dst_dtype, int accumulate,
                         const cinema_patch_geom* geom_host, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Row copy with optional gather/scatter indices and an optional added row (reference: bool-mask token selection,
 * cat/split, pos-embed adds  cinema/mae/mae.py:98-104,550,580-582, cinema/vit.py:672-674, cinema/convvit.py:205,288).
 *   dst[di(i), :] (+)= src[si(i), :] + add[ai(i), :]      for i in [0, n_rows)
 * with xi(i) = idx ? idx[i] : i.  dtypes: 0 = bf16, 1 = fp32.  src may be NULL (treated as zeros).
 */
int cinema_row_copy(void* dst, int dst_dtype, int ld_dst, const int* dst_idx, const void* src, int src_dtype, int ld_src,
                    const int* src_idx, const void* add, int add_dtype, int ld_add, const int* add_idx, int n_rows, int c,
                    int accumulate, void* stream);
/* The arguments of one cinema_row_copy as a struct, and up to any number of them (12 per launch) in ONE grid: the token assembly / split /
 * concatenation ops of a step (cinema/vit.py:672-674, cinema/mae/mae.py:98-104,580-585) are 4-9 independent small copies each. */
typedef struct {
  void* dst; int dst_dtype, ld_dst; const int* dst_idx;
  const void* src; int src_dtype, ld_src; const int* src_idx;
  const void* add; int add_dtype, ld_add; const int* add_idx;
  int n_rows, c, accumulate;
} cinema_row_copy_args;
int cinema_row_copy_multi(const cinema_row_copy_args* segs_host, int count, void* stream);

/* Segmentation loss of one view (reference _segmentation_loss, cinema/segmentation/train.py:77-103): cross_entropy(ignore_index = -1) +
 * monai DiceLoss(include_background=False, softmax=True) against one_hot(max(labels, 0)); smooth_nr = smooth_dr = 1e-5, mean over samples x foreground classes.
 * logits fp32 channels-last rows [b*vox][c] (2 <= c <= 16), labels int32 [b*vox].
 * fwd: acc = scratch of (b*c*3 + 2) floats; out4 = {loss, cross entropy, mean dice loss, 1/count}; coef [b][c][2] is consumed by bwd.
 * bwd: dlogits [b*vox][c] = upstream[0] * d loss / d logits (upstream NULL = 1). */
int cinema_seg_loss_fwd(const float* logits, const int* labels, int b, int vox, int c, float* acc, float* out4, float* coef, void* stream);
int cinema_seg_loss_bwd(const float* logits, const int* labels, int b, int vox, int c, const float* coef, const float* out4, const float* upstream,
                        float* dlogits, void* stream);
/* Losses of the ConvViT classification / regression heads (a handful of rows; forward value and gradient in one launch):
 *   head_ce:  F.cross_entropy(logits, label, label_smoothing) of the reference's classification_loss (cinema/classification/train.py:80-110): logits fp32
 *       [b][c], labels int32 [b] in [0, c); out1[0] = mean loss, dlogits [b][c] = d loss / d logits.
 *   head_mse: F.mse_loss(pred, label) with the values regression_loss reports (cinema/regression/train.py:21-56): pred / label fp32 [n] (n = batch x targets);
 *       out6 = {mse, mae (F.l1_loss), max label, min label, max pred, min pred}, dpred [n] = d mse / d pred. */
int cinema_head_ce(const float* logits, const int* labels, int b, int c, float label_smoothing, float* out1, float* dlogits, void* stream);
int cinema_head_mse(const float* pred, const float* label, int n, float* out6, float* dpred, void* stream);
/* Evaluation path of the segmentation task (reference cinema/segmentation/train.py:148-286, cinema/transform.py:86-124, cinema/metric.py:21-45,84-96).
 *   seg_window_accumulate: one sliding window - softmax over the c classes of every window voxel (window_logits fp32 channels-last rows
 *       [px*py*pz][c]) added into prob_sum (channels-last [X*Y*Z][c]) at offset (sx,sy,sz), count [X*Y*Z] += 1 (both zeroed by the caller).
 *   seg_window_finish: logits_out [c][X*Y*Z] (channels-first) = log(prob_sum / count).
 *   seg_metric_counts: counts [b][c][6] (uint32, zeroed here) of voxels per (sample, class): argmax prediction, label, both; and the stability-score
 *       masks (logit - mean over classes >= +1, >= -1, both).  logits fp32 channels-first [b][c][vox], labels int32 [b][vox]. */
int cinema_seg_window_accumulate(const float* window_logits, int c, int px, int py, int pz, int sx, int sy, int sz, int X, int Y, int Z,
                                 float* prob_sum, float* count, void* stream);
int cinema_seg_window_finish(const float* prob_sum, const float* count, int c, long long n_voxels, float* logits_out, void* stream);
int cinema_seg_metric_counts(const float* logits, const int* labels, int b, int vox, int c, unsigned int* counts, void* stream);
/* The two device pieces of `hausdorff_distance_95` in segmentation_metrics (reference cinema/segmentation/train.py:262-267,277,284 -> monai 1.5.2
 * compute_hausdorff_distance(percentile=95, spacing), absent from the image; restated from its published algorithm = scipy binary_erosion + distance_transform_edt):
 *   mask_edges: edges uint8 [b][c][X*Y*Z] = surface voxels of every class of the label map int32 [b][X][Y][Z] (mask XOR erosion with the face-neighbour cross,
 *       border value 0); ndim = 2: X == 1 and the x axis does not exist.
 *   min_dist: out[i] = min_j |a_i - b_j| for point sets a [na][3], b [nb][3] in physical units (= the Euclidean distance transform of B's complement at a_i). */
int cinema_mask_edges(const int* label, int b, int X, int Y, int Z, int c, int ndim, unsigned char* edges, void* stream);
int cinema_min_dist(const float* a, const float* b_points, int na, int nb, float* out, void* stream);

/* Token pooling of the ConvViT heads (reference cinema/convvit.py:523-547, `x.mean(dim=1, keepdim=True)` and the mean over head outputs):
 * out[s][:] = scale * sum of the seg_rows consecutive rows of segment s of x (fp32 [n_seg*seg_rows][c]); bwd broadcasts scale * dy[s] back. */
int cinema_segment_mean_fwd(const float* x, int ldx, int n_seg, int seg_rows, int c, float scale, float* out, void* stream);
int cinema_segment_mean_bwd(const float* dy, int n_seg, int seg_rows, int c, float scale, float* dx, int lddx, int accumulate, void* stream);
/* y = alpha * x (fp32) */
int cinema_scale_f32(const float* x, float alpha, float* y, long long n, void* stream);
/* y[i] = x[i] * s[0], the scalar s read from device memory (chain rule through the scalar loss mean, cinema/mae/mae.py:604-608). */
int cinema_mul_scalar_f32(const float* x, const float* s, float* y, long long n, void* stream);
/* y[r][j] = a[r][j] * (b_rows ? b[r][j] : b[j]) on contiguous [rows][c] tensors (fp32 or bf16 each).  timm LayerScale as used at cinema/vit.py:561,576
 * (forward x * gamma, backward dy * gamma), and - with a full second operand - the element products behind d gamma = column sums of dy * x (cinema_colsum)
 * and the GELU derivative of an Mlp whose activation is followed by nn.Dropout (timm Mlp, cinema/vit.py:570-575 with proj_drop > 0). */
int cinema_mul_rows(const void* a, int a_bf16, const void* b, int b_bf16, int b_rows, void* y, int y_bf16, long long rows, int c, void* stream);

/* Rotary embedding of q / k as the reference calls it (cinema/vit.py:496-499 -> cinema/rotary.py:30-60, 109-128): the table row is the HEAD index
 * (q, k are (batch, heads, tokens, head_dim) and the module indexes dim 1), the rotation acts on the two halves of the first rotary_dim columns of
 * every head (rotate_half, cinema/rotary.py:12-24).  In place on bf16 rows [rows][ld]: n_slots consecutive head slots of head_dim columns from
 * column 0, slot s uses table row s % heads (fused q|k rows: n_slots = 2*heads).  cos/sin fp32 [heads][rotary_dim/2].  inverse=1: transposed
 * rotation (backward pass). */
int cinema_rope_heads(uint16_t* x, int ld, long long rows, int n_slots, int heads, int head_dim, int rotary_dim, const float* cos_table,
                      const float* sin_table, int inverse, void* stream);
/* Stochastic regularisers of the fine-tuning recipes: nn.Dropout inside ConvResBlock (cinema/conv.py:329,343) and timm DropPath around both paths of
 * a transformer Block (cinema/vit.py:561-577,606-609); cinema/segmentation/acdc/config.yaml:64-65 sets both to 0.1.  Counter-based Philox4x32-10:
 * the draw for element i of call site `salt` is a function of (state[1] = seed, state[0] = step, salt, i) - the backward pass regenerates the mask
 * (call the same function on the gradient), nothing is stored; state is 2 x uint64 in device memory, cinema_rng_advance bumps the step.
 *   dropout_bf16  : y[i] = keep(i) ? x[i] / (1 - p) : 0                       (x may equal y)
 *   droppath_scale: scale[b] = keep(b) ? 1 / (1 - p) : 0 per sample           (timm DropPath, scale_by_keep=True)
 *   scale_rows_add: out[r,:] = residual[r,:] + scale[r / rows_per_sample] * h[r,:]  (fp32 rows; residual may be NULL: the backward of the same op) */
int cinema_rng_advance(unsigned long long* state, void* stream);
int cinema_dropout_bf16(const uint16_t* x, uint16_t* y, long long n, float p, const unsigned long long* state, unsigned int salt, void* stream);
int cinema_droppath_scale(float* scale, int batch, float p, const unsigned long long* state, unsigned int salt, void* stream);
int cinema_scale_rows_add(const float* h, const float* residual, const float* scale, float* out, long long rows, int c, int rows_per_sample, void* stream);
/* out16[r,:] = bf16(scale[r / rows_per_sample] * h[r,:]): the gradient of a DropPath branch (timm DropPath backward, cinema/vit.py:606-609) in the dtype its
 * only readers - the branch's weight- and data-gradient GEMMs - take (no fp32 tensor, no cast launch behind it). */
int cinema_scale_rows_bf16(const float* h, const float* scale, uint16_t* out, long long rows, int c, int rows_per_sample, void* stream);
/* Thin linear layer (n <= 8 outputs, k <= 64 inputs, k % 8 == 0; the 4-class segmentation head, cinema/segmentation/convunetr.py pred_head_dict, over millions of
 * voxels): streaming kernels instead of a GEMM.  fwd: y fp32 [rows][n] = x bf16 [rows][k] . w^T (fp32 [n][k]) + bias.  bwd (dy fp32 [rows][n]): dx bf16 [rows][k] =
 * dy . w (NULL: skip), dw fp32 [n][k] += dy^T x, db [n] += column sums of dy (NULL: skip). */
int cinema_thin_linear_fwd(const uint16_t* x, const float* w, const float* bias, float* y, long long rows, int n, int k, void* stream);
int cinema_thin_linear_bwd(const uint16_t* x, const float* w, const float* dy, uint16_t* dx, float* dw, float* db, long long rows, int n, int k, void* stream);
/* The mirrored case, few inputs and a few dozen outputs (the 1x1 shortcut convolution of the raw-image ConvResBlock, 1 -> 32 channels over every voxel,
 * cinema/conv.py:322-327): k <= 8, n in {4, 8, 16, 32, 64}; x bf16 [rows][k], w fp32 [n][k], y / dy fp32 [rows][n] (16-byte aligned); same contract as the thin pair. */
int cinema_fanout_linear_fwd(const uint16_t* x, const float* w, const float* bias, float* y, long long rows, int n, int k, void* stream);
int cinema_fanout_linear_bwd(const uint16_t* x, const float* w, const float* dy, uint16_t* dx, float* dw, float* db, long long rows, int n, int k, void* stream);
/* "Same" convolution of a ONE-channel channels-last volume x bf16 [b*X*Y*Z] with the fp32 master weight w [n][kx*ky*kz] (torch layout (n, 1, kx, ky, kz); extents 1 or 3,
 * n in {4, 8, 16, 32, 64}) as direct stencil kernels - the first conv of the raw-image ConvResBlock (cinema/conv.py:320-345, in_chans = 1), too thin for the MFMA
 * path: y fp32 [rows][n] = conv(x, w) + bias.  bwd: dy fp32 [rows][n]; dw [n][taps] / db [n] accumulated (NULL: skipped), dx bf16 [rows] written (NULL: skipped). */
int cinema_conv1ch_fwd(const uint16_t* x, const float* w, const float* bias, float* y, int b, int X, int Y, int Z, int kx, int ky, int kz, int n, void* stream);
int cinema_conv1ch_bwd(const uint16_t* x, const float* w, const float* dy, uint16_t* dx, float* dw, float* db, int b, int X, int Y, int Z, int kx, int ky, int kz, int n,
                       void* stream);
/* dst[0 .. n_words) = word (32-bit pattern; torch.zeros / torch.full of the reference's host code as a launch of this library). */
int cinema_fill_u32(void* dst, unsigned int word, long long n_words, void* stream);

/* Stream ordering (no reference counterpart: torch hides its streams; the reference runs one).  The host launches the weight-gradient
 * GEMMs of the backward pass on a second stream; these three calls are the fork / completion bookkeeping, on per-device event rings.
 *   cinema_stream_fork:   to_stream waits for everything queued on from_stream so far.
 *   cinema_marker_record: returns a ticket (>= 0) that completes when everything queued on `stream` so far has run; at most 4096
 *                         tickets may be outstanding per device.
 *   cinema_marker_done:   1 = complete, 0 = still running, < 0 = error. */
int cinema_stream_fork(void* from_stream, void* to_stream);
/* Lane groups: n independent, identically shaped launch sequences (the three long-axis views of an MAE step run the same ~45 forward / ~95 backward
 * tiny stem kernels on different pointers; their cost is launch latency, not work).  Between cinema_lanes_begin(n) (n <= 4) and cinema_lanes_end()
 * every launch of this library - and every cinema_stream_fork - is RECORDED in the current lane's sequence instead of being issued;
 * cinema_lanes_select(i) switches the lane.  cinema_lanes_end walks the n sequences in lock step: a position at which all lanes hold the same kernel
 * with the same launch geometry goes out as ONE launch of the kernel's lanes form (grid x n, one parameter block per lane); anything else goes out one
 * by one in lane order (also when the sequences differ in length).  Contract: the lanes are independent of each other; buffers touched inside the
 * group stay allocated until the group ends; scratch handed to the library is per lane.  merged_out / single_out (may be NULL) receive the number of
 * merged / single kernel launches issued. */
int cinema_lanes_begin(int n);
int cinema_lanes_select(int lane);
int cinema_lanes_end(int* merged_out, int* single_out);
/* error paths: close an open group WITHOUT issuing its recorded launches (cinema_lanes_begin also discards a stale open group) */
int cinema_lanes_abort(void);
long long cinema_marker_record(void* stream);
int cinema_marker_done(long long ticket);
/* Kernels this library has handed to the HIP runtime since it was loaded (a merged lane-group launch counts once; stream forks and markers are not kernels):
 * bench.py reports the difference over the timed steps as config.kernel_launches_per_step. */
long long cinema_kernel_launch_count(void);
/* n back-to-back launches of an empty kernel (measures the host cost of one launch; used by tools/launch_rate.py and DESIGN.md section 5). */
int cinema_launch_probe(int n, void* stream);
/* grid x 4 waves x iters x 16 independent v_mfma_f32_32x32x16_bf16 from registers: the sustained rate of the matrix pipe alone
 * (clock / power limits included), the practical ceiling the GEMM kernels are priced against in DESIGN.md (tools/mfma_peak.py). */
int cinema_mfma_probe(int grid, int iters, float* out, void* stream);

/* Random-mask bookkeeping (cinema/mae/mae.py:30-65 get_batch_random_patch_mask, :550 boolean-mask indexing; cinema/convvit.py:153-170).
 *   cinema_mask_select: noise != NULL: mask[b][i] = rank of noise[b][i] in its row >= n_keep (True = removed; ties by index, as
 *       argsort(argsort(noise)) with a stable sort); noise == NULL: mask is an input.  Then, for the lists that are not NULL, the kept /
 *       dropped tokens of every row in raster order: positions i (keep_pos / drop_pos) and flat ids b*n + i (keep / drop).  Every row of
 *       `mask` must hold the same number of kept tokens when lists are requested.
 *   cinema_visible_index: rank[keep[r]] = r (rank pre-filled with -1) and idx1[r*block_vol + q] = flat id of the stage-1 voxel stored at row
 *       offset q of kept token r (inv1[q] = raster index inside the token's block); grid / block: n_dims host ints (tokens per sample, voxels per token). */
int cinema_mask_select(const float* noise, uint8_t* mask, int batch, int n, int n_keep, int* keep_pos, int* drop_pos, int* keep, int* drop, void* stream);
int cinema_visible_index(const int* keep, int n_rows, int n_dims, const int* grid_host, const int* block_host, const int* inv1, int* rank, int* idx1,
                         void* stream);

/* k == s / dense conv weights (out, c, *k) fp32 <-> GEMM operand rows [out][ld] in the patch feature order (*k, c)  (the re-layout the
 * reference gets for free from cuDNN/MIOpen's own filter layouts; here it feeds cinema_gemm_bf16):
 *   direction 0: rows (bf16 or fp32) <- w, columns beyond kvol*c zero-filled;   direction 1: w += rows (fp32), the gradient way back.
 *   jmap[kvol] (optional): row feature block jj holds kernel voxel jmap[jj] (visible-voxel stem ordering). */
int cinema_patch_weight_relayout(float* w, void* rows, int rows_is_bf16, int outer, int c, int kvol, int ld, const int* jmap, int direction, void* stream);

/* The same for the k == s transposed convolutions of UpsampleDecoder (cinema/segmentation/convunetr.py:62-101): w fp32 (c_in, c_out, kvol) <-> GEMM rows
 * [(kv, co)][ci]; direction 0: rows bf16 <- w and, with bias, bias_t[kv * c_out + co] = bias[co]; direction 1: w += rows (fp32 gradient rows). */
int cinema_convt_weight_relayout(float* w, void* rows, int c_in, int c_out, int kvol, int direction, const float* bias, float* bias_t, void* stream);

/* elementwise: dtype codes as above */
int cinema_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long long n, void* stream);
/* dst[c][r] = src[r][c] (bf16 out); src fp32 or bf16 */
int cinema_transpose_cast(const void* src, int src_dtype, int rows, int cols, uint16_t* dst, void* stream);
/* y = x * row_mask[row]  on [rows][c] bf16 */
int cinema_gelu_fwd(const uint16_t* x, uint16_t* y, long long n, void* stream);
int cinema_gelu_bwd(const uint16_t* x, const uint16_t* dy, uint16_t* dx, long long n, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Masked-patch MSE (reference: mse_loss cinema/mae/mae.py:107-152 with patchify(image, dec_patch) as target; the
 * target patch is gathered on the fly from the fp32 image through `geom`, never materialised).
 *   fwd : loss_out[0] += inv_count * sum_{rows, f} (pred - target)^2        (rows = masked tokens, token_idx = their ids)
 *   bwd : dpred[row, f] = 2 * host_scale * upstream[0] * (pred - target)      (bf16)
 *   patch_stats: out2[0] += mean over ALL patches of the patch mean, out2[1] += mean of the patch unbiased std
 *                (the `target_mean` / `target_std` metrics, mae.py:129-137)
 *   mean_finite: mean_out[0] = mean of the finite vals (NaN if none), coef_out[i] = d mean / d vals[i]
 *                (the per-view `torch.isfinite` filter + mean over views, mae.py:604-608, without a host sync)
 */
int cinema_mse_fwd(const float* image, const cinema_patch_geom* geom_masked_host, const void* pred, int pred_dtype, int ld_pred,
                   int norm_target, float eps, float inv_count, float* loss_out, float* max_out /* NULL or [2], pre-filled -inf: {max target, max pred} */,
                   void* stream);
int cinema_mse_bwd(const float* image, const cinema_patch_geom* geom_masked_host, const void* pred, int pred_dtype, int ld_pred,
                   int norm_target, float eps, const float* upstream, float host_scale, uint16_t* dpred, int ld_dpred, void* stream);
int cinema_patch_stats(const float* image, const cinema_patch_geom* geom_all_host, float* out2, void* stream);
int cinema_mean_finite(const float* vals, int n, float* mean_out, float* coef_out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * GPU input pipeline (reference cinema/mae/pretrain.py:157-200: monai RandZoomd -> ScaleIntensityd -> SpatialPadd(method="end") in the CPU workers).
 * fp32 single-channel image / volume src [X][Y][Z] (Z = 1 for 2-D).
 *   zoom_resample: monai Zoom(keep_size=True, padding_mode="constant"): interpolate to floor(size*zoom) (align_corners=False; mode 0 = (tri)linear,
 *       mode 1 = bicubic a=-0.75 over (x, y), Z must be 1), centred zero-pad / crop back to [X][Y][Z]; minmax (2 x uint32, order-preserving float
 *       encoding, initialised here) receives min / max of the result.  zoom = 1 is the identity.
 *   scale_intensity_pad: dst [PX][PY][PZ] = (src - min) / (max - min) inside [X][Y][Z], 0 in the end padding; all zeros when max == min. */
int cinema_zoom_resample(const float* src, int X, int Y, int Z, float zoom_x, float zoom_y, float zoom_z, int mode, float* dst, unsigned int* minmax,
                         void* stream);
int cinema_scale_intensity_pad(const float* src, int X, int Y, int Z, const unsigned int* minmax, float* dst, int PX, int PY, int PZ, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Optimiser (reference harness: torch.optim.AdamW + clip_grad_norm_ via GradScaler, cinema/optim.py:204-215,
 * cinema/mae/pretrain.py:365-366).  Flat fp32 buffers.
 *   sqnorm: out[0] += sum g^2, deterministic (per-block partials in `workspace`, >= 2048 floats of scratch, then one fixed-order sum).
 *   clip_coef: coef = min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: 1); a NON-FINITE norm gives coef = 0 = "skip this update"
 *           (cinema/mae/pretrain.py:255-257 NaN-loss skip; torch GradScaler.step inf/NaN skip).  step_state (int32[2], may be NULL):
 *           [0] += 1 when the update will be applied, [1] += 1 when it is skipped.
 *   adamw : p,m,v updated in place; grad is multiplied by *clip_coef (device scalar, may be NULL=1);
 *           optional bf16 shadow copy of the updated parameter written to p_bf16.  With step_state (needs clip_coef) the kernel returns
 *           without touching anything when *clip_coef is not > 0, and takes the Adam step for the bias corrections from step_state[0]
 *           (bias_corr1/2 are ignored) - the whole NaN decision stays on the device.
 */
int cinema_sqnorm_f32(const float* g, long long n, float* out, float* workspace, void* stream);
int cinema_clip_coef(const float* sqnorm, float max_norm, float* coef_out, float* norm_out, int* step_state, void* stream);
int cinema_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, float bias_corr1, float bias_corr2, const float* clip_coef, uint16_t* p_bf16, const int* step_state,
                 void* stream);
/* The same update for up to CINEMA_ADAMW_MAX_GROUPS parameter groups of ONE flat buffer in one launch (torch.optim.AdamW param_groups with their own lr /
 * weight_decay: the layer-decay groups of cinema/convvit.py param_groups_lr_decay, cinema/train.py:262-268 - 28 launches per step for ViT-Base otherwise).
 * groups[i] = element range [begin, end) (both multiples of 4), lr, weight_decay; the ranges must be ascending and non-overlapping (gaps are left untouched).
 * Same arithmetic per element as cinema_adamw (bit-identical results). */
#define CINEMA_ADAMW_MAX_GROUPS 64
typedef struct { long long begin, end; float lr, weight_decay; } cinema_adamw_group;
int cinema_adamw_groups(float* p, const float* g, float* m, float* v, const cinema_adamw_group* groups_host, int n_groups, float beta1, float beta2, float eps,
                        const float* clip_coef, uint16_t* p_bf16, const int* step_state, void* stream);
/* ... with the number of workgroups capped at max_blocks (0: the default, 4096 x 256 threads, grid-stride): an update that runs BESIDE other work on another
 * stream (cinema_amd/optim.py TrainStep(overlap_update=True): the next step's convolution stems) must not take every workgroup slot of the chip. */
int cinema_adamw_groups_grid(float* p, const float* g, float* m, float* v, const cinema_adamw_group* groups_host, int n_groups, float beta1, float beta2, float eps,
                             const float* clip_coef, uint16_t* p_bf16, const int* step_state, int max_blocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif
