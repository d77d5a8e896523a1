// Shared pieces of the bf16 GEMM kernels (gemm.hip: 128x128 tiles, gemm256.hip: persistent 256x256 tiles): the launch parameter block,
// the fused epilogue (bias / GELU / GELU' / residual / bf16 + fp32 outputs) staged through LDS, the operand tile loaders (LDS-DMA with the
// swizzle on the source address) and fragment readers, and the XCD-aware tile order.
#pragma once
#include <cstdlib>
#include "common.cuh"
#include "../../include/cinema_hip.h"

namespace {

constexpr int TAIL_ERROR_WORD = 2047;  // GemmP::tail_cnt: 8 KiB = 2 counters per tail tile (<= 1023 tiles) + this word, set when a bounded spin gives up
struct GemmP {
  const bf16_t* a; const bf16_t* b; void* d;
  int m, n, k, lda, ldb, ldd;
  float alpha;
  const float* bias;
  const float* res_f32; const bf16_t* res_bf16; int ld_res;
  const bf16_t* gelu_in; int ld_gelu;
  const uint8_t* row_mask;
  bf16_t* aux_out; int ld_aux;
  int act, out_f32, accumulate;
  int gelu_deriv;  // 1: the auxiliary GELU tensor holds GELU'(pre-activation), not the pre-activation: act == 1 writes it to aux_out, gelu_in is multiplied in as it is
                   // 2: the same as an 8-bit code (common.cuh gelu8_*): aux_out / gelu_in are uint8 rows, ld_aux / ld_gelu count bytes
  int ktiles_per_split;
  float* ws;  // split-K partial slabs [gridDim.z][m][n] (fp32) or nullptr
  float* a_rowsum;  // optional: a_rowsum[m] += sum_k A[m][k] (the bias gradient of a weight-gradient GEMM), fused into the MFMA loop
  // split tail (128x128 kernel, gridDim.z == 1): logical tiles >= tail_begin do not fill the last round of workgroup slots, so each
  // is cut into tail_split k-slices (tail_ktiles k-tiles each) whose fp32 partial tiles go to tail_ws[(tile - tail_begin) * tail_split
  // + slice][128][128]; tail_fixup_kernel sums the slices and runs the fused epilogue.  tail_split == 0: off.
  // tail_cnt != NULL: the slices are finished INSIDE the launch by the tile's last arriver (two counters per tail tile, zero on entry, left zero;
  // the partial tiles are then fragment-ordered) and no fix-up launch follows.
  int tail_begin, tail_split, tail_ktiles;
  float* tail_ws;
  unsigned* tail_cnt;
  // fp8 (e4m3) operands: per-tensor dequantisation scales in device memory (the product multiplies alpha); NULL for bf16 operands
  const float* scale_a; const float* scale_b;
  int scale_a_rows;  // 1: scale_a holds one scale per row of A (per-token activation scaling), 0: one scalar
  // implicit-GEMM "same" convolution (MODE_CONV): A is not a stored matrix but the channels-last volume x [batch][cX][cY][cZ][cC] (p.a); row r of the
  // virtual im2col matrix is output voxel r, its 16-byte k-chunk j (8 channels of one tap) is read at x[(r + tap_rows[j]) * cC + tap_ci[j]] when
  // the neighbour lies inside the volume, zeros otherwise.  conv_taps[j] = {row delta, packed (dx+1, dy+1, dz+1), first channel, valid}.
  const int4* conv_taps;
  int cX, cY, cZ, cC;
  int cZB;  // z-blocking of the implicit convolution (see cinema_conv_gemm_bf16): a row is a group of cZB consecutive z voxels; 1 = plain
  // weight gradient of that convolution (MODE_CONVW): dW[co][(tap, ci)] = sum_r dy[r][co] * x[nbr_tap(r)][ci]: the B operand (reduction-strided,
  // [rows][taps * C]) is the virtual im2col matrix; conv_coords[r] = x | y << 10 | z << 20 of voxel r (one int per row, shape-only table)
  const int* conv_coords;
  // optional 8-bit copy of a bf16 output with per-tensor delayed scaling (cinema_q8_out semantics; the epilogue classes with a bf16 D): out8[m][ld_out8] =
  // e4m3(sat(D * *out8_inv)); the launch's max|D| goes into out8_amax[64] (also without out8: calibration)
  uint8_t* out8 = nullptr; int ld_out8 = 0; const float* out8_inv = nullptr; unsigned int* out8_amax = nullptr;
  // optional (bf16-output classes): column sums of D over every strip of 32 rows, colsum_partials[m / 32][n] fp32 (every element written exactly once: no
  // zeroing, no atomics); summed over the strips they are the bias gradient of the layer whose dY this GEMM produces
  float* colsum_partials = nullptr;
};

__device__ __forceinline__ float frag_sum8(const short8v& f) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; e++) s += bf2f((bf16_t)f[e]);
  return s;
}

// ---- epilogue on 4 consecutive columns (n0..n0+3) of row m; n0 % 4 == 0 and n0+3 < N guaranteed by the caller
__device__ __forceinline__ void epilogue4(const GemmP& p, int m, int n0, float v0, float v1, float v2, float v3, bool add_bias) {
  float v[4] = {v0 * p.alpha, v1 * p.alpha, v2 * p.alpha, v3 * p.alpha};
  if (p.bias && add_bias) {
    const float4 bv = *reinterpret_cast<const float4*>(p.bias + n0);
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
  }
  if (p.aux_out && p.act == 1 && p.gelu_deriv == 2) {
    *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(p.aux_out) + (size_t)m * p.ld_aux + n0) =
        gelu8_pack4(gelu_grad_f(v[0]), gelu_grad_f(v[1]), gelu_grad_f(v[2]), gelu_grad_f(v[3]));
  } else if (p.aux_out) {
    const bool dv = p.act == 1 && p.gelu_deriv;  // the derivative instead of the pre-activation (see GemmP::gelu_deriv)
    uint2 pk;
    pk.x = dv ? pack_bf2(gelu_grad_f(v[0]), gelu_grad_f(v[1])) : pack_bf2(v[0], v[1]);
    pk.y = dv ? pack_bf2(gelu_grad_f(v[2]), gelu_grad_f(v[3])) : pack_bf2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p.aux_out + (size_t)m * p.ld_aux + n0) = pk;
  }
  if (p.act == 1) {
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = gelu_f(v[i]);
  }
  if (p.gelu_in && p.gelu_deriv == 2) {
    const uint32_t gi = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(p.gelu_in) + (size_t)m * p.ld_gelu + n0);
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] *= gelu8_dec(gi, i);
  } else if (p.gelu_in) {
    const uint2 gi = *reinterpret_cast<const uint2*>(p.gelu_in + (size_t)m * p.ld_gelu + n0);
    const float g0 = bf2f((bf16_t)(gi.x & 0xffff)), g1 = bf2f((bf16_t)(gi.x >> 16)), g2 = bf2f((bf16_t)(gi.y & 0xffff)), g3 = bf2f((bf16_t)(gi.y >> 16));
    v[0] *= p.gelu_deriv ? g0 : gelu_grad_f(g0); v[1] *= p.gelu_deriv ? g1 : gelu_grad_f(g1);
    v[2] *= p.gelu_deriv ? g2 : gelu_grad_f(g2); v[3] *= p.gelu_deriv ? g3 : gelu_grad_f(g3);
  }
  if (p.row_mask) {
    const float s = p.row_mask[m] ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] *= s;
  }
  if (p.res_f32) {
    const float4 rv = *reinterpret_cast<const float4*>(p.res_f32 + (size_t)m * p.ld_res + n0);
    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
  } else if (p.res_bf16) {
    const uint2 rv = *reinterpret_cast<const uint2*>(p.res_bf16 + (size_t)m * p.ld_res + n0);
    v[0] += bf2f((bf16_t)(rv.x & 0xffff)); v[1] += bf2f((bf16_t)(rv.x >> 16));
    v[2] += bf2f((bf16_t)(rv.y & 0xffff)); v[3] += bf2f((bf16_t)(rv.y >> 16));
  }
  if (p.out_f32) {
    float* dp = reinterpret_cast<float*>(p.d) + (size_t)m * p.ldd + n0;
    if (p.accumulate) {
#pragma unroll
      for (int i = 0; i < 4; i++) unsafeAtomicAdd(dp + i, v[i]);
    } else {
      *reinterpret_cast<float4*>(dp) = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else {
    uint2 pk; pk.x = pack_bf2(v[0], v[1]); pk.y = pack_bf2(v[2], v[3]);
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.d) + (size_t)m * p.ldd + n0) = pk;
  }
}

// ---- epilogue on W (4 or 8) consecutive columns n0.. of row m (all 16-byte aligned; n0 + W <= N guaranteed by the caller), split into
// a LOAD half and an APPLY half: the staged tile epilogue issues the loads of all its sub-blocks up front, so the residual / GELU-input
// reads are in flight while the accumulators go through LDS (as one function the loads sat in the dependency chain of every 32x32
// sub-block: +20 us for a bias, +130 us for bias + fp32 residual on the 32848x2048 decoder GEMM).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4e __attribute__((ext_vector_type(4)));
// The outputs stream out (67-270 MB per GEMM against 4 MiB of L2 per XCD): non-temporal stores keep them from evicting the operand panels the
// resident tiles share - 31.23 -> 30.99 ms/step in a same-box A/B.
constexpr bool NT_STORES = true;
template <int W>
struct EpiPre {   // plain vector members (arrays inside the struct were left in scratch memory by the compiler)
  u32x4 ra, rb;   // fp32 residual words 0-3 / 4-7, or packed bf16 residual in ra (W/2 words)
  u32x4 g;        // packed bf16 GELU input (pre-activation of the forward pass), W/2 words
  float keep;     // row mask as 0/1
};
__device__ __forceinline__ u32x4 ldg128(const void* ptr) { return *reinterpret_cast<const u32x4*>(ptr); }
__device__ __forceinline__ u32x4 ldg64(const void* ptr) {
  const uint2 u = *reinterpret_cast<const uint2*>(ptr);
  u32x4 r = {u.x, u.y, 0u, 0u};
  return r;
}
template <int W>
__device__ __forceinline__ void epi_load_bias(const GemmP& p, int n0, bool add_bias, float (&bv)[W]) {
#pragma unroll
  for (int i = 0; i < W; i++) bv[i] = 0.f;
  if (p.bias && add_bias && n0 < p.n) {
#pragma unroll
    for (int i = 0; i < W; i += 4) {
      const float4 t = *reinterpret_cast<const float4*>(p.bias + n0 + i);
      bv[i] = t.x; bv[i + 1] = t.y; bv[i + 2] = t.z; bv[i + 3] = t.w;
    }
  }
}
template <int W>
__device__ __forceinline__ void epi_load(const GemmP& p, int m, int n0, EpiPre<W>& e) {
  e.keep = 1.f;
  if (p.gelu_in) {
    if (p.gelu_deriv == 2) {  // W bytes of the 8-bit derivative code
      const uint8_t* g8 = reinterpret_cast<const uint8_t*>(p.gelu_in) + (size_t)m * p.ld_gelu + n0;
      if (W == 8) e.g = ldg64(g8);
      else { const uint32_t u = *reinterpret_cast<const uint32_t*>(g8); u32x4 r = {u, 0u, 0u, 0u}; e.g = r; }
    } else {
      e.g = W == 8 ? ldg128(p.gelu_in + (size_t)m * p.ld_gelu + n0) : ldg64(p.gelu_in + (size_t)m * p.ld_gelu + n0);
    }
  }
  if (p.row_mask) e.keep = p.row_mask[m] ? 1.f : 0.f;
  if (p.res_f32) {
    e.ra = ldg128(p.res_f32 + (size_t)m * p.ld_res + n0);
    if (W == 8) e.rb = ldg128(p.res_f32 + (size_t)m * p.ld_res + n0 + 4);
  } else if (p.res_bf16) {
    e.ra = W == 8 ? ldg128(p.res_bf16 + (size_t)m * p.ld_res + n0) : ldg64(p.res_bf16 + (size_t)m * p.ld_res + n0);
  }
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
template <int W>
__device__ __forceinline__ void epi_apply(const GemmP& p, int m, int n0, float (&v)[W], const float (&bv)[W], const EpiPre<W>& e, float& q8max, float q8inv = 0.f) {
  const float al = p.scale_a_rows ? p.alpha * p.scale_a[m] : p.alpha;  // fp8 operands with per-row activation scales (p.alpha already holds scale_b)
#pragma unroll
  for (int i = 0; i < W; i++) v[i] = fmaf(v[i], al, bv[i]);
  auto store_bf16 = [&](bf16_t* dst) {
    if (W == 8) {
      u32x4 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[W - 4], v[W - 3]), pack_bf2(v[W - 2], v[W - 1])};
      if (NT_STORES) __builtin_nontemporal_store(pk, reinterpret_cast<u32x4*>(dst));
      else *reinterpret_cast<u32x4*>(dst) = pk;
    } else {
      uint2 pk; pk.x = pack_bf2(v[0], v[1]); pk.y = pack_bf2(v[2], v[3]);
      *reinterpret_cast<uint2*>(dst) = pk;
    }
  };
  if (p.act == 1 && p.aux_out && p.gelu_deriv) {
    // GELU and its derivative from ONE evaluation of the erf terms; the derivative (bf16) is what the data gradient of this layer multiplies in later
    float dv[W];
#pragma unroll
    for (int i = 0; i < W; i += 2) gelu_both2(v[i], v[i + 1], dv[i], dv[i + 1]);
    bf16_t* dst = p.aux_out + (size_t)m * p.ld_aux + n0;
    if (p.gelu_deriv == 2) {
      uint8_t* d8 = reinterpret_cast<uint8_t*>(p.aux_out) + (size_t)m * p.ld_aux + n0;
      if (W == 8) {
        typedef uint32_t u32x2e __attribute__((ext_vector_type(2)));
        const u32x2e pk = {gelu8_pack4(dv[0], dv[1], dv[2], dv[3]), gelu8_pack4(dv[W - 4], dv[W - 3], dv[W - 2], dv[W - 1])};
        if (NT_STORES) __builtin_nontemporal_store(pk, reinterpret_cast<u32x2e*>(d8));
        else *reinterpret_cast<u32x2e*>(d8) = pk;
      } else {
        *reinterpret_cast<uint32_t*>(d8) = gelu8_pack4(dv[0], dv[1], dv[2], dv[3]);
      }
    } else if (W == 8) {
      u32x4 pk = {pack_bf2(dv[0], dv[1]), pack_bf2(dv[2], dv[3]), pack_bf2(dv[W - 4], dv[W - 3]), pack_bf2(dv[W - 2], dv[W - 1])};
      if (NT_STORES) __builtin_nontemporal_store(pk, reinterpret_cast<u32x4*>(dst));
      else *reinterpret_cast<u32x4*>(dst) = pk;
    } else {
      uint2 pk; pk.x = pack_bf2(dv[0], dv[1]); pk.y = pack_bf2(dv[2], dv[3]);
      *reinterpret_cast<uint2*>(dst) = pk;
    }
  } else {
    if (p.aux_out) store_bf16(p.aux_out + (size_t)m * p.ld_aux + n0);
    if (p.act == 1) {
#pragma unroll
      for (int i = 0; i < W; i += 2) gelu2(v[i], v[i + 1]);
    }
  }
  if (p.gelu_in) {
    if (p.gelu_deriv == 2) {
#pragma unroll
      for (int i = 0; i < W; i++) v[i] *= gelu8_dec(e.g[i >> 2], i & 3);
    } else if (p.gelu_deriv) {
#pragma unroll
      for (int i = 0; i < W / 2; i++) { v[2 * i] *= bf_lo(e.g[i]); v[2 * i + 1] *= bf_hi(e.g[i]); }
    } else {
#pragma unroll
      for (int i = 0; i < W / 2; i++) {
        float ga, gb;
        gelu_grad2(bf_lo(e.g[i]), bf_hi(e.g[i]), ga, gb);
        v[2 * i] *= ga; v[2 * i + 1] *= gb;
      }
    }
  }
  if (p.row_mask) {
#pragma unroll
    for (int i = 0; i < W; i++) v[i] *= e.keep;
  }
  if (p.res_f32) {
#pragma unroll
    for (int i = 0; i < W; i++) v[i] += __uint_as_float(i < 4 ? e.ra[i & 3] : e.rb[i & 3]);
  } else if (p.res_bf16) {
#pragma unroll
    for (int i = 0; i < W / 2; i++) { v[2 * i] += bf_lo(e.ra[i]); v[2 * i + 1] += bf_hi(e.ra[i]); }
  }
  if (p.out_f32) {
    float* dp = reinterpret_cast<float*>(p.d) + (size_t)m * p.ldd + n0;
#pragma unroll
    for (int i = 0; i < W; i += 4) {
      f32x4e o = {v[i], v[i + 1], v[i + 2], v[i + 3]};
      if (NT_STORES) __builtin_nontemporal_store(o, reinterpret_cast<f32x4e*>(dp + i));
      else *reinterpret_cast<f32x4e*>(dp + i) = o;
    }
  } else {
    if (p.d) store_bf16(reinterpret_cast<bf16_t*>(p.d) + (size_t)m * p.ldd + n0);  // (omitted when only the 8-bit copy below is wanted)
    if (p.out8_amax) {  // 8-bit copy of the same values (per-tensor delayed scale) + this launch's maximum
#pragma unroll
      for (int i = 0; i < W; i++) q8max = fmaxf(q8max, fabsf(v[i]));
      if (p.out8) {
        // (the caller loads the scale ONCE per tile, ahead of the staging: read here it was a vector load + s_waitcnt vmcnt(0) in every row pass, which on
        // gfx9 also drains the pass's output stores - 15 us on a 13824 x 4096 GEMM)
        const float inv = q8inv != 0.f ? q8inv : *p.out8_inv;
        uint8_t* dst8 = p.out8 + (size_t)m * p.ld_out8 + n0;
        if (W == 8) { uint2 pk; pk.x = (uint32_t)q8_pack4(v[0], v[1], v[2], v[3], inv); pk.y = (uint32_t)q8_pack4(v[W - 4], v[W - 3], v[W - 2], v[W - 1], inv); *reinterpret_cast<uint2*>(dst8) = pk; }
        else *reinterpret_cast<int*>(dst8) = q8_pack4(v[0], v[1], v[2], v[3], inv);
      }
    }
  }
}
template <int W>
__device__ __forceinline__ void epi_apply(const GemmP& p, int m, int n0, float (&v)[W], const float (&bv)[W], const EpiPre<W>& e) {
  float unused = 0.f;
  epi_apply<W>(p, m, n0, v, bv, e, unused);
}
template <int W>
__device__ __forceinline__ void epilogue_row(const GemmP& p, int m, int n0, float (&v)[W], bool add_bias) {
  float bv[W];
  EpiPre<W> e;
  epi_load_bias<W>(p, n0, add_bias, bv);
  epi_load<W>(p, m, n0, e);
  epi_apply<W>(p, m, n0, v, bv, e);
}

// Epilogue classes of the 128x128 kernel, fixed at compile time.  With every flag tested at run time the kernel was 69.5 KB of code - more
// than the 64 KiB instruction cache its CU pair shares - and a workgroup spent 4.3 us (plain bf16) to 7.4 us (GELU + pre-activation) in an
// epilogue whose arithmetic takes a few hundred cycles (per-phase timestamps, tools/gemm_phase_timing.py); with the flags folded the same
// code is a fraction of that size.  The host (cinema_gemm_bf16) maps the argument combination to a class; GENERAL keeps all of them.
enum { EPI_GENERAL = 0, EPI_BF16 = 1, EPI_BF16_GELU = 2, EPI_BF16_GELU_GRAD = 3, EPI_F32 = 4 };
template <int EPI>
__device__ __forceinline__ GemmP epi_fold(GemmP p) {  // a by-value copy with the fields the class fixes set to constants (the compiler folds the tests)
  if (EPI == EPI_GENERAL) return p;
  p.row_mask = nullptr; p.res_bf16 = nullptr; p.accumulate = 0;
  if (EPI == EPI_BF16) { p.out_f32 = 0; p.act = 0; p.aux_out = nullptr; p.gelu_in = nullptr; p.res_f32 = nullptr; }
  if (EPI == EPI_BF16_GELU) { p.out_f32 = 0; p.act = 1; p.gelu_in = nullptr; p.res_f32 = nullptr; }                     // bias + GELU, optional pre-activation out
  if (EPI == EPI_BF16_GELU_GRAD) { p.out_f32 = 0; p.act = 0; p.aux_out = nullptr; p.res_f32 = nullptr; p.bias = nullptr; }  // dY * GELU'(pre-activation)
  if (EPI == EPI_F32) { p.out_f32 = 1; p.act = 0; p.aux_out = nullptr; p.gelu_in = nullptr; p.out8 = nullptr; p.out8_amax = nullptr; p.colsum_partials = nullptr; }                           // optional bias and fp32 residual
  return p;
}

// Fused epilogue of one 64x64 wave tile of the 128x128 kernel through LDS.  All four 32x32 accumulator blocks are staged at once (16 KiB
// per wave: the two operand stages are free after the loop's last barrier), so there is ONE LDS write -> read turn-around per tile instead
// of one per 32x32 block (measured: 0.6-0.9 us each under the other workgroup's loop traffic), and a lane then owns W = 8 (bf16 out) or
// 4 (fp32 out) consecutive columns of one row: every store instruction writes whole 128- / 256-byte row segments.  Staging layout:
// [64 rows][16 chunks of 16 B], chunk index XORed with (row & 15): the 16 lanes of a ds_write_b128 group (16 rows, one chunk column) and
// of a ds_read_b128 group (1-2 rows, 16 chunks) both touch 16 distinct chunks.  Operand loads (bias, residual, GELU input) of all row
// passes are issued before the staging so that they are in flight during it.  ws_base != nullptr: plain fp32 partial tile (split-K slab
// or split-tail slice) instead of the fused epilogue.
template <int W>
__device__ __forceinline__ void tile_epilogue_rows(const GemmP& p, const float16v (&acc)[2][2], int mw, int nw, int lane, bool add_bias, float* stg,
                                                   float* ws_base, long long ws_ld) {
  constexpr int LPR = 64 / W, RPP = 64 / LPR, PASSES = 64 / RPP;  // lanes per row, rows per pass, passes
  const int ml = lane & 31, hi = lane >> 5;
  const int rl = lane / LPR, cl = (lane % LPR) * W;
  const int n = nw + cl;
  float bv[W];
  float q8max = 0.f;
  const float q8inv = (!ws_base && p.out8) ? *p.out8_inv : 0.f;  // in flight during the LDS staging
  constexpr int NSTRIP = (PASSES * RPP) / 32;  // 32-row strips of this call's rows (2 for a 64-row wave tile, 1 for a half)
  float csum[NSTRIP][W];
#pragma unroll
  for (int h = 0; h < NSTRIP; h++)
#pragma unroll
    for (int i = 0; i < W; i++) csum[h][i] = 0.f;
  EpiPre<W> pre[PASSES];
  if (!ws_base) {
    epi_load_bias<W>(p, n, add_bias, bv);
#pragma unroll
    for (int pss = 0; pss < PASSES; pss++) {
      const int m = mw + pss * RPP + rl;
      if (m < p.m && n < p.n) epi_load<W>(p, m, n, pre[pss]);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 4; q++)
        *reinterpret_cast<float4*>(stg + (i * 32 + ml) * 64 + (((j * 8 + 2 * q + hi) ^ (ml & 15)) << 2)) =
            make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
#pragma unroll
  for (int pss = 0; pss < PASSES; pss++) {
    const int r = pss * RPP + rl;
    const int m = mw + r;
    float v[W];
#pragma unroll
    for (int c = 0; c < W / 4; c++) {
      const float4 t = *reinterpret_cast<const float4*>(stg + r * 64 + ((((cl >> 2) + c) ^ (r & 15)) << 2));
      v[4 * c] = t.x; v[4 * c + 1] = t.y; v[4 * c + 2] = t.z; v[4 * c + 3] = t.w;
    }
    if (m < p.m && n < p.n) {
      if (ws_base) {
#pragma unroll
        for (int c = 0; c < W; c += 4) *reinterpret_cast<float4*>(ws_base + (long long)m * ws_ld + n + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
      } else {
        epi_apply<W>(p, m, n, v, bv, pre[pss], q8max, q8inv);
        if (W == 8 && p.colsum_partials) {
#pragma unroll
          for (int i = 0; i < W; i++) csum[(pss * RPP) / 32][i] += v[i];
        }
      }
    }
  }
  if (!ws_base && p.out8_amax) q8_amax_commit(p.out8_amax, q8max, (int)blockIdx.x + (int)(threadIdx.x >> 6));
  if (W == 8 && !ws_base && p.colsum_partials) {  // lanes with the same column group (lane % LPR) hold different rows: butterfly over the row lanes, one store per column group
#pragma unroll
    for (int h = 0; h < NSTRIP; h++) {
#pragma unroll
      for (int i = 0; i < W; i++) {
        float t = csum[h][i];
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
        csum[h][i] = t;
      }
      const int strip = (mw >> 5) + h;
      if (rl == 0 && n < p.n && strip * 32 < p.m) {
        float* dst = p.colsum_partials + (size_t)strip * p.n + n;
        *reinterpret_cast<float4*>(dst) = make_float4(csum[h][0], csum[h][1], csum[h][2], csum[h][3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(csum[h][W - 4], csum[h][W - 3], csum[h][W - 2], csum[h][W - 1]);
      }
    }
  }
}

template <int EPI>
__device__ __forceinline__ void tile_epilogue(const GemmP& p, const float16v (&acc)[2][2], int mw, int nw, int lane, int z, float* stg, float* ws_base = nullptr,
                                              long long ws_ld = 0) {
  if (!ws_base && p.ws) { ws_base = p.ws + (size_t)z * p.m * p.n; ws_ld = p.n; }  // split-K slab of slice z
  if (ws_base) { tile_epilogue_rows<4>(p, acc, mw, nw, lane, false, stg, ws_base, ws_ld); return; }
  const GemmP q = epi_fold<EPI>(p);
  if (EPI == EPI_GENERAL) {
    if (q.out_f32) tile_epilogue_rows<4>(q, acc, mw, nw, lane, z == 0, stg, nullptr, 0);
    else tile_epilogue_rows<8>(q, acc, mw, nw, lane, z == 0, stg, nullptr, 0);
  } else if (EPI == EPI_F32) {
    tile_epilogue_rows<4>(q, acc, mw, nw, lane, z == 0, stg, nullptr, 0);
  } else {
    tile_epilogue_rows<8>(q, acc, mw, nw, lane, z == 0, stg, nullptr, 0);
  }
}

// scalar epilogue for the generic kernel
__device__ __forceinline__ void epilogue1(const GemmP& p, int m, int n, float acc, bool add_bias) {
  float v = acc * p.alpha;
  if (p.bias && add_bias) v += p.bias[n];
  if (p.aux_out && p.act == 1 && p.gelu_deriv == 2) {
    reinterpret_cast<uint8_t*>(p.aux_out)[(size_t)m * p.ld_aux + n] = (uint8_t)(gelu8_pack4(gelu_grad_f(v), 0.f, 0.f, 0.f) & 0xffu);
  } else if (p.aux_out) {
    p.aux_out[(size_t)m * p.ld_aux + n] = f2bf((p.act == 1 && p.gelu_deriv) ? gelu_grad_f(v) : v);
  }
  if (p.act == 1) v = gelu_f(v);
  if (p.gelu_in && p.gelu_deriv == 2) v *= gelu8_dec(reinterpret_cast<const uint8_t*>(p.gelu_in)[(size_t)m * p.ld_gelu + n], 0);
  else if (p.gelu_in) v *= p.gelu_deriv ? bf2f(p.gelu_in[(size_t)m * p.ld_gelu + n]) : gelu_grad_f(bf2f(p.gelu_in[(size_t)m * p.ld_gelu + n]));
  if (p.row_mask) v *= p.row_mask[m] ? 1.f : 0.f;
  if (p.res_f32) v += p.res_f32[(size_t)m * p.ld_res + n];
  else if (p.res_bf16) v += bf2f(p.res_bf16[(size_t)m * p.ld_res + n]);
  if (p.out_f32) {
    float* dp = reinterpret_cast<float*>(p.d) + (size_t)m * p.ldd + n;
    if (p.accumulate) unsafeAtomicAdd(dp, v); else *dp = v;
  } else {
    reinterpret_cast<bf16_t*>(p.d)[(size_t)m * p.ldd + n] = f2bf(v);
  }
}

// ------------------------------------------------------------------------------------------------
// MFMA kernel: 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32x16 tiles.
// The MFMA is issued with the B-tile fragment as its first operand (rows = n) and the A-tile fragment
// as its second (cols = m): accumulator reg r of lane l holds D[m = l&31][n = (r&3)+8*(r>>2)+4*(l>>5)],
// i.e. 4 consecutive n per register quad -> 8/16-byte epilogue accesses.
// ------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int KMAJ_BYTES = 128 * BK * 2;        // [128 rows][64 k] bf16, 128-byte rows, XOR swizzled
constexpr int MNMAJ_STRIDE = 128 * 2;           // [64 k rows][128 cols] bf16; the 64-byte block index is XORed with (k row & 3) so that the
constexpr int MNMAJ_BYTES = BK * MNMAJ_STRIDE;  // 4 rows x 64 B touched by one 32-lane tr-read group cover all 64 banks (no padding needed)
__device__ __forceinline__ int mn_off(int kr, int chunk16) { return kr * MNMAJ_STRIDE + ((chunk16 ^ ((kr & 3) << 2)) << 4); }


template <bool KMAJ>
struct TileIO {
  // global -> registers (4 x 16 B per thread) for the 128(rows of M or N) x 64(k) operand tile
  static __device__ __forceinline__ void load(uint4 (&r)[4], const bf16_t* base, int ld, int row0, int nrows, int k0, int kdim, int tid) {
    if (KMAJ) {
      const int chunk = tid & 7;
      const int kk = k0 + chunk * 8;
#pragma unroll
      for (int pss = 0; pss < 4; pss++) {
        int row = row0 + pss * 32 + (tid >> 3);
        row = row < nrows ? row : nrows - 1;
        if (kk < kdim) r[pss] = *reinterpret_cast<const uint4*>(base + (size_t)row * ld + kk);
        else r[pss] = make_uint4(0, 0, 0, 0);
      }
    } else {
      const int chunk = tid & 15;
      int col = row0 + chunk * 8;
      col = col < nrows ? col : 0;
#pragma unroll
      for (int pss = 0; pss < 4; pss++) {
        const int kr = k0 + pss * 16 + (tid >> 4);
        if (kr < kdim) r[pss] = *reinterpret_cast<const uint4*>(base + (size_t)kr * ld + col);
        else r[pss] = make_uint4(0, 0, 0, 0);
      }
    }
  }
  // registers -> LDS
  static __device__ __forceinline__ void store(const uint4 (&r)[4], char* lds, int tid) {
    if (KMAJ) {
      const int chunk = tid & 7;
#pragma unroll
      for (int pss = 0; pss < 4; pss++) {
        const int row = pss * 32 + (tid >> 3);
        *reinterpret_cast<uint4*>(lds + swz_off<128>(row, chunk)) = r[pss];
      }
    } else {
      const int chunk = tid & 15;
#pragma unroll
      for (int pss = 0; pss < 4; pss++) {
        const int kr = pss * 16 + (tid >> 4);
        *reinterpret_cast<uint4*>(lds + mn_off(kr, chunk)) = r[pss];
      }
    }
  }
  // MFMA operand fragment: 32 rows starting at `base`, 16 k starting at ks*16; lane l -> row base+(l&31), k-group l>>5
  static __device__ __forceinline__ short8v frag(const char* lds, int base, int ks, int lane) {
    if (KMAJ) {
      const int row = base + (lane & 31);
      return *reinterpret_cast<const short8v*>(lds + swz_off<128>(row, ks * 2 + (lane >> 5)));
    } else {
      const int q4 = lane >> 4, t = lane & 15;
      const int col = base + 16 * (q4 & 1) + 4 * (t & 3);
      const int kr = ks * 16 + 8 * (q4 >> 1) + (t >> 2);
      const short4v lo = lds_tr16_b64(lds + mn_off(kr, col >> 3) + (col & 7) * 2);
      const short4v hi = lds_tr16_b64(lds + mn_off(kr + 4, col >> 3) + (col & 7) * 2);
      short8v out;
      out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; out[3] = lo[3];
      out[4] = hi[0]; out[5] = hi[1]; out[6] = hi[2]; out[7] = hi[3];
      return out;
    }
  }
  // global -> LDS directly (LDS-DMA, 16 B per lane, no VGPR staging, no ds_write).  The destination of one wave-instruction
  // is lane-linear (1 KiB = 8 k-major rows or 4 mn-major rows); the XOR swizzles only permute 16-byte chunks inside a
  // row, so they are applied to the per-lane SOURCE address and every row segment is still fetched as whole cache lines.
  // Out-of-range reduction indices read a zero page (M/N tails just re-read a valid row; their outputs are never stored).
  static __device__ __forceinline__ void glds(char* lds, const bf16_t* base, int ld, int row0, int nrows, int k0, int kdim, int lane, int wave,
                                              const bf16_t* zero_page) {
#pragma unroll
    for (int pss = 0; pss < 4; pss++) {
      const int blk = pss * 4 + wave;  // 1 KiB block of the 16 KiB tile
      const bf16_t* src;
      if (KMAJ) {
        const int row = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int rg = row0 + row;
        rg = rg < nrows ? rg : nrows - 1;
        const int kk = k0 + c * 8;
        src = kk < kdim ? base + (size_t)rg * ld + kk : zero_page;
      } else {
        const int kr = blk * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((kr & 3) << 2);
        int col = row0 + c * 8;
        col = col < nrows ? col : 0;
        src = (k0 + kr) < kdim ? base + (size_t)(k0 + kr) * ld + col : zero_page;
      }
      glds16(__builtin_amdgcn_readfirstlane(lds_address(lds) + blk * 1024), src);
    }
  }
  // The same with the per-lane source pointers of k-tile 0 computed once (src4) and advanced by a k offset per tile (glds_at): the
  // per-tile address arithmetic with its range tests was ~100 instructions per wave in front of every tile's first MFMA.  Only for
  // k-tiles that lie entirely inside the reduction range (the caller uses glds() for a ragged last tile).
  struct Src4 { const bf16_t* p[4]; };
  static __device__ __forceinline__ Src4 src4(const bf16_t* base, int ld, int row0, int nrows, int lane, int wave) {
    Src4 r;
#pragma unroll
    for (int pss = 0; pss < 4; pss++) {
      const int blk = pss * 4 + wave;
      if (KMAJ) {
        const int row = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int rg = row0 + row;
        rg = rg < nrows ? rg : nrows - 1;
        r.p[pss] = base + (size_t)rg * ld + c * 8;
      } else {
        const int kr = blk * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((kr & 3) << 2);
        int col = row0 + c * 8;
        col = col < nrows ? col : 0;
        r.p[pss] = base + (size_t)kr * ld + col;
      }
    }
    return r;
  }
  static __device__ __forceinline__ size_t k_step(int ld) { return KMAJ ? (size_t)BK : (size_t)BK * ld; }  // elements per k-tile
  static __device__ __forceinline__ void glds_at(uint32_t lds_addr, const Src4& s, size_t koff, int wave) {
    const uint32_t a = lds_addr + wave * 1024;
    glds16x4(a, a + 4096, a + 8192, a + 12288, s.p[0] + koff, s.p[1] + koff, s.p[2] + koff, s.p[3] + koff);
  }
  static constexpr int BYTES = KMAJ ? KMAJ_BYTES : MNMAJ_BYTES;
};

__device__ __attribute__((aligned(16))) uint32_t g_zero_page[4] = {0u, 0u, 0u, 0u};


// XCD-aware work order.  Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with a private 4 MiB L2, so
// neighbouring ids never share a cache: with the plain row-major map the 6 column tiles of one A row block were fetched
// by 6 different L2s (PMC: fabric reads 3-4x the operand bytes).  xcd_remap() hands XCD x the CONTIGUOUS logical range
// [x*n/8, (x+1)*n/8); tile_of() then walks groups of 8 row tiles with m fastest, so the ~64 tiles resident on one XCD
// form an 8x8 patch (8 A blocks + 8 B blocks feed 64 tiles).  Placement is a speed hint only, never a correctness one.
__device__ __forceinline__ void tile_of(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
  constexpr int G = 8;
  const int per_group = G * tiles_n;
  const int g = t / per_group, first_m = g * G;
  const int gsz = min(G, tiles_m - first_m);
  const int in_g = t - g * per_group;
  tn = in_g / gsz;
  tm = first_m + in_g - tn * gsz;
}

// ---- BK = 32 variant of the operand tiles (8 KiB each): for the short-reduction GEMMs, where a 32 KiB workgroup (2 stages) lets THREE
// workgroups share a CU: finer rounds over the tile count (768 slots) and one more workgroup to cover a neighbour's prologue / epilogue.
constexpr int BK32 = 32;
template <bool KMAJ>
struct TileIO32 {
  static constexpr int BYTES = 8192;  // k-major: [128 rows][64 B]; mn-major: [32 k rows][256 B]
  static __device__ __forceinline__ short8v frag(const char* lds, int base, int ks, int lane) {
    if (KMAJ) {
      const int row = base + (lane & 31);
      return *reinterpret_cast<const short8v*>(lds + swz_off<64>(row, ks * 2 + (lane >> 5)));
    } else {
      const int q4 = lane >> 4, t = lane & 15;
      const int col = base + 16 * (q4 & 1) + 4 * (t & 3);
      const int kr = ks * 16 + 8 * (q4 >> 1) + (t >> 2);
      const short4v lo = lds_tr16_b64(lds + mn_off(kr, col >> 3) + (col & 7) * 2);
      const short4v hi = lds_tr16_b64(lds + mn_off(kr + 4, col >> 3) + (col & 7) * 2);
      short8v out;
      out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; out[3] = lo[3];
      out[4] = hi[0]; out[5] = hi[1]; out[6] = hi[2]; out[7] = hi[3];
      return out;
    }
  }
  struct Src2 { const bf16_t* p[2]; };  // per-lane DMA sources of k-tile 0: two 1 KiB pieces per wave and tile
  static __device__ __forceinline__ Src2 src2(const bf16_t* base, int ld, int row0, int nrows, int lane, int wave) {
    Src2 r;
#pragma unroll
    for (int pss = 0; pss < 2; pss++) {
      const int blk = pss * 4 + wave;
      if (KMAJ) {
        const int row = blk * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        int rg = row0 + row;
        rg = rg < nrows ? rg : nrows - 1;
        r.p[pss] = base + (size_t)rg * ld + c * 8;
      } else {
        const int kr = blk * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((kr & 3) << 2);
        int col = row0 + c * 8;
        col = col < nrows ? col : 0;
        r.p[pss] = base + (size_t)kr * ld + col;
      }
    }
    return r;
  }
  static __device__ __forceinline__ size_t k_step(int ld) { return KMAJ ? (size_t)BK32 : (size_t)BK32 * ld; }
};

// fused epilogue of a 32x64 half of a wave tile through 8 KiB of LDS (the BK = 32 kernel has 32 KiB in all): same row-contiguous scheme
// as tile_epilogue_rows, 32 rows at a time
template <int W>
__device__ __forceinline__ void half_epilogue_rows(const GemmP& p, const float16v (&acc)[2], int mw, int nw, int lane, bool add_bias, float* stg,
                                                   float* ws_base, long long ws_ld) {
  constexpr int LPR = 64 / W, RPP = 64 / LPR, PASSES = 32 / RPP;
  const int ml = lane & 31, hi = lane >> 5;
  const int rl = lane / LPR, cl = (lane % LPR) * W;
  const int n = nw + cl;
  float bv[W];
  float q8max = 0.f;
  const float q8inv = (!ws_base && p.out8) ? *p.out8_inv : 0.f;  // in flight during the LDS staging
  constexpr int NSTRIP = (PASSES * RPP) / 32;  // 32-row strips of this call's rows (2 for a 64-row wave tile, 1 for a half)
  float csum[NSTRIP][W];
#pragma unroll
  for (int h = 0; h < NSTRIP; h++)
#pragma unroll
    for (int i = 0; i < W; i++) csum[h][i] = 0.f;
  EpiPre<W> pre[PASSES];
  if (!ws_base) {
    epi_load_bias<W>(p, n, add_bias, bv);
#pragma unroll
    for (int pss = 0; pss < PASSES; pss++) {
      const int m = mw + pss * RPP + rl;
      if (m < p.m && n < p.n) epi_load<W>(p, m, n, pre[pss]);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int q = 0; q < 4; q++)
      *reinterpret_cast<float4*>(stg + ml * 64 + (((j * 8 + 2 * q + hi) ^ (ml & 15)) << 2)) =
          make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
#pragma unroll
  for (int pss = 0; pss < PASSES; pss++) {
    const int r = pss * RPP + rl;
    const int m = mw + r;
    float v[W];
#pragma unroll
    for (int c = 0; c < W / 4; c++) {
      const float4 t = *reinterpret_cast<const float4*>(stg + r * 64 + ((((cl >> 2) + c) ^ (r & 15)) << 2));
      v[4 * c] = t.x; v[4 * c + 1] = t.y; v[4 * c + 2] = t.z; v[4 * c + 3] = t.w;
    }
    if (m < p.m && n < p.n) {
      if (ws_base) {
#pragma unroll
        for (int c = 0; c < W; c += 4) *reinterpret_cast<float4*>(ws_base + (long long)m * ws_ld + n + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
      } else {
        epi_apply<W>(p, m, n, v, bv, pre[pss], q8max, q8inv);
        if (W == 8 && p.colsum_partials) {
#pragma unroll
          for (int i = 0; i < W; i++) csum[(pss * RPP) / 32][i] += v[i];
        }
      }
    }
  }
  if (!ws_base && p.out8_amax) q8_amax_commit(p.out8_amax, q8max, (int)blockIdx.x + (int)(threadIdx.x >> 6));
  if (W == 8 && !ws_base && p.colsum_partials) {  // lanes with the same column group (lane % LPR) hold different rows: butterfly over the row lanes, one store per column group
#pragma unroll
    for (int h = 0; h < NSTRIP; h++) {
#pragma unroll
      for (int i = 0; i < W; i++) {
        float t = csum[h][i];
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
        csum[h][i] = t;
      }
      const int strip = (mw >> 5) + h;
      if (rl == 0 && n < p.n && strip * 32 < p.m) {
        float* dst = p.colsum_partials + (size_t)strip * p.n + n;
        *reinterpret_cast<float4*>(dst) = make_float4(csum[h][0], csum[h][1], csum[h][2], csum[h][3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(csum[h][W - 4], csum[h][W - 3], csum[h][W - 2], csum[h][W - 1]);
      }
    }
  }
}

template <int EPI>
__device__ __forceinline__ void half_epilogue(const GemmP& p, const float16v (&acc)[2], int mw, int nw, int lane, int z, float* stg) {
  static_assert(EPI != EPI_GENERAL, "the BK = 32 kernel is instantiated for the specialised epilogue classes only");
  float* ws_base = nullptr;
  long long ws_ld = 0;
  if (p.ws) { ws_base = p.ws + (size_t)z * p.m * p.n; ws_ld = p.n; }
  if (ws_base) { half_epilogue_rows<4>(p, acc, mw, nw, lane, false, stg, ws_base, ws_ld); return; }
  const GemmP q = epi_fold<EPI>(p);
  if (EPI == EPI_F32) {
    half_epilogue_rows<4>(q, acc, mw, nw, lane, z == 0, stg, nullptr, 0);
  } else {
    half_epilogue_rows<8>(q, acc, mw, nw, lane, z == 0, stg, nullptr, 0);
  }
}

}  // namespace
