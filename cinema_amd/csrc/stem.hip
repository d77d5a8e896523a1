// Fused per-voxel halves of the conv stem's MaskedConvBlock (reference cinema/conv.py:349-415) on the visible-voxel compact rows of an MAE step.
//
//   x1 = x  + conv2(dw5(conv1(LN1(x))))          x2 = x1 + fc2(GELU(fc1(LN2(x1))))
//
// Everything except the depthwise 5^n conv is per voxel.  The unfused path ran LN / GEMM / GEMM / LN / GEMM / GEMM as separate launches whose K = 64 / 128
// GEMMs are bound by what they write: the 4C-wide hidden activation (and GELU') of the MLP went out to HBM and came back (75 MB per tensor and block at
// stage 1).  Here a wave owns 16 voxels and keeps them in registers through a whole chain of projections:
//
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand (read from LDS) and the ACTIVATIONS as the B operand: D[out channel][voxel], lane = voxel
//     (l & 15), the four lane groups g = l >> 4 hold out channels 16 rb + 4 g + {0..3} of every 16-channel block rb.  Two such blocks ARE the B operand of
//     the next projection (k = 32 s + 4 g + i and 32 s + 16 + 4 g + i: a permutation of the 32 reduction indices that the weight fragment simply mirrors:
//     two 8-byte LDS reads at those columns) - no cross-lane move, no LDS round trip between the GEMMs of a chain, no barrier inside a tile.
//   * transposed projections (data gradients) read the same row-major weight tiles with ds_read_b64_tr_b16.
//   * LayerNorm statistics: the lane sums its C / 4 channels, two xor-shuffles add the four lane groups.
//   * the MLP's hidden layer is walked in chunks of 64 units (weights of a chunk double-buffered in LDS): fc1 -> GELU -> fc2 accumulate per chunk, so the
//     4C-wide activation never exists outside registers in the forward pass.
//
// Kernels: stem_ln_linear (LN1 -> conv1), stem_mlp_fwd (conv2 + residual -> LN2 -> fc1 -> GELU -> fc2 + residual), stem_mlp_bwd (the backward of that half,
// recomputing the hidden layer from the saved x1), stem_ln_linear_bwd (conv1 data gradient -> LN1 backward + residual gradient), stem_wgrad (all weight
// gradients of a block from row-major operand pairs: every workgroup owns a full small dW for a slice of rows, deterministic slab reduce).
#include "common.cuh"
#include "../../include/cinema_hip.h"
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) uint32_t u4;
constexpr int NW = 8;          // waves per workgroup
constexpr int NT = NW * 64;
constexpr int HC = 64;         // hidden units per chunk

__device__ __forceinline__ f4 mfma16(short8v a, short8v b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ short8v mk8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  const u4 u = {a, b, c, d};
  return __builtin_bit_cast(short8v, u);
}

// ---- LDS weight tiles -----------------------------------------------------------------------------------------------------------------------------
// row-major bf16 [rows][cols], row pitch in bytes: 2 cols + 16 for tiles read as N fragments (16 rows x 2 lane groups of one ds_read_b64 half-wave cover the
// 64 banks once), 2 cols + 32 for tiles read only through the transpose read (8 rows x 32 B).
template <int ROWS, int COLS>
struct TileRegs {  // one tile on its way global -> registers -> LDS (the next chunk's weights are fetched while the current chunk is computed)
  static constexpr int CPR = COLS / 8, N = ROWS * CPR / NT;
  static_assert(ROWS * CPR % NT == 0 && N >= 1, "a tile is a whole number of 16-byte pieces per thread (a guarded piece ends up in scratch)");
  u4 v[N];
  __device__ __forceinline__ void load(const bf16_t* src, int ld, int tid) {
#pragma unroll
    for (int q = 0; q < N; q++) {
      const int i = tid + q * NT;
      v[q] = *reinterpret_cast<const u4*>(src + (size_t)(i / CPR) * ld + (i % CPR) * 8);
    }
  }
  __device__ __forceinline__ void store(char* lds, int pitch, int tid) const {
#pragma unroll
    for (int q = 0; q < N; q++) {
      const int i = tid + q * NT;
      *reinterpret_cast<u4*>(lds + (i / CPR) * pitch + (i % CPR) * 16) = v[q];
    }
  }
};
template <int ROWS, int COLS>
__device__ __forceinline__ void stage_tile(char* lds, int pitch, const bf16_t* src, int ld, int tid) {
  TileRegs<ROWS, COLS> t;
  t.load(src, ld, tid);
  t.store(lds, pitch, tid);
}
template <int ROWS, int COLS, int NTH>  // one-off staging by any number of threads
__device__ __forceinline__ void stage_tile_n(char* lds, int pitch, const bf16_t* src, int ld, int tid) {
  constexpr int CPR = COLS / 8;
  for (int i = tid; i < ROWS * CPR; i += NTH) *reinterpret_cast<u4*>(lds + (i / CPR) * pitch + (i % CPR) * 16) = *reinterpret_cast<const u4*>(src + (size_t)(i / CPR) * ld + (i % CPR) * 8);
}
// A fragment (weights) for out rows m0 .. m0+15 and reduction indices k0 .. k0+31 in the CHAINED order: lane (m = l & 15, g = l >> 4) holds
// W[m0 + m][k0 + 4g + {0..3}] and W[m0 + m][k0 + 16 + 4g + {0..3}]
__device__ __forceinline__ short8v wfrag_n(const char* lds, int pitch, int m0, int k0, int lane) {
  const char* p = lds + (m0 + (lane & 15)) * pitch + (k0 + 4 * (lane >> 4)) * 2;
  // (hipcc fuses the two reads into one ds_read2_b64: half rate and banks mod 32, 2-way conflicts at this pitch - 42 % of the LDS cycles; keeping them apart through an
  // opaque pointer removed the conflicts and made the kernels SLOWER (56 -> 64 us): they are bound by instruction issue and waits, not by the LDS array)
  const uint2 lo = *reinterpret_cast<const uint2*>(p), hi = *reinterpret_cast<const uint2*>(p + 32);
  return mk8(lo.x, lo.y, hi.x, hi.y);
}
// the same fragment of the TRANSPOSED matrix: the tile holds Wt[k][m] (rows = reduction index), lane (m, g) receives Wt[k0 + 4g + j][m0 + m] and
// Wt[k0 + 16 + 4g + j][m0 + m], j = 0..3 (ds_read_b64_tr_b16: a 16-lane group reads a 4 x 16 block, lane t gets column t; tools/probe/probe.hip)
__device__ __forceinline__ short8v wfrag_t(const char* lds, int pitch, int k0, int m0, int lane) {
  const int t = lane & 15, g = lane >> 4;
  const char* p = lds + (k0 + 4 * g + (t >> 2)) * pitch + (m0 + 4 * (t & 3)) * 2;
  const short4v lo = lds_tr16_b64(p), hi = lds_tr16_b64(p + 16 * pitch);
  short8v o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3]; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}
__device__ __forceinline__ void gelu_v4(f4& z) {  // z <- GELU(z)
  float a = z[0], b = z[1], c = z[2], d = z[3];
  gelu2(a, b); gelu2(c, d);
  z = f4{a, b, c, d};
}
__device__ __forceinline__ f4 gelu_both_v4(f4& z) {  // z <- GELU(z), returns GELU'(z)
  float a = z[0], b = z[1], c = z[2], d = z[3], ga, gb, gc, gd;
  gelu_both2(a, b, ga, gb); gelu_both2(c, d, gc, gd);
  z = f4{a, b, c, d};
  return f4{ga, gb, gc, gd};
}
__device__ __forceinline__ f4 ldsv4(const float* vec, int c0) { return *reinterpret_cast<const f4*>(vec + c0); }

// ---- activations in the D layout: a[rb][i] = channel 16 rb + 4 g + i of voxel (l & 15) -----------------------------------------------------------------
template <int C>
__device__ __forceinline__ void load_act(f4 (&a)[C / 16], const float* base, int row, int g) {
#pragma unroll
  for (int rb = 0; rb < C / 16; rb++) a[rb] = *reinterpret_cast<const f4*>(base + (size_t)row * C + rb * 16 + g * 4);
}
template <int C>
__device__ __forceinline__ void store_act(const f4 (&a)[C / 16], float* base, int row, int g) {
#pragma unroll
  for (int rb = 0; rb < C / 16; rb++) *reinterpret_cast<f4*>(base + (size_t)row * C + rb * 16 + g * 4) = a[rb];
}
// bf16 rows -> B operands in the chained order (two 8-byte loads per k-step)
template <int C>
__device__ __forceinline__ void load_b16(short8v (&b)[C / 32], const bf16_t* base, int row, int g) {
#pragma unroll
  for (int s = 0; s < C / 32; s++) {
    const bf16_t* p = base + (size_t)row * C + s * 32 + g * 4;
    const uint2 lo = *reinterpret_cast<const uint2*>(p), hi = *reinterpret_cast<const uint2*>(p + 16);
    b[s] = mk8(lo.x, lo.y, hi.x, hi.y);
  }
}
__device__ __forceinline__ uint2 pack4(const f4& v) { return make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])); }
// D-layout activations -> B operands (and optionally the bf16 rows in memory, W channels per row)
template <int C>
__device__ __forceinline__ void pack_act(short8v (&b)[C / 32], const f4 (&a)[C / 16], bf16_t* out, int row, int g, bool ok) {
#pragma unroll
  for (int s = 0; s < C / 32; s++) {
    const uint2 lo = pack4(a[2 * s]), hi = pack4(a[2 * s + 1]);
    b[s] = mk8(lo.x, lo.y, hi.x, hi.y);
    if (out && ok) {
      *reinterpret_cast<uint2*>(out + (size_t)row * C + s * 32 + g * 4) = lo;
      *reinterpret_cast<uint2*>(out + (size_t)row * C + s * 32 + 16 + g * 4) = hi;
    }
  }
}
__device__ __forceinline__ float group4_sum(float v) {  // over the four lane groups that share a voxel
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// LayerNorm statistics of the voxel's C channels (two passes over the registers, like ln_fwd_kernel); a <- (a - mean) * rstd
template <int C>
__device__ __forceinline__ float normalise(f4 (&a)[C / 16], float eps) {
  float s = 0.f;
#pragma unroll
  for (int rb = 0; rb < C / 16; rb++) s += (a[rb][0] + a[rb][1]) + (a[rb][2] + a[rb][3]);
  const float mu = group4_sum(s) * (1.f / C);
  float q = 0.f;
#pragma unroll
  for (int rb = 0; rb < C / 16; rb++) {
    a[rb] -= mu;
    q += (a[rb][0] * a[rb][0] + a[rb][1] * a[rb][1]) + (a[rb][2] * a[rb][2] + a[rb][3] * a[rb][3]);
  }
  const float rs = rsqrtf(group4_sum(q) * (1.f / C) + eps);
#pragma unroll
  for (int rb = 0; rb < C / 16; rb++) a[rb] *= rs;
  return rs;
}
// per-lane LayerNorm parameter-gradient sums of a workgroup -> ONE partial row [d gamma (C) | d beta (C)] (the layout of cinema_ln_reduce_item)
template <int C>
__device__ __forceinline__ void emit_ln_partials(f4 (&dg)[C / 16], f4 (&db)[C / 16], float* red /* LDS [NW][2 C] */, float* partials, int tid) {
  const int lane = tid & 63, wave = tid >> 6, g = lane >> 4;
#pragma unroll
  for (int rb = 0; rb < C / 16; rb++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float a = dg[rb][i], b = db[rb][i];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      dg[rb][i] = a; db[rb][i] = b;
    }
  if ((lane & 15) == 0) {
#pragma unroll
    for (int rb = 0; rb < C / 16; rb++) {
      *reinterpret_cast<f4*>(red + wave * 2 * C + rb * 16 + g * 4) = dg[rb];
      *reinterpret_cast<f4*>(red + wave * 2 * C + C + rb * 16 + g * 4) = db[rb];
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += NT) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) s += red[w * 2 * C + i];
    partials[(size_t)blockIdx.x * 2 * C + i] = s;
  }
}

// =====================================================================================================================================================
// LN1 -> conv1:   xn = LN(x) (bf16, kept for conv1's weight gradient),  h = xn W^T + b (bf16: the depthwise conv's input)
// =====================================================================================================================================================
struct LnLinP {
  const float* x; const float* gamma; const float* beta; const bf16_t* w; const float* bias;
  bf16_t* xn; bf16_t* h;
  int rows; float eps;
};
template <int C>
__device__ __forceinline__ void ln_linear_body(const LnLinP& p) {
  constexpr int RB = C / 16, KS = C / 32, PITCH = 2 * C + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;
  float* vec = reinterpret_cast<float*>(smem + C * PITCH);  // gamma | beta | bias
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, v = lane & 15, g = lane >> 4;
  stage_tile<C, C>(wl, PITCH, p.w, C, tid);
  for (int i = tid; i < C; i += NT) { vec[i] = p.gamma[i]; vec[C + i] = p.beta[i]; vec[2 * C + i] = p.bias ? p.bias[i] : 0.f; }
  __syncthreads();
  const int n_tiles = (p.rows + 15) >> 4;
  for (int tile = wave * gridDim.x + blockIdx.x; tile < n_tiles; tile += gridDim.x * NW) {  // consecutive tiles on different workgroups: a ragged last round is spread over the chip
    const int row = tile * 16 + v;
    const bool ok = row < p.rows;
    const int rowc = ok ? row : p.rows - 1;
    f4 a[RB];
    load_act<C>(a, p.x, rowc, g);
    normalise<C>(a, p.eps);
#pragma unroll
    for (int rb = 0; rb < RB; rb++) a[rb] = a[rb] * ldsv4(vec, rb * 16 + g * 4) + ldsv4(vec + C, rb * 16 + g * 4);
    short8v bx[KS];
    pack_act<C>(bx, a, p.xn, row, g, ok);
#pragma unroll
    for (int ob = 0; ob < RB; ob++) {
      f4 acc = ldsv4(vec + 2 * C, ob * 16 + g * 4);
#pragma unroll
      for (int s = 0; s < KS; s++) acc = mfma16(wfrag_n(wl, PITCH, ob * 16, s * 32, lane), bx[s], acc);
      if (ok) *reinterpret_cast<uint2*>(p.h + (size_t)row * C + ob * 16 + g * 4) = pack4(acc);
      __builtin_amdgcn_sched_barrier(0);  // (keeps the live weight fragments to one output block: the unrolled loop otherwise hoists all of them)
    }
  }
}
template <int C> __global__ __launch_bounds__(NT) void stem_ln_linear_kernel(LnLinP p) { ln_linear_body<C>(p); }
template <int C> __global__ __launch_bounds__(NT) void stem_ln_linear_lanes_kernel(Lanes<LnLinP> L) { ln_linear_body<C>(L.p[blockIdx.y]); }

// =====================================================================================================================================================
// conv1 data gradient -> LN1 backward:   dx = dres + LN'(x; dh W),  partial d gamma / d beta per workgroup
// =====================================================================================================================================================
struct LnLinBwdP {
  const bf16_t* dh; const float* x; const float* dres; const float* gamma; const bf16_t* w;
  float* dx; float* partials;
  int rows; float eps;
};
template <int C>
__device__ __forceinline__ void ln_linear_bwd_body(const LnLinBwdP& p) {
  constexpr int RB = C / 16, KS = C / 32, PITCH = 2 * C + 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;                                          // W[o][i]: rows = reduction index o of the data gradient
  float* vec = reinterpret_cast<float*>(smem + C * PITCH);  // gamma
  float* red = vec + C;                                     // [NW][2 C]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, v = lane & 15, g = lane >> 4;
  stage_tile<C, C>(wl, PITCH, p.w, C, tid);
  for (int i = tid; i < C; i += NT) vec[i] = p.gamma[i];
  __syncthreads();
  f4 dgam[RB], dbet[RB];
#pragma unroll
  for (int rb = 0; rb < RB; rb++) { dgam[rb] = f4{0.f, 0.f, 0.f, 0.f}; dbet[rb] = f4{0.f, 0.f, 0.f, 0.f}; }
  const int n_tiles = (p.rows + 15) >> 4;
  for (int tile = wave * gridDim.x + blockIdx.x; tile < n_tiles; tile += gridDim.x * NW) {  // consecutive tiles on different workgroups: a ragged last round is spread over the chip
    const int row = tile * 16 + v;
    const bool ok = row < p.rows;
    const int rowc = ok ? row : p.rows - 1;
    f4 xh[RB];
    load_act<C>(xh, p.x, rowc, g);
    const float rs = normalise<C>(xh, p.eps);
    short8v bd[KS];
    load_b16<C>(bd, p.dh, rowc, g);
    float s1 = 0.f, s2 = 0.f;
    f4 gy[RB];
#pragma unroll
    for (int ob = 0; ob < RB; ob++) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; s++) acc = mfma16(wfrag_t(wl, PITCH, s * 32, ob * 16, lane), bd[s], acc);
      if (ok) { dgam[ob] += acc * xh[ob]; dbet[ob] += acc; }
      gy[ob] = acc * ldsv4(vec, ob * 16 + g * 4);
      s1 += (gy[ob][0] + gy[ob][1]) + (gy[ob][2] + gy[ob][3]);
      s2 += (gy[ob][0] * xh[ob][0] + gy[ob][1] * xh[ob][1]) + (gy[ob][2] * xh[ob][2] + gy[ob][3] * xh[ob][3]);
      __builtin_amdgcn_sched_barrier(0);
    }
    s1 = group4_sum(s1) * (1.f / C);
    s2 = group4_sum(s2) * (1.f / C);
    if (ok) {
#pragma unroll
      for (int rb = 0; rb < RB; rb++) {
        f4 dx = (gy[rb] - s1 - xh[rb] * s2) * rs;
        if (p.dres) dx += *reinterpret_cast<const f4*>(p.dres + (size_t)row * C + rb * 16 + g * 4);
        *reinterpret_cast<f4*>(p.dx + (size_t)row * C + rb * 16 + g * 4) = dx;
      }
    }
  }
  emit_ln_partials<C>(dgam, dbet, red, p.partials, tid);
}
template <int C> __global__ __launch_bounds__(NT) void stem_ln_linear_bwd_kernel(LnLinBwdP p) { ln_linear_bwd_body<C>(p); }
template <int C> __global__ __launch_bounds__(NT) void stem_ln_linear_bwd_lanes_kernel(Lanes<LnLinBwdP> L) { ln_linear_bwd_body<C>(L.p[blockIdx.y]); }

// =====================================================================================================================================================
// conv2 + residual -> LN2 -> fc1 -> GELU -> fc2 + residual
// =====================================================================================================================================================
struct MlpFwdP {
  const bf16_t* d; const float* x;
  const bf16_t* w2; const float* b2; const float* gamma; const float* beta;
  const bf16_t* wf1; const float* bf1; const bf16_t* wf2; const float* bf2;
  float* x1; float* x2;
  int rows; float eps;
};
// RES: every chunk of the MLP weights stays in LDS (c = 64: 83 KB) - no weight stream, no barrier in the loop, the waves of a workgroup drift apart so that one
// wave's row loads / stores hide behind the others' matrix work; otherwise (c = 128: 288 KB of weights) the chunks are double-buffered and the waves walk them in step.
template <int C, bool RES>
struct MlpFwdLds {
  static constexpr int P_C = 2 * C + 16, P_H = 2 * HC + 16, NBUF = RES ? 4 * C / HC : 2;
  static constexpr int W2 = 0, BUF = C * P_C, BUF_BYTES = HC * P_C + C * P_H, WF2_OFF = HC * P_C;
  static constexpr int VEC = BUF + NBUF * BUF_BYTES;      // b2 | gamma | beta | bf2 | bf1 (4 C)
  static constexpr int BYTES = VEC + 8 * C * 4;
};
template <int C, bool RES, int NWV>
__device__ __forceinline__ void mlp_fwd_body(const MlpFwdP& p) {
  static_assert(RES || NWV == NW, "the streamed form moves its tiles with NT threads");
  constexpr int NTH = NWV * 64;
  using L = MlpFwdLds<C, RES>;
  constexpr int RB = C / 16, KS = C / 32, H = 4 * C, NCH = H / HC, P_C = L::P_C, P_H = L::P_H;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* w2l = smem + L::W2;
  float* vec = reinterpret_cast<float*>(smem + L::VEC);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, v = lane & 15, g = lane >> 4;
  stage_tile_n<C, C, NTH>(w2l, P_C, p.w2, C, tid);
#pragma unroll
  for (int j = 0; j < (RES ? NCH : 1); j++) {
    stage_tile_n<HC, C, NTH>(smem + L::BUF + j * L::BUF_BYTES, P_C, p.wf1 + (size_t)j * HC * C, C, tid);
    stage_tile_n<C, HC, NTH>(smem + L::BUF + j * L::BUF_BYTES + L::WF2_OFF, P_H, p.wf2 + j * HC, H, tid);
  }
  for (int i = tid; i < C; i += NTH) { vec[i] = p.b2[i]; vec[C + i] = p.gamma[i]; vec[2 * C + i] = p.beta[i]; vec[3 * C + i] = p.bf2[i]; }
  for (int i = tid; i < H; i += NTH) vec[4 * C + i] = p.bf1[i];
  __syncthreads();
  const int n_tiles = (p.rows + 15) >> 4;
  const int passes = (n_tiles + gridDim.x * NWV - 1) / (gridDim.x * NWV);
  TileRegs<HC, C> na1, nb1;
  TileRegs<C, HC> na2, nb2;
  if constexpr (!RES) {
    nb1.load(p.wf1 + (size_t)HC * C, C, tid);   // chunk 1 on its way
    nb2.load(p.wf2 + HC, H, tid);
  }
  for (int pass = 0; pass < passes; pass++) {
    // consecutive tiles go to different workgroups, so a ragged last pass (config 2, c = 128: 2304 tiles on 256 x 8 waves) puts ONE more tile on every compute unit
    // instead of a full pass on 32 of them - the pass time follows the LDS traffic of the busiest unit
    const int tile = (pass * NWV + wave) * gridDim.x + blockIdx.x;
    const bool has = tile < n_tiles;
    if (RES && !has) break;  // (no barrier below: a wave without rows just leaves)
    const int row = tile * 16 + v;
    const bool ok = row < p.rows;
    const int rowc = ok ? row : p.rows - 1;
    f4 x1[RB];
    short8v bx[KS];
    {
      short8v bd[KS];
      load_b16<C>(bd, p.d, rowc, g);
      load_act<C>(x1, p.x, rowc, g);
#pragma unroll
      for (int ob = 0; ob < RB; ob++) {
        f4 acc = x1[ob] + ldsv4(vec, ob * 16 + g * 4);
#pragma unroll
        for (int s = 0; s < KS; s++) acc = mfma16(wfrag_n(w2l, P_C, ob * 16, s * 32, lane), bd[s], acc);
        x1[ob] = acc;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (ok && p.x1) store_act<C>(x1, p.x1, row, g);
    f4 y[RB];
    {
      f4 xn[RB];
#pragma unroll
      for (int rb = 0; rb < RB; rb++) { xn[rb] = x1[rb]; y[rb] = x1[rb] + ldsv4(vec + 3 * C, rb * 16 + g * 4); }
      normalise<C>(xn, p.eps);
#pragma unroll
      for (int rb = 0; rb < RB; rb++) xn[rb] = xn[rb] * ldsv4(vec + C, rb * 16 + g * 4) + ldsv4(vec + 2 * C, rb * 16 + g * 4);
      pack_act<C>(bx, xn, nullptr, row, g, false);
    }
    auto chunk = [&](int j, const char* wf1c) {
      const char* wf2c = wf1c + L::WF2_OFF;
      const float* b1 = vec + 4 * C + j * HC;
#pragma unroll
      for (int t = 0; t < HC / 32; t++) {
        f4 z0 = ldsv4(b1, t * 32 + g * 4), z1 = ldsv4(b1, t * 32 + 16 + g * 4);
#pragma unroll
        for (int s = 0; s < KS; s++) {
          z0 = mfma16(wfrag_n(wf1c, P_C, t * 32, s * 32, lane), bx[s], z0);
          z1 = mfma16(wfrag_n(wf1c, P_C, t * 32 + 16, s * 32, lane), bx[s], z1);
        }
        gelu_v4(z0); gelu_v4(z1);
        const uint2 lo = pack4(z0), hi = pack4(z1);
        const short8v ba = mk8(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
        for (int ob = 0; ob < RB; ob++) y[ob] = mfma16(wfrag_n(wf2c, P_H, ob * 16, t * 32, lane), ba, y[ob]);
      }
    };
    if constexpr (RES) {
      for (int j = 0; j < NCH; j++) chunk(j, smem + L::BUF + j * L::BUF_BYTES);
    } else {
      // the chunk's weights are fetched TWO chunks ahead (registers: sets A / B) and written to the other LDS buffer one chunk ahead: one chunk of matrix work
      // (~1 us) did not cover the fetch (8 x ~2.7 us of waiting per pass at c = 128)
      for (int j = 0; j < NCH; j += 2) {
        na1.load(p.wf1 + (size_t)((j + 2) % NCH) * HC * C, C, tid);
        na2.load(p.wf2 + ((j + 2) % NCH) * HC, H, tid);
        if (has) chunk(j, smem + L::BUF);   // (a wave without a tile still moves the weights and meets the barriers)
        nb1.store(smem + L::BUF + L::BUF_BYTES, P_C, tid);          // chunk j + 1 (fetched during chunk j - 1)
        nb2.store(smem + L::BUF + L::BUF_BYTES + L::WF2_OFF, P_H, tid);
        __syncthreads();
        nb1.load(p.wf1 + (size_t)((j + 3) % NCH) * HC * C, C, tid);
        nb2.load(p.wf2 + ((j + 3) % NCH) * HC, H, tid);
        if (has) chunk(j + 1, smem + L::BUF + L::BUF_BYTES);
        na1.store(smem + L::BUF, P_C, tid);                          // chunk j + 2
        na2.store(smem + L::BUF + L::WF2_OFF, P_H, tid);
        __syncthreads();
      }
    }
    if (ok) store_act<C>(y, p.x2, row, g);
  }
}
constexpr int fwd_waves(int c) { return c == 64 ? 16 : NW; }  // c = 64: resident weights, 16 free-running waves per CU
template <int C> __global__ __launch_bounds__(fwd_waves(C) * 64) void stem_mlp_fwd_kernel(MlpFwdP p) { mlp_fwd_body<C, C == 64, fwd_waves(C)>(p); }
template <int C> __global__ __launch_bounds__(fwd_waves(C) * 64) void stem_mlp_fwd_lanes_kernel(Lanes<MlpFwdP> L) { mlp_fwd_body<C, C == 64, fwd_waves(C)>(L.p[blockIdx.y]); }

// =====================================================================================================================================================
// backward of that half.  In: g2 = dL/dx2 (fp32), x1 (saved).  Out: dx1 (fp32 + bf16), dd = dL/d(dw output) (bf16), and the operands of the weight
// gradients: a = GELU(fc1), dz = dL/d(fc1 output) (bf16, 4 C wide), xn2 = LN2(x1), g2 in bf16; partial d gamma / d beta of LN2.
// =====================================================================================================================================================
struct MlpBwdP {
  const float* g2; const float* x1;
  const bf16_t* w2; const float* gamma; const float* beta; const bf16_t* wf1; const float* bf1; const bf16_t* wf2;
  float* dx1; bf16_t* dx1_16; bf16_t* dd; bf16_t* a; bf16_t* dz; bf16_t* xn2; bf16_t* g2_16; float* partials;
  int rows; float eps;
};
template <int C>
struct MlpBwdLds {
  static constexpr int P_C = 2 * C + 16, PT_C = 2 * C + 32, PT_H = 2 * HC + 32;
  static constexpr int W2 = 0, BUF = C * PT_C, BUF_BYTES = HC * P_C + C * PT_H, WF2_OFF = HC * P_C;
  static constexpr int VEC = BUF + 2 * BUF_BYTES;      // gamma | beta | bf1 (4 C)
  static constexpr int BYTES = VEC + 6 * C * 4;        // (the partial-sum staging [NW][2 C] reuses the chunk buffers)
};
template <int C>
__device__ __forceinline__ void mlp_bwd_body(const MlpBwdP& p) {
  using L = MlpBwdLds<C>;
  constexpr int RB = C / 16, KS = C / 32, H = 4 * C, NCH = H / HC, P_C = L::P_C, PT_C = L::PT_C, PT_H = L::PT_H;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* w2l = smem + L::W2;
  float* vec = reinterpret_cast<float*>(smem + L::VEC);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, v = lane & 15, g = lane >> 4;
  stage_tile<C, C>(w2l, PT_C, p.w2, C, tid);
  stage_tile<HC, C>(smem + L::BUF, P_C, p.wf1, C, tid);
  stage_tile<C, HC>(smem + L::BUF + L::WF2_OFF, PT_H, p.wf2, H, tid);
  for (int i = tid; i < C; i += NT) { vec[i] = p.gamma[i]; vec[C + i] = p.beta[i]; }
  for (int i = tid; i < H; i += NT) vec[2 * C + i] = p.bf1[i];
  __syncthreads();
  f4 dgam[RB], dbet[RB];
#pragma unroll
  for (int rb = 0; rb < RB; rb++) { dgam[rb] = f4{0.f, 0.f, 0.f, 0.f}; dbet[rb] = f4{0.f, 0.f, 0.f, 0.f}; }
  const int n_tiles = (p.rows + 15) >> 4;
  const int passes = (n_tiles + gridDim.x * NW - 1) / (gridDim.x * NW);
  for (int pass = 0; pass < passes; pass++) {
    const int tile = (pass * NW + wave) * gridDim.x + blockIdx.x;   // (as in the forward kernel: a ragged last pass is spread over all workgroups)
    const bool has = tile < n_tiles;
    const int row = tile * 16 + v;
    const bool ok = row < p.rows;
    const int rowc = ok ? row : p.rows - 1;
    short8v bx[KS], bg[KS];
    {
      f4 t[RB];
      load_act<C>(t, p.x1, rowc, g);
      normalise<C>(t, p.eps);
#pragma unroll
      for (int rb = 0; rb < RB; rb++) t[rb] = t[rb] * ldsv4(vec, rb * 16 + g * 4) + ldsv4(vec + C, rb * 16 + g * 4);
      pack_act<C>(bx, t, p.xn2, row, g, ok);
      load_act<C>(t, p.g2, rowc, g);
      pack_act<C>(bg, t, p.g2_16, row, g, ok);
    }
    f4 dxn[RB];
#pragma unroll
    for (int rb = 0; rb < RB; rb++) dxn[rb] = f4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < NCH; j++) {
      const int jn = (j + 1) % NCH;
      TileRegs<HC, C> n1;
      TileRegs<C, HC> n2;
      n1.load(p.wf1 + (size_t)jn * HC * C, C, tid);
      n2.load(p.wf2 + jn * HC, H, tid);
      const char* wf1c = smem + L::BUF + (j & 1) * L::BUF_BYTES;
      const char* wf2c = wf1c + L::WF2_OFF;
      const float* b1 = vec + 2 * C + j * HC;
      if (has) {
#pragma unroll
      for (int t = 0; t < HC / 32; t++) {
        f4 z0 = ldsv4(b1, t * 32 + g * 4), z1 = ldsv4(b1, t * 32 + 16 + g * 4);
        f4 da0 = {0.f, 0.f, 0.f, 0.f}, da1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; s++) {
          z0 = mfma16(wfrag_n(wf1c, P_C, t * 32, s * 32, lane), bx[s], z0);
          z1 = mfma16(wfrag_n(wf1c, P_C, t * 32 + 16, s * 32, lane), bx[s], z1);
          da0 = mfma16(wfrag_t(wf2c, PT_H, s * 32, t * 32, lane), bg[s], da0);
          da1 = mfma16(wfrag_t(wf2c, PT_H, s * 32, t * 32 + 16, lane), bg[s], da1);
          if (KS > 2) __builtin_amdgcn_sched_barrier(0);  // c = 128: four k-steps x four fragments hoisted at once do not fit beside the accumulators
        }
        da0 *= gelu_both_v4(z0); da1 *= gelu_both_v4(z1);
        const uint2 dlo = pack4(da0), dhi = pack4(da1);
        if (ok) {
          const size_t off = (size_t)row * H + j * HC + t * 32 + g * 4;
          *reinterpret_cast<uint2*>(p.a + off) = pack4(z0);
          *reinterpret_cast<uint2*>(p.a + off + 16) = pack4(z1);
          *reinterpret_cast<uint2*>(p.dz + off) = dlo;
          *reinterpret_cast<uint2*>(p.dz + off + 16) = dhi;
        }
        const short8v bz = mk8(dlo.x, dlo.y, dhi.x, dhi.y);
#pragma unroll
        for (int ob = 0; ob < RB; ob++) dxn[ob] = mfma16(wfrag_t(wf1c, P_C, t * 32, ob * 16, lane), bz, dxn[ob]);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      char* nb = smem + L::BUF + ((j + 1) & 1) * L::BUF_BYTES;
      n1.store(nb, P_C, tid);
      n2.store(nb + L::WF2_OFF, PT_H, tid);
      __syncthreads();
    }
    if (!has) continue;  // (the barriers of the chunk loop are behind it)
    // LN2 backward + the residual gradient (the normalised rows are formed again from x1 - an L2 hit - instead of living in 32 registers through the chunk loop)
    f4 xh[RB];
    load_act<C>(xh, p.x1, rowc, g);
    const float rs = normalise<C>(xh, p.eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int rb = 0; rb < RB; rb++) {
      if (ok) { dgam[rb] += dxn[rb] * xh[rb]; dbet[rb] += dxn[rb]; }
      dxn[rb] *= ldsv4(vec, rb * 16 + g * 4);
      s1 += (dxn[rb][0] + dxn[rb][1]) + (dxn[rb][2] + dxn[rb][3]);
      s2 += (dxn[rb][0] * xh[rb][0] + dxn[rb][1] * xh[rb][1]) + (dxn[rb][2] * xh[rb][2] + dxn[rb][3] * xh[rb][3]);
    }
    s1 = group4_sum(s1) * (1.f / C);
    s2 = group4_sum(s2) * (1.f / C);
    short8v bdx[KS];
    {
      f4 g2r[RB];
      load_act<C>(g2r, p.g2, rowc, g);
#pragma unroll
      for (int rb = 0; rb < RB; rb++) dxn[rb] = g2r[rb] + (dxn[rb] - s1 - xh[rb] * s2) * rs;
    }
    if (ok) store_act<C>(dxn, p.dx1, row, g);
    pack_act<C>(bdx, dxn, p.dx1_16, row, g, ok);
    // conv2 data gradient
#pragma unroll
    for (int ob = 0; ob < RB; ob++) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; s++) acc = mfma16(wfrag_t(w2l, PT_C, s * 32, ob * 16, lane), bdx[s], acc);
      if (ok) *reinterpret_cast<uint2*>(p.dd + (size_t)row * C + ob * 16 + g * 4) = pack4(acc);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  emit_ln_partials<C>(dgam, dbet, reinterpret_cast<float*>(smem + L::BUF), p.partials, tid);
}
template <int C> __global__ __launch_bounds__(NT) void stem_mlp_bwd_kernel(MlpBwdP p) { mlp_bwd_body<C>(p); }
template <int C> __global__ __launch_bounds__(NT) void stem_mlp_bwd_lanes_kernel(Lanes<MlpBwdP> L) { mlp_bwd_body<C>(L.p[blockIdx.y]); }

// =====================================================================================================================================================
// weight gradients of a block: dW[n][k] += sum_r dy[r][n] x[r][k], db[n] += sum_r dy[r][n] from row-major bf16 operand pairs whose outputs are SMALL
// (n, k <= 512, n k <= 64 K): a workgroup owns the whole output for a slice of the rows (every operand byte is read once), v_mfma_f32_32x32x16_bf16 with the
// rows as the reduction (fragments by transpose reads of the row-major LDS tiles), partial slabs summed in order by stem_wgrad_reduce_kernel.
// =====================================================================================================================================================
struct WgProb {
  const bf16_t* dy; const bf16_t* x;   // [rows][n], [rows][k]
  float* dw; float* db;                // accumulated [n][k] (row stride k), [n] or null
  int rows, n, k;
  int slab_off;                        // floats: this problem's partial [n k + n] inside a workgroup's slab
};
constexpr int WG_MAX = 6;
struct WgP { WgProb pr[WG_MAX]; int count; float* slabs; int slab_floats; int n_slices; };
constexpr int WG_RT = 64;              // rows per staging step (four k-steps of 16)
constexpr int WG_MAX_PQ = 128 + 512;   // widest operand pair
__device__ __forceinline__ short8v frag32_t(const char* lds, int pitch, int kr0, int col0, int lane) {  // 32x32x16 operand from a row-major [k][col] tile
  const int q4 = lane >> 4, t = lane & 15;
  const char* p = lds + (kr0 + 8 * (q4 >> 1) + (t >> 2)) * pitch + (col0 + 16 * (q4 & 1) + 4 * (t & 3)) * 2;
  const short4v lo = lds_tr16_b64(p), hi = lds_tr16_b64(p + 4 * pitch);
  short8v o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3]; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}
// MP x MQ 32x32 tiles per wave (P = the narrower operand); RT rows per staging step.  The rows of step i + 1 are fetched into registers while step i is multiplied
// out of LDS (the loop is bound by bytes in flight: the products are a few hundred cycles per step), widths are powers of two (index arithmetic by shifts).
template <int MP, int MQ, int RT>
__device__ __forceinline__ void wgrad_body(const WgP& q) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WgProb& pr = q.pr[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool swap = pr.n > pr.k;                       // P = narrower operand
  const bf16_t* P = swap ? pr.x : pr.dy;
  const bf16_t* Q = swap ? pr.dy : pr.x;
  const int np = swap ? pr.k : pr.n, nq = swap ? pr.n : pr.k;
  const int pp = 2 * np + 32, pq = 2 * nq + 32;        // LDS pitches (bytes)
  char* lp = smem;
  char* lq = smem + RT * pp;
  const int tp = np >> 5, tq = nq >> 5;                // 32-wide tiles
  const int wp = tp >= 2 ? 2 : 1, wq = NW / wp;        // wave grid
  const int wpi = wave % wp, wqi = wave / wp;
  const int per_p = (tp + wp - 1) / wp, per_q = (tq + wq - 1) / wq;   // <= MP, MQ (host-checked)
  float16v acc[MP][MQ];
#pragma unroll
  for (int a = 0; a < MP; a++)
#pragma unroll
    for (int b = 0; b < MQ; b++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
  float bsum = 0.f;                                     // column tid of dy
  const char* ldy = swap ? lq : lp;
  const int pdy = swap ? pq : pp;
  const int per = (pr.rows + q.n_slices - 1) / q.n_slices;
  const int rows_per = ((per + RT - 1) / RT) * RT;
  const int r0 = blockIdx.x * rows_per, r1 = min(pr.rows, r0 + rows_per);
  const int sp = __ffs(np >> 3) - 1, sq = __ffs(nq >> 3) - 1;   // log2 of the 16-byte chunks per row
  const int n_p = RT << sp, n_chunks = n_p + (RT << sq);
  constexpr int NLD = (RT * WG_MAX_PQ / 8 + NT - 1) / NT;
  // two staging sets: the rows of step i + 2 are requested as soon as set (i & 1) has been written to LDS, so two steps' bytes are in flight per workgroup (with one
  // set the loop ran at one HBM latency per step: 3.1 us per 45 KB step, 2.3 TB/s over the chip)
  u4 stg_a[NLD], stg_b[NLD];
  auto fetch = [&](u4 (&stg)[NLD], int rb) {
#pragma unroll
    for (int u = 0; u < NLD; u++) {
      const int i = tid + u * NT;
      stg[u] = u4{0u, 0u, 0u, 0u};
      if (i < n_chunks) {
        const bool isq = i >= n_p;
        const int ii = isq ? i - n_p : i, sh = isq ? sq : sp;
        const int r = ii >> sh, c = ii & ((1 << sh) - 1);
        if (rb + r < r1) stg[u] = *reinterpret_cast<const u4*>((isq ? Q : P) + ((size_t)(rb + r) << (sh + 3)) + c * 8);
      }
    }
  };
  auto step = [&](u4 (&stg)[NLD], int rb) {
    __syncthreads();   // the previous step's fragments have been read
#pragma unroll
    for (int u = 0; u < NLD; u++) {
      const int i = tid + u * NT;
      if (i < n_chunks) {
        const bool isq = i >= n_p;
        const int ii = isq ? i - n_p : i, sh = isq ? sq : sp;
        const int r = ii >> sh, c = ii & ((1 << sh) - 1);
        *reinterpret_cast<u4*>((isq ? lq : lp) + r * (isq ? pq : pp) + c * 16) = stg[u];
      }
    }
    __syncthreads();
    if (rb + 2 * RT < r1) fetch(stg, rb + 2 * RT);   // in flight while this step and the next are multiplied
    if (pr.db && tid < pr.n) {
#pragma unroll 8
      for (int r = 0; r < RT; r++) bsum += bf2f(*reinterpret_cast<const bf16_t*>(ldy + r * pdy + tid * 2));
    }
#pragma unroll
    for (int ks = 0; ks < RT / 16; ks++) {
      short8v fq[MQ];
#pragma unroll
      for (int b = 0; b < MQ; b++)
        if (b < per_q && wqi * per_q + b < tq) fq[b] = frag32_t(lq, pq, ks * 16, (wqi * per_q + b) * 32, lane);
#pragma unroll
      for (int a = 0; a < MP; a++) {
        if (a >= per_p || wpi * per_p + a >= tp) continue;
        const short8v fp = frag32_t(lp, pp, ks * 16, (wpi * per_p + a) * 32, lane);
#pragma unroll
        for (int b = 0; b < MQ; b++)
          if (b < per_q && wqi * per_q + b < tq) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp, fq[b], acc[a][b], 0, 0, 0);
      }
    }
  };
  if (r0 < r1) fetch(stg_a, r0);
  if (r0 + RT < r1) fetch(stg_b, r0 + RT);
  for (int rb = r0; rb < r1; rb += 2 * RT) {
    step(stg_a, rb);
    if (rb + RT < r1) step(stg_b, rb + RT);
  }
  // partial slab of this workgroup: [n][k] in the destination's orientation, then [n] column sums
  float* slab = q.slabs + (size_t)blockIdx.x * q.slab_floats + pr.slab_off;
#pragma unroll
  for (int a = 0; a < MP; a++) {
    if (a >= per_p || wpi * per_p + a >= tp) continue;
#pragma unroll
    for (int b = 0; b < MQ; b++) {
      if (b >= per_q || wqi * per_q + b >= tq) continue;
      const int p0 = (wpi * per_p + a) * 32, q0 = (wqi * per_q + b) * 32;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ip = p0 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5), iq = q0 + (lane & 31);   // D[row = P index][col = Q index]
        if (swap) slab[(size_t)iq * pr.k + ip] = acc[a][b][i];
        else slab[(size_t)ip * pr.k + iq] = acc[a][b][i];
      }
    }
  }
  if (pr.db && tid < pr.n) slab[(size_t)pr.n * pr.k + tid] = bsum;
}
template <int MP, int MQ, int RT> __global__ __launch_bounds__(NT) void stem_wgrad_kernel(WgP q) { wgrad_body<MP, MQ, RT>(q); }
template <int MP, int MQ, int RT> __global__ __launch_bounds__(NT) void stem_wgrad_lanes_kernel(Lanes<WgP> L) { wgrad_body<MP, MQ, RT>(L.p[blockIdx.z]); }

// The same row-slice weight gradient with EVERYTHING about the shapes known at compile time (the four problems of a MaskedConvBlock are [c x 4c], [4c x c] and twice
// [c x c]): the general body above guards every staging chunk and every fragment with run-time conditions, which the compiler turns into ~40 basic blocks per step with
// a full s_waitcnt at each boundary - the second staging set was waited for before the first was refilled, and every transpose read was waited for before the next was
// issued.  Here a step is straight-line code: NP x RT / 8 = 512 chunks of the narrow operand (one per thread) + NQ / NP chunks of the wide one per thread, no guards in
// the full steps; the bias gradient is summed from the staged registers (8 columns per thread) and reduced across the threads once, at the end.
template <int NP, int NQ, int RT>
__device__ __forceinline__ void wgrad_fixed_body(const WgP& q, const WgProb& pr, char* smem) {
  constexpr int PP = 2 * NP + 32, PQ = 2 * NQ + 32;          // LDS pitches (bytes)
  constexpr int TP = NP / 32, TQ = NQ / 32, WP = 2, WQ = NW / WP;
  constexpr int MP = TP / WP, MQ = TQ >= WQ ? TQ / WQ : 1, QW = TQ >= WQ ? WQ : TQ;
  constexpr int CP = NP / 8, CQ = NQ / 8;                     // 16-byte chunks per row
  static_assert(RT * CP == NT && (RT * CQ) % NT == 0 && TP % WP == 0, "one narrow chunk per thread and step");
  constexpr int NQL = RT * CQ / NT, QSTEP = NT / CQ;          // wide chunks per thread and step, rows between them
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool swap = pr.n > pr.k;                              // P = the narrower operand
  const bf16_t* P = swap ? pr.x : pr.dy;
  const bf16_t* Q = swap ? pr.dy : pr.x;
  char* lp = smem;
  char* lq = smem + RT * PP;
  const int wpi = wave % WP, wqi = wave / WP;
  float16v acc[MP][MQ];
#pragma unroll
  for (int a = 0; a < MP; a++)
#pragma unroll
    for (int b = 0; b < MQ; b++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
  const int per = (pr.rows + q.n_slices - 1) / q.n_slices;
  const int rows_per = ((per + RT - 1) / RT) * RT;
  const int r0 = blockIdx.x * rows_per, r1 = min(pr.rows, r0 + rows_per);
  const int prow = tid / CP, pc = tid % CP, qrow = tid / CQ, qc = tid % CQ;
  float bacc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) bacc[i] = 0.f;
  struct Stage { u4 p; u4 qv[NQL]; };
  Stage sa, sb;
  auto fetch = [&](Stage& st, int rb) {
    if (rb + RT <= r1) {   // a whole step: no guards
      st.p = *reinterpret_cast<const u4*>(P + (size_t)(rb + prow) * NP + pc * 8);
#pragma unroll
      for (int u = 0; u < NQL; u++) st.qv[u] = *reinterpret_cast<const u4*>(Q + (size_t)(rb + qrow + u * QSTEP) * NQ + qc * 8);
    } else {               // the ragged last step of the tensor: missing rows are zeros
      st.p = u4{0u, 0u, 0u, 0u};
      if (rb + prow < r1) st.p = *reinterpret_cast<const u4*>(P + (size_t)(rb + prow) * NP + pc * 8);
#pragma unroll
      for (int u = 0; u < NQL; u++) {
        st.qv[u] = u4{0u, 0u, 0u, 0u};
        if (rb + qrow + u * QSTEP < r1) st.qv[u] = *reinterpret_cast<const u4*>(Q + (size_t)(rb + qrow + u * QSTEP) * NQ + qc * 8);
      }
    }
  };
  auto bias_add = [&](const u4& v) {
#pragma unroll
    for (int w = 0; w < 4; w++) { bacc[2 * w] += __uint_as_float(v[w] << 16); bacc[2 * w + 1] += __uint_as_float(v[w] & 0xffff0000u); }
  };
  auto step = [&](Stage& st, int rb) {
    __syncthreads();   // the previous step's fragments have been read
    *reinterpret_cast<u4*>(lp + prow * PP + pc * 16) = st.p;
#pragma unroll
    for (int u = 0; u < NQL; u++) *reinterpret_cast<u4*>(lq + (qrow + u * QSTEP) * PQ + qc * 16) = st.qv[u];
    if (pr.db) {
      if (!swap) bias_add(st.p);
      else {
#pragma unroll
        for (int u = 0; u < NQL; u++) bias_add(st.qv[u]);
      }
    }
    __syncthreads();
    if (rb + 2 * RT < r1) fetch(st, rb + 2 * RT);   // in flight while this step and the next are multiplied
    if (wqi < QW) {
#pragma unroll
      for (int ks = 0; ks < RT / 16; ks++) {
        short8v fq[MQ], fp[MP];
#pragma unroll
        for (int b = 0; b < MQ; b++) fq[b] = frag32_t(lq, PQ, ks * 16, (wqi * MQ + b) * 32, lane);
#pragma unroll
        for (int a = 0; a < MP; a++) fp[a] = frag32_t(lp, PP, ks * 16, (wpi * MP + a) * 32, lane);
#pragma unroll
        for (int a = 0; a < MP; a++)
#pragma unroll
          for (int b = 0; b < MQ; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp[a], fq[b], acc[a][b], 0, 0, 0);
      }
    }
  };
  if (r0 < r1) fetch(sa, r0);
  if (r0 + RT < r1) fetch(sb, r0 + RT);
  for (int rb = r0; rb < r1; rb += 2 * RT) {
    step(sa, rb);
    if (rb + RT < r1) step(sb, rb + RT);
  }
  float* slab = q.slabs + (size_t)blockIdx.x * q.slab_floats + pr.slab_off;
  if (wqi < QW) {
#pragma unroll
    for (int a = 0; a < MP; a++)
#pragma unroll
      for (int b = 0; b < MQ; b++) {
        const int p0 = (wpi * MP + a) * 32, q0 = (wqi * MQ + b) * 32;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int ip = p0 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5), iq = q0 + (lane & 31);   // D[row = P index][col = Q index]
          if (swap) slab[(size_t)iq * pr.k + ip] = acc[a][b][i];
          else slab[(size_t)ip * pr.k + iq] = acc[a][b][i];
        }
      }
  }
  if (pr.db) {   // column sums of dy: the threads that staged the same 8 columns hold partial sums over different rows
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [8 columns of a chunk][NT threads] - 16 KB, the staging tiles are free now
#pragma unroll
    for (int i = 0; i < 8; i++) red[i * NT + tid] = bacc[i];
    __syncthreads();
    const int cd = swap ? CQ : CP;                  // chunks per dy row; thread t staged chunk t % cd
    if (tid < pr.n) {
      const int c = tid >> 3, i = tid & 7;
      float sum = 0.f;
      for (int t = c; t < NT; t += cd) sum += red[i * NT + t];
      slab[(size_t)pr.n * pr.k + tid] = sum;
    }
  }
}
template <int C, int RT>
__device__ __forceinline__ void wgrad_block_body(const WgP& q) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WgProb& pr = q.pr[blockIdx.y];
  if (pr.n == pr.k) wgrad_fixed_body<C, C, RT>(q, pr, smem);
  else wgrad_fixed_body<C, 4 * C, RT>(q, pr, smem);
}
template <int C, int RT> __global__ __launch_bounds__(NT) void stem_wgrad_block_kernel(WgP q) { wgrad_block_body<C, RT>(q); }
template <int C, int RT> __global__ __launch_bounds__(NT) void stem_wgrad_block_lanes_kernel(Lanes<WgP> L) { wgrad_block_body<C, RT>(L.p[blockIdx.z]); }

__device__ __forceinline__ void wgrad_reduce_body(const WgP& q) {  // block = 32 elements x 8 slice groups
  __shared__ float red[8][32];
  const WgProb& pr = q.pr[blockIdx.y];
  const int total = pr.n * pr.k + (pr.db ? pr.n : 0);
  const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e;
  if (blockIdx.x * 32 >= total) return;
  const int per = (q.n_slices + 7) / 8, b0 = grp * per, b1 = min(q.n_slices, b0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (i < total) {
    const float* s = q.slabs + pr.slab_off + i;
    int b = b0;
    for (; b + 3 < b1; b += 4) {
      a0 += s[(size_t)b * q.slab_floats]; a1 += s[(size_t)(b + 1) * q.slab_floats]; a2 += s[(size_t)(b + 2) * q.slab_floats]; a3 += s[(size_t)(b + 3) * q.slab_floats];
    }
    for (; b < b1; b++) a0 += s[(size_t)b * q.slab_floats];
  }
  red[grp][e] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && i < total) {
    const float t = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) + ((red[4][e] + red[5][e]) + (red[6][e] + red[7][e]));
    if (i < pr.n * pr.k) pr.dw[i] += t;
    else pr.db[i - pr.n * pr.k] += t;
  }
}
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(WgP q) { wgrad_reduce_body(q); }
__global__ __launch_bounds__(256) void stem_wgrad_reduce_lanes_kernel(Lanes<WgP> L) { wgrad_reduce_body(L.p[blockIdx.z]); }

int n_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  return cus;
}
int grid_for(int rows, int per_cu) {
  const int tiles = (rows + 15) / 16;
  const int want = (tiles + NW - 1) / NW, cap = n_cus() * per_cu;
  return want < cap ? want : cap;
}
template <typename K1, typename K2>
int set_lds(bool (&flags)[16], K1 single, K2 lanes, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (flags[dev]) return 0;
  if (hipFuncSetAttribute((const void*)single, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)lanes, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 1;
  flags[dev] = true;
  return 0;
}

}  // namespace

CINEMA_API int cinema_stem_supported(int c) { return c == 64 || c == 128; }

CINEMA_API int cinema_stem_ln_linear(const float* x, const float* gamma, const float* beta, float eps, const uint16_t* w, const float* bias, uint16_t* xn, uint16_t* h,
                                     int rows, int c, void* stream) {
  if (!x || !gamma || !beta || !w || !h || rows <= 0) return CINEMA_ERR_BAD_ARG;
  if (!cinema_stem_supported(c)) return CINEMA_ERR_UNSUPPORTED;
  const LnLinP p{x, gamma, beta, w, bias, xn, h, rows, eps};
  const int lds = c * (2 * c + 16) + 3 * c * 4;
  const dim3 grid(grid_for(rows, 2));
  if (c == 64) {
    static bool f[16] = {};
    if (set_lds(f, stem_ln_linear_kernel<64>, stem_ln_linear_lanes_kernel<64>, lds)) return CINEMA_ERR_UNSUPPORTED;
    launch_lanes(stem_ln_linear_kernel<64>, stem_ln_linear_lanes_kernel<64>, 1, grid, dim3(NT), lds, (hipStream_t)stream, p);
  } else {
    static bool f[16] = {};
    if (set_lds(f, stem_ln_linear_kernel<128>, stem_ln_linear_lanes_kernel<128>, lds)) return CINEMA_ERR_UNSUPPORTED;
    launch_lanes(stem_ln_linear_kernel<128>, stem_ln_linear_lanes_kernel<128>, 1, grid, dim3(NT), lds, (hipStream_t)stream, p);
  }
  return launch_status();
}

CINEMA_API int cinema_stem_partials(int rows) { return grid_for(rows, 2); }

CINEMA_API int cinema_stem_ln_linear_bwd(const uint16_t* dh, const float* x, const float* dres, const float* gamma, float eps, const uint16_t* w, float* dx, float* partials,
                                         int rows, int c, int* n_partials_out, void* stream) {
  if (!dh || !x || !gamma || !w || !dx || !partials || !n_partials_out || rows <= 0) return CINEMA_ERR_BAD_ARG;
  if (!cinema_stem_supported(c)) return CINEMA_ERR_UNSUPPORTED;
  const LnLinBwdP p{dh, x, dres, gamma, w, dx, partials, rows, eps};
  const int lds = c * (2 * c + 32) + c * 4 + NW * 2 * c * 4;
  const dim3 grid(grid_for(rows, 2));
  *n_partials_out = (int)grid.x;
  if (c == 64) {
    static bool f[16] = {};
    if (set_lds(f, stem_ln_linear_bwd_kernel<64>, stem_ln_linear_bwd_lanes_kernel<64>, lds)) return CINEMA_ERR_UNSUPPORTED;
    launch_lanes(stem_ln_linear_bwd_kernel<64>, stem_ln_linear_bwd_lanes_kernel<64>, 1, grid, dim3(NT), lds, (hipStream_t)stream, p);
  } else {
    static bool f[16] = {};
    if (set_lds(f, stem_ln_linear_bwd_kernel<128>, stem_ln_linear_bwd_lanes_kernel<128>, lds)) return CINEMA_ERR_UNSUPPORTED;
    launch_lanes(stem_ln_linear_bwd_kernel<128>, stem_ln_linear_bwd_lanes_kernel<128>, 1, grid, dim3(NT), lds, (hipStream_t)stream, p);
  }
  return launch_status();
}

CINEMA_API int cinema_stem_mlp_fwd(const uint16_t* d, const float* x, const uint16_t* w2, const float* b2, const float* gamma, const float* beta, float eps, const uint16_t* wf1,
                                   const float* bf1, const uint16_t* wf2, const float* bf2, float* x1, float* x2, int rows, int c, void* stream) {
  if (!d || !x || !w2 || !b2 || !gamma || !beta || !wf1 || !bf1 || !wf2 || !bf2 || !x2 || rows <= 0) return CINEMA_ERR_BAD_ARG;
  if (!cinema_stem_supported(c)) return CINEMA_ERR_UNSUPPORTED;
  const MlpFwdP p{d, x, w2, b2, gamma, beta, wf1, bf1, wf2, bf2, x1, x2, rows, eps};
  if (c == 64) {
    static bool f[16] = {};
    constexpr int lds = MlpFwdLds<64, true>::BYTES;
    if (set_lds(f, stem_mlp_fwd_kernel<64>, stem_mlp_fwd_lanes_kernel<64>, lds)) return CINEMA_ERR_UNSUPPORTED;
    const int tiles = (rows + 15) / 16, want = (tiles + 15) / 16;
    launch_lanes(stem_mlp_fwd_kernel<64>, stem_mlp_fwd_lanes_kernel<64>, 1, dim3(want < n_cus() ? want : n_cus()), dim3(16 * 64), lds, (hipStream_t)stream, p);
  } else {
    static bool f[16] = {};
    constexpr int lds = MlpFwdLds<128, false>::BYTES;
    if (set_lds(f, stem_mlp_fwd_kernel<128>, stem_mlp_fwd_lanes_kernel<128>, lds)) return CINEMA_ERR_UNSUPPORTED;
    launch_lanes(stem_mlp_fwd_kernel<128>, stem_mlp_fwd_lanes_kernel<128>, 1, dim3(grid_for(rows, 1)), dim3(NT), lds, (hipStream_t)stream, p);
  }
  return launch_status();
}

CINEMA_API int cinema_stem_mlp_bwd(const float* g2, const float* x1, const uint16_t* w2, const float* gamma, const float* beta, float eps, const uint16_t* wf1, const float* bf1,
                                   const uint16_t* wf2, float* dx1, uint16_t* dx1_16, uint16_t* dd, uint16_t* a, uint16_t* dz, uint16_t* xn2, uint16_t* g2_16, float* partials,
                                   int rows, int c, int* n_partials_out, void* stream) {
  if (!g2 || !x1 || !w2 || !gamma || !beta || !wf1 || !bf1 || !wf2 || !dx1 || !dx1_16 || !dd || !a || !dz || !xn2 || !g2_16 || !partials || !n_partials_out || rows <= 0)
    return CINEMA_ERR_BAD_ARG;
  if (!cinema_stem_supported(c)) return CINEMA_ERR_UNSUPPORTED;
  const MlpBwdP p{g2, x1, w2, gamma, beta, wf1, bf1, wf2, dx1, dx1_16, dd, a, dz, xn2, g2_16, partials, rows, eps};
  if (c == 64) {
    static bool f[16] = {};
    constexpr int lds = MlpBwdLds<64>::BYTES;
    const dim3 grid(grid_for(rows, 2));
    *n_partials_out = (int)grid.x;
    if (set_lds(f, stem_mlp_bwd_kernel<64>, stem_mlp_bwd_lanes_kernel<64>, lds)) return CINEMA_ERR_UNSUPPORTED;
    launch_lanes(stem_mlp_bwd_kernel<64>, stem_mlp_bwd_lanes_kernel<64>, 1, grid, dim3(NT), lds, (hipStream_t)stream, p);
  } else {
    static bool f[16] = {};
    constexpr int lds = MlpBwdLds<128>::BYTES;
    const dim3 grid(grid_for(rows, 1));
    *n_partials_out = (int)grid.x;
    if (set_lds(f, stem_mlp_bwd_kernel<128>, stem_mlp_bwd_lanes_kernel<128>, lds)) return CINEMA_ERR_UNSUPPORTED;
    launch_lanes(stem_mlp_bwd_kernel<128>, stem_mlp_bwd_lanes_kernel<128>, 1, grid, dim3(NT), lds, (hipStream_t)stream, p);
  }
  return launch_status();
}

CINEMA_API int cinema_stem_wgrad_slices(int rows) {
  int s = (rows + 255) / 256;
  // a block hands over four problems (fc2, fc1, conv2, conv1) that run side by side, one workgroup per CU: a quarter of the CUs per problem is one round of the chip,
  // and the partial slabs (and the reduce pass over them) shrink fourfold against one slice per CU (42 MB -> 10 MB at config 2's stage 1)
  const int cap = n_cus() / 4 > 32 ? n_cus() / 4 : 32;
  return s < 1 ? 1 : (s > cap ? cap : s);
}
CINEMA_API long long cinema_stem_wgrad_workspace_bytes(const cinema_stem_wgrad_problem* probs, int count) {
  if (!probs || count <= 0 || count > WG_MAX) return 0;
  long long fl = 0;
  for (int i = 0; i < count; i++) fl += (long long)probs[i].n * probs[i].k + probs[i].n;
  return fl * 4 * cinema_stem_wgrad_slices(probs[0].rows);
}
CINEMA_API int cinema_stem_wgrad(const cinema_stem_wgrad_problem* probs, int count, float* workspace, long long workspace_bytes, void* stream) {
  if (!probs || count <= 0 || count > WG_MAX || !workspace) return CINEMA_ERR_BAD_ARG;
  WgP q{};
  q.count = count;
  int off = 0, max_np = 0, max_nq = 0;
  for (int i = 0; i < count; i++) {
    const cinema_stem_wgrad_problem& s = probs[i];
    if (!s.dy || !s.x || !s.dw || s.rows <= 0 || s.rows != probs[0].rows) return CINEMA_ERR_BAD_ARG;
    if ((s.n & (s.n - 1)) || (s.k & (s.k - 1)) || s.n < 32 || s.k < 32 || s.n > 512 || s.k > 512 || s.n + s.k > WG_MAX_PQ) return CINEMA_ERR_UNSUPPORTED;  // powers of two
    q.pr[i] = WgProb{s.dy, s.x, s.dw, s.db, s.rows, s.n, s.k, off};
    off += s.n * s.k + s.n;
    const int np = s.n < s.k ? s.n : s.k, nq = s.n < s.k ? s.k : s.n;
    if (np > max_np) max_np = np;
    if (nq > max_nq) max_nq = nq;
  }
  // tiles per wave: P direction over 2 waves (1 if a single tile), Q direction over the rest
  int mp = 1, mq = 1;
  for (int i = 0; i < count; i++) {
    const int np = (q.pr[i].n < q.pr[i].k ? q.pr[i].n : q.pr[i].k) >> 5, nq = (q.pr[i].n < q.pr[i].k ? q.pr[i].k : q.pr[i].n) >> 5;
    const int wp = np >= 2 ? 2 : 1, wq = NW / wp;
    const int pp = (np + wp - 1) / wp, pq = (nq + wq - 1) / wq;
    if (pp > mp) mp = pp;
    if (pq > mq) mq = pq;
  }
  if (mp > 2 || mq > 4) return CINEMA_ERR_UNSUPPORTED;
  q.slabs = workspace;
  q.slab_floats = off;
  q.n_slices = cinema_stem_wgrad_slices(probs[0].rows);
  if (workspace_bytes < (long long)off * 4 * q.n_slices) return CINEMA_ERR_BAD_ARG;
  // the four problems of a MaskedConvBlock ([c x 4c], [4c x c], [c x c] twice; c = 64 / 128): the compile-time form
  int block_c = 0;
  for (int c : {64, 128}) {
    bool all = true;
    for (int i = 0; i < count; i++) {
      const int np = q.pr[i].n < q.pr[i].k ? q.pr[i].n : q.pr[i].k, nq = q.pr[i].n < q.pr[i].k ? q.pr[i].k : q.pr[i].n;
      all = all && np == c && (nq == c || nq == 4 * c);
    }
    if (all) block_c = c;
  }
  const bool big = !(mp <= 1 && mq <= 2);
  const int lds = (big ? 32 : WG_RT) * ((2 * max_np + 32) + (2 * max_nq + 32));
  const dim3 grid(q.n_slices, count);
  hipStream_t st = (hipStream_t)stream;
  if (block_c == 64) {
    static bool f[16] = {};
    if (set_lds(f, stem_wgrad_block_kernel<64, 64>, stem_wgrad_block_lanes_kernel<64, 64>, 96 * 1024)) return CINEMA_ERR_UNSUPPORTED;
    launch_lanes(stem_wgrad_block_kernel<64, 64>, stem_wgrad_block_lanes_kernel<64, 64>, 2, grid, dim3(NT), 64 * ((2 * 64 + 32) + (2 * 256 + 32)), st, q);
  } else if (block_c == 128) {
    static bool f[16] = {};
    if (set_lds(f, stem_wgrad_block_kernel<128, 32>, stem_wgrad_block_lanes_kernel<128, 32>, 96 * 1024)) return CINEMA_ERR_UNSUPPORTED;
    launch_lanes(stem_wgrad_block_kernel<128, 32>, stem_wgrad_block_lanes_kernel<128, 32>, 2, grid, dim3(NT), 32 * ((2 * 128 + 32) + (2 * 512 + 32)), st, q);
  } else if (!big) {
    static bool f[16] = {};
    if (set_lds(f, stem_wgrad_kernel<1, 2, WG_RT>, stem_wgrad_lanes_kernel<1, 2, WG_RT>, 96 * 1024)) return CINEMA_ERR_UNSUPPORTED;
    launch_lanes(stem_wgrad_kernel<1, 2, WG_RT>, stem_wgrad_lanes_kernel<1, 2, WG_RT>, 2, grid, dim3(NT), lds, st, q);
  } else {
    static bool f[16] = {};
    if (set_lds(f, stem_wgrad_kernel<2, 4, 32>, stem_wgrad_lanes_kernel<2, 4, 32>, 96 * 1024)) return CINEMA_ERR_UNSUPPORTED;
    launch_lanes(stem_wgrad_kernel<2, 4, 32>, stem_wgrad_lanes_kernel<2, 4, 32>, 2, grid, dim3(NT), lds, st, q);
  }
  int max_total = 0;
  for (int i = 0; i < count; i++) { const int t = q.pr[i].n * q.pr[i].k + q.pr[i].n; if (t > max_total) max_total = t; }
  launch_lanes(stem_wgrad_reduce_kernel, stem_wgrad_reduce_lanes_kernel, 2, dim3((max_total + 31) / 32, count), dim3(256), 0, st, q);
  return launch_status();
}
