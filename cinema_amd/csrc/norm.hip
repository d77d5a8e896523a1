// LayerNorm forward / backward for gfx950: sub-wave row groups (4..64 lanes per row), 16-byte accesses,
// xor-shuffle reductions, optional fused exact GELU (ConvNormActBlock), fused residual-gradient add and
// bf16 copy in the backward, dgamma/dbeta block-reduced then fp32 atomics.
#include "common.cuh"
#include <initializer_list>
#include <type_traits>
#include "../../include/cinema_hip.h"

namespace {

__device__ __forceinline__ float4 load4(const void* base, bool is_bf16, size_t off) {
  if (is_bf16) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(base) + off);
    return make_float4(bf2f((bf16_t)(u.x & 0xffff)), bf2f((bf16_t)(u.x >> 16)), bf2f((bf16_t)(u.y & 0xffff)), bf2f((bf16_t)(u.y >> 16)));
  }
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
}
__device__ __forceinline__ void store4_bf16(bf16_t* p, float4 v) {
  uint2 u; u.x = pack_bf2(v.x, v.y); u.y = pack_bf2(v.z, v.w);
  *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ float group_sum(float v, int lpr) {
  for (int o = lpr >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct LnFwdP {
  const void* x; int x_bf16; int ldx;
  const float* gamma; const float* beta;
  int rows, c; float eps; int act;
  bf16_t* y_bf16; float* y_f32; int ldy;
  float* mean; float* rstd;
  int lpr;  // lanes per row (power of two, <= 64)
  uint8_t* y_fp8; float* row_scale;  // optional e4m3 copy of y with ONE scale per row (amax(row) / 448): the A operand of cinema_gemm_fp8(row scales)
  const float* q8_inv; unsigned int* q8_amax;  // per-TENSOR delayed scale instead (cinema_q8_out): y_fp8 = e4m3(sat(y * *q8_inv)), the launch's maximum into q8_amax
};

template <int CPL>
__device__ __forceinline__ void ln_fwd_body(const LnFwdP& p) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & (p.lpr - 1);
  const int rows_per_wave = 64 / p.lpr;
  const int wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * (blockDim.x >> 6);
  const int nch = p.c >> 2;
  const float inv_c = 1.f / (float)p.c;
  float tmax = 0.f;
  const float tinv = (p.q8_amax && p.y_fp8) ? *p.q8_inv : 1.f;  // once, ahead of the row loop
  for (int row0 = wave_global * rows_per_wave; row0 < p.rows; row0 += n_waves * rows_per_wave) {
    const int row = row0 + lane / p.lpr;
    const bool rv = row < p.rows;
    float4 v[CPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; i++) {
      const int ch = sub + i * p.lpr;
      v[i] = (rv && ch < nch) ? load4(p.x, p.x_bf16, (size_t)row * p.ldx + ch * 4) : make_float4(0, 0, 0, 0);
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mu = group_sum(s, p.lpr) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; i++) {
      const int ch = sub + i * p.lpr;
      if (ch < nch) {
        const float a = v[i].x - mu, b = v[i].y - mu, c2 = v[i].z - mu, d = v[i].w - mu;
        q += a * a + b * b + c2 * c2 + d * d;
      }
    }
    const float rs = rsqrtf(group_sum(q, p.lpr) * inv_c + p.eps);
    if (!rv) continue;
    if (sub == 0) {
      if (p.mean) p.mean[row] = mu;
      if (p.rstd) p.rstd[row] = rs;
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; i++) {
      const int ch = sub + i * p.lpr;
      if (ch >= nch) continue;
      const float4 g = *reinterpret_cast<const float4*>(p.gamma + ch * 4);
      const float4 b = *reinterpret_cast<const float4*>(p.beta + ch * 4);
      float4 y;
      y.x = (v[i].x - mu) * rs * g.x + b.x; y.y = (v[i].y - mu) * rs * g.y + b.y;
      y.z = (v[i].z - mu) * rs * g.z + b.z; y.w = (v[i].w - mu) * rs * g.w + b.w;
      if (p.act == 1) { gelu2(y.x, y.y); gelu2(y.z, y.w); }
      if (p.y_bf16) store4_bf16(p.y_bf16 + (size_t)row * p.ldy + ch * 4, y);
      if (p.y_f32) *reinterpret_cast<float4*>(p.y_f32 + (size_t)row * p.ldy + ch * 4) = y;
      if (p.y_fp8 || p.q8_amax) { v[i] = y; amax = fmaxf(amax, fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w)))); }
    }
    if (p.q8_amax) {  // per-tensor delayed scale: no reduction at all in the row, the maximum is committed once per wave after the loop
      tmax = fmaxf(tmax, amax);
      if (p.y_fp8) {
        const float inv = tinv;
#pragma unroll
        for (int i = 0; i < CPL; i++) {
          const int ch = sub + i * p.lpr;
          if (ch >= nch) continue;
          *reinterpret_cast<int*>(p.y_fp8 + (size_t)row * p.c + ch * 4) = q8_pack4(v[i].x, v[i].y, v[i].z, v[i].w, inv);
        }
      }
    } else if (p.y_fp8) {  // per-row e4m3 copy: the row's maximum is a reduction over the row's own lanes only, so the quantisation costs no extra pass
      for (int o = p.lpr >> 1; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
      const float scale = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f, inv = 1.0f / scale;
      if (sub == 0) p.row_scale[row] = scale;
#pragma unroll
      for (int i = 0; i < CPL; i++) {
        const int ch = sub + i * p.lpr;
        if (ch >= nch) continue;
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].x * inv, v[i].y * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].z * inv, v[i].w * inv, w, true);
        *reinterpret_cast<int*>(p.y_fp8 + (size_t)row * p.c + ch * 4) = w;
      }
    }
  }
  if (p.q8_amax) q8_amax_commit(p.q8_amax, tmax, wave_global);
}
template <int CPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwdP p) { ln_fwd_body<CPL>(p); }
template <int CPL>
__global__ __launch_bounds__(256) void ln_fwd_lanes_kernel(Lanes<LnFwdP> L) { ln_fwd_body<CPL>(L.p[blockIdx.y]); }
struct LnBwdP {
  const void* dy; int dy_bf16; int lddy;
  const void* x; int x_bf16; int ldx;
  const float* gamma; const float* beta; const float* mean; const float* rstd;
  int rows, c, act;
  const float* dx_res; float* dx_f32; bf16_t* dx_bf16; int lddx;
  float* dgamma; float* dbeta;
  int lpr;
  float* ws;  // per-block partial sums [gridDim.x][2][c] (then ln_param_reduce_kernel) or nullptr: atomics from every block
  uint8_t* dx_fp8; const float* q8_inv; unsigned int* q8_amax;  // optional 8-bit copy of dx, dense [rows][c] (cinema_q8_out)
  float* dcol;  // optional: dcol[c] += column sums of dx (the bias gradient of the projection that produced x), a THIRD row of the per-block partials
};

// RG = independent row groups per wave iteration: narrow rows (CPL <= 2) carry only 2-4 16-byte loads per lane, too few
// bytes in flight to cover the HBM latency (measured 2.5 TB/s at C = 64), so those instantiations interleave RG groups.
template <int CPL, int RG, bool DCOL>
__device__ __forceinline__ void ln_bwd_body(const LnBwdP& p) {  // DCOL: also the column sums of dx (a compile-time flag: its accumulators cost 12-16 registers, a wave per SIMD at c = 768)
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  float* red = reinterpret_cast<float*>(dyn_smem);  // [n_waves_in_block][2][c]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int sub = lane & (p.lpr - 1);
  const int rows_per_wave = 64 / p.lpr;
  const int wave_global = blockIdx.x * nw + wave;
  const int n_waves = gridDim.x * nw;
  const int nch = p.c >> 2;
  const float inv_c = 1.f / (float)p.c;
  float4 ag[CPL], ab[CPL];
#pragma unroll
  for (int i = 0; i < CPL; i++) { ag[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0); }
  float qmax = 0.f;
  const float q8inv = (p.q8_amax && p.dx_fp8) ? *p.q8_inv : 1.f;
  float4 ac[CPL];
#pragma unroll
  for (int i = 0; i < CPL; i++) ac[i] = make_float4(0, 0, 0, 0);

  for (int row0 = wave_global * rows_per_wave * RG; row0 < p.rows; row0 += n_waves * rows_per_wave * RG) {
    float4 xh[RG][CPL], dxh[RG][CPL];
    float s1[RG], s2[RG], rsv[RG];
    bool rvv[RG];
    // phase 1: every load of the RG groups is issued before the first use
    float4 xv[RG][CPL], dv[RG][CPL];
#pragma unroll
    for (int g = 0; g < RG; g++) {
      const int row = row0 + g * rows_per_wave + lane / p.lpr;
      rvv[g] = row < p.rows;
#pragma unroll
      for (int i = 0; i < CPL; i++) {
        const int ch = sub + i * p.lpr;
        xv[g][i] = make_float4(0, 0, 0, 0); dv[g][i] = make_float4(0, 0, 0, 0);
        if (rvv[g] && ch < nch) {
          xv[g][i] = load4(p.x, p.x_bf16, (size_t)row * p.ldx + ch * 4);
          dv[g][i] = load4(p.dy, p.dy_bf16, (size_t)row * p.lddy + ch * 4);
        }
      }
    }
#pragma unroll
    for (int g = 0; g < RG; g++) {
      const int row = row0 + g * rows_per_wave + lane / p.lpr;
      const bool rv = rvv[g];
      const float mu = rv ? p.mean[row] : 0.f, rs = rv ? p.rstd[row] : 0.f;
      rsv[g] = rs;
      s1[g] = 0.f; s2[g] = 0.f;
#pragma unroll
      for (int i = 0; i < CPL; i++) {
        const int ch = sub + i * p.lpr;
        xh[g][i] = make_float4(0, 0, 0, 0); dxh[g][i] = make_float4(0, 0, 0, 0);
        if (rv && ch < nch) {
          const float4 x4 = xv[g][i];
          float4 d = dv[g][i];
          const float4 gm = *reinterpret_cast<const float4*>(p.gamma + ch * 4);
          xh[g][i] = make_float4((x4.x - mu) * rs, (x4.y - mu) * rs, (x4.z - mu) * rs, (x4.w - mu) * rs);
          if (p.act == 1) {
            const float4 b = *reinterpret_cast<const float4*>(p.beta + ch * 4);
            float g0, g1, g2, g3;
            gelu_grad2(xh[g][i].x * gm.x + b.x, xh[g][i].y * gm.y + b.y, g0, g1);
            gelu_grad2(xh[g][i].z * gm.z + b.z, xh[g][i].w * gm.w + b.w, g2, g3);
            d.x *= g0; d.y *= g1; d.z *= g2; d.w *= g3;
          }
          ag[i].x += d.x * xh[g][i].x; ag[i].y += d.y * xh[g][i].y; ag[i].z += d.z * xh[g][i].z; ag[i].w += d.w * xh[g][i].w;
          ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
          dxh[g][i] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
          s1[g] += dxh[g][i].x + dxh[g][i].y + dxh[g][i].z + dxh[g][i].w;
          s2[g] += dxh[g][i].x * xh[g][i].x + dxh[g][i].y * xh[g][i].y + dxh[g][i].z * xh[g][i].z + dxh[g][i].w * xh[g][i].w;
        }
      }
    }
#pragma unroll
    for (int g = 0; g < RG; g++) {
      s1[g] = group_sum(s1[g], p.lpr) * inv_c;
      s2[g] = group_sum(s2[g], p.lpr) * inv_c;
    }
#pragma unroll
    for (int g = 0; g < RG; g++) {
      if (!rvv[g]) continue;
      const int row = row0 + g * rows_per_wave + lane / p.lpr;
      const float rs = rsv[g];
#pragma unroll
      for (int i = 0; i < CPL; i++) {
        const int ch = sub + i * p.lpr;
        if (ch >= nch) continue;
        float4 dx;
        dx.x = rs * (dxh[g][i].x - s1[g] - xh[g][i].x * s2[g]); dx.y = rs * (dxh[g][i].y - s1[g] - xh[g][i].y * s2[g]);
        dx.z = rs * (dxh[g][i].z - s1[g] - xh[g][i].z * s2[g]); dx.w = rs * (dxh[g][i].w - s1[g] - xh[g][i].w * s2[g]);
        const size_t off = (size_t)row * p.lddx + ch * 4;
        if (p.dx_res) {
          const float4 r = *reinterpret_cast<const float4*>(p.dx_res + off);
          dx.x += r.x; dx.y += r.y; dx.z += r.z; dx.w += r.w;
        }
        if (p.dx_f32) *reinterpret_cast<float4*>(p.dx_f32 + off) = dx;
        if (p.dx_bf16) store4_bf16(p.dx_bf16 + off, dx);
        if (DCOL) { ac[i].x += dx.x; ac[i].y += dx.y; ac[i].z += dx.z; ac[i].w += dx.w; }
        if (p.q8_amax) {
          qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(dx.x), fabsf(dx.y)), fmaxf(fabsf(dx.z), fabsf(dx.w))));
          if (p.dx_fp8) *reinterpret_cast<int*>(p.dx_fp8 + (size_t)row * p.c + ch * 4) = q8_pack4(dx.x, dx.y, dx.z, dx.w, q8inv);
        }
      }
    }
  }
  if (p.q8_amax) q8_amax_commit(p.q8_amax, qmax, wave_global);
  if (!p.dgamma && !p.dbeta && !DCOL) return;
  // reduce over the row sub-groups of the wave (lanes that own the same columns), then over the block's waves; NR rows per wave / block: d gamma, d beta[, columns of dx]
  constexpr int NR = DCOL ? 3 : 2;
#pragma unroll
  for (int i = 0; i < CPL; i++) {
    for (int o = p.lpr; o < 64; o <<= 1) {
      ag[i].x += __shfl_xor(ag[i].x, o, 64); ag[i].y += __shfl_xor(ag[i].y, o, 64);
      ag[i].z += __shfl_xor(ag[i].z, o, 64); ag[i].w += __shfl_xor(ag[i].w, o, 64);
      ab[i].x += __shfl_xor(ab[i].x, o, 64); ab[i].y += __shfl_xor(ab[i].y, o, 64);
      ab[i].z += __shfl_xor(ab[i].z, o, 64); ab[i].w += __shfl_xor(ab[i].w, o, 64);
      if (DCOL) {
        ac[i].x += __shfl_xor(ac[i].x, o, 64); ac[i].y += __shfl_xor(ac[i].y, o, 64);
        ac[i].z += __shfl_xor(ac[i].z, o, 64); ac[i].w += __shfl_xor(ac[i].w, o, 64);
      }
    }
    const int ch = sub + i * p.lpr;
    if (lane < p.lpr && ch < nch) {
      *reinterpret_cast<float4*>(red + (size_t)(wave * NR + 0) * p.c + ch * 4) = ag[i];
      *reinterpret_cast<float4*>(red + (size_t)(wave * NR + 1) * p.c + ch * 4) = ab[i];
      if (DCOL) *reinterpret_cast<float4*>(red + (size_t)(wave * NR + 2) * p.c + ch * 4) = ac[i];
    }
  }
  __syncthreads();
  for (int col = threadIdx.x; col < p.c; col += blockDim.x) {
    float g = 0.f, b = 0.f, cs = 0.f;
    for (int w = 0; w < nw; w++) {
      g += red[(size_t)(w * NR + 0) * p.c + col]; b += red[(size_t)(w * NR + 1) * p.c + col];
      if (DCOL) cs += red[(size_t)(w * NR + 2) * p.c + col];
    }
    if (p.ws) {  // plain coalesced stores; 1024 blocks x 2c fp32 atomics on 2c addresses cost as much as the whole streaming pass
      p.ws[((size_t)blockIdx.x * NR + 0) * p.c + col] = g;
      p.ws[((size_t)blockIdx.x * NR + 1) * p.c + col] = b;
      if (DCOL) p.ws[((size_t)blockIdx.x * NR + 2) * p.c + col] = cs;
    } else {
      if (p.dgamma) unsafeAtomicAdd(p.dgamma + col, g);
      if (p.dbeta) unsafeAtomicAdd(p.dbeta + col, b);
      if (DCOL) unsafeAtomicAdd(p.dcol + col, cs);
    }
  }
}
template <int CPL, int RG>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdP p) { ln_bwd_body<CPL, RG, false>(p); }
template <int CPL, int RG>
__global__ __launch_bounds__(256) void ln_bwd_lanes_kernel(Lanes<LnBwdP> L) { ln_bwd_body<CPL, RG, false>(L.p[blockIdx.y]); }
template <int CPL, int RG>
__global__ __launch_bounds__(256) void ln_bwd_dcol_kernel(LnBwdP p) { ln_bwd_body<CPL, RG, true>(p); }  // + column sums of dx (fp8 path: the producing projection's bias gradient)
// dgamma[col] += sum_blocks ws[block][0][col], dbeta likewise.  Workgroup = 64 columns of the [2c] row x 4 sub-slices of the block range
// (combined through LDS), grid.y = 16 slices: 64 independent partial rows per column group in flight, 16 atomics per column.
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* ws, int nblocks, int c, float* dgamma, float* dbeta) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const int slices = gridDim.y * 4;
  const int per = (nblocks + slices - 1) / slices;
  const int b0 = (blockIdx.y * 4 + sub) * per, b1 = min(nblocks, b0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < 2 * c) {
    const float* src = ws + col;
    const size_t ld = (size_t)2 * c;
    int b = b0;
    for (; b + 3 < b1; b += 4) { s0 += src[b * ld]; s1 += src[(b + 1) * ld]; s2 += src[(b + 2) * ld]; s3 += src[(b + 3) * ld]; }
    for (; b < b1; b++) s0 += src[b * ld];
  }
  red[sub][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sub == 0 && col < 2 * c) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    float* dst = col < c ? dgamma : dbeta;
    if (dst) unsafeAtomicAdd(dst + (col < c ? col : col - c), t);
  }
}

// Channel counts that are not a multiple of 4 (the raw-image block of the segmentation decoder normalises ONE channel, reference
// cinema/segmentation/convunetr.py:320-329): one thread per row, scalar accesses, c <= 64.  Same math as the vector kernels.
__device__ __forceinline__ float ld1(const void* base, int is_bf16, size_t i) {
  return is_bf16 ? bf2f(reinterpret_cast<const bf16_t*>(base)[i]) : reinterpret_cast<const float*>(base)[i];
}
__global__ __launch_bounds__(256) void ln_fwd_small_kernel(LnFwdP p) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= p.rows) return;
  float s = 0.f;
  for (int j = 0; j < p.c; j++) s += ld1(p.x, p.x_bf16, (size_t)row * p.ldx + j);
  const float mu = s / (float)p.c;
  float q = 0.f;
  for (int j = 0; j < p.c; j++) { const float d = ld1(p.x, p.x_bf16, (size_t)row * p.ldx + j) - mu; q += d * d; }
  const float rs = rsqrtf(q / (float)p.c + p.eps);
  if (p.mean) p.mean[row] = mu;
  if (p.rstd) p.rstd[row] = rs;
  for (int j = 0; j < p.c; j++) {
    float y = (ld1(p.x, p.x_bf16, (size_t)row * p.ldx + j) - mu) * rs * p.gamma[j] + p.beta[j];
    if (p.act == 1) y = gelu_f(y);
    if (p.y_bf16) p.y_bf16[(size_t)row * p.ldy + j] = f2bf(y);
    if (p.y_f32) p.y_f32[(size_t)row * p.ldy + j] = y;
  }
}
__global__ __launch_bounds__(256) void ln_bwd_small_kernel(LnBwdP p) {
  __shared__ float red[2][64];
  if (threadIdx.x < 128) red[threadIdx.x >> 6][threadIdx.x & 63] = 0.f;
  __syncthreads();
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row < p.rows) {
    const float mu = p.mean[row], rs = p.rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int j = 0; j < p.c; j++) {
      const float xh = (ld1(p.x, p.x_bf16, (size_t)row * p.ldx + j) - mu) * rs;
      float d = ld1(p.dy, p.dy_bf16, (size_t)row * p.lddy + j);
      if (p.act == 1) d *= gelu_grad_f(xh * p.gamma[j] + p.beta[j]);
      const float dxh = d * p.gamma[j];
      s1 += dxh; s2 += dxh * xh;
      atomicAdd(&red[0][j], d * xh);
      atomicAdd(&red[1][j], d);
    }
    s1 /= (float)p.c; s2 /= (float)p.c;
    for (int j = 0; j < p.c; j++) {
      const float xh = (ld1(p.x, p.x_bf16, (size_t)row * p.ldx + j) - mu) * rs;
      float d = ld1(p.dy, p.dy_bf16, (size_t)row * p.lddy + j);
      if (p.act == 1) d *= gelu_grad_f(xh * p.gamma[j] + p.beta[j]);
      float dx = rs * (d * p.gamma[j] - s1 - xh * s2);
      const size_t off = (size_t)row * p.lddx + j;
      if (p.dx_res) dx += p.dx_res[off];
      if (p.dx_f32) p.dx_f32[off] = dx;
      if (p.dx_bf16) p.dx_bf16[off] = f2bf(dx);
    }
  }
  __syncthreads();
  if (threadIdx.x < p.c) {
    if (p.dgamma) unsafeAtomicAdd(p.dgamma + threadIdx.x, red[0][threadIdx.x]);
    if (p.dbeta) unsafeAtomicAdd(p.dbeta + threadIdx.x, red[1][threadIdx.x]);
  }
}

int pick_lpr(int c) {
  int nch = c >> 2, l = 1;
  while (l < nch && l < 64) l <<= 1;
  return l;
}

template <typename P, typename F>
int dispatch_cpl(int cpl, F&& f) {
  if (cpl <= 1) return f(std::integral_constant<int, 1>{});
  if (cpl <= 2) return f(std::integral_constant<int, 2>{});
  if (cpl <= 3) return f(std::integral_constant<int, 3>{});
  if (cpl <= 4) return f(std::integral_constant<int, 4>{});
  if (cpl <= 5) return f(std::integral_constant<int, 5>{});
  if (cpl <= 8) return f(std::integral_constant<int, 8>{});
  if (cpl <= 12) return f(std::integral_constant<int, 12>{});
  if (cpl <= 16) return f(std::integral_constant<int, 16>{});
  return CINEMA_ERR_UNSUPPORTED;
}

}  // namespace

static int layernorm_fwd_impl(const void* x, int x_is_bf16, int ldx, const float* gamma, const float* beta, int rows, int c, float eps, int act, uint16_t* y_bf16,
                              float* y_f32, int ldy, float* mean, float* rstd, uint8_t* y_fp8, float* row_scale, void* stream, const cinema_q8_out* q8 = nullptr);
CINEMA_API int cinema_layernorm_fwd(const void* x, int x_is_bf16, int ldx, const float* gamma, const float* beta, int rows, int c, float eps,
                                    int act, uint16_t* y_bf16, float* y_f32, int ldy, float* mean, float* rstd, void* stream) {
  return layernorm_fwd_impl(x, x_is_bf16, ldx, gamma, beta, rows, c, eps, act, y_bf16, y_f32, ldy, mean, rstd, nullptr, nullptr, stream);
}
// the same, also writing an e4m3 copy of y ([rows][c] bytes, dense) with one dequantisation scale per row (fp8 forward GEMMs)
CINEMA_API int cinema_layernorm_fwd_fp8(const void* x, int x_is_bf16, int ldx, const float* gamma, const float* beta, int rows, int c, float eps,
                                        int act, uint16_t* y_bf16, float* y_f32, int ldy, float* mean, float* rstd, uint8_t* y_fp8, float* row_scale, void* stream) {
  if (!y_fp8 || !row_scale || (c & 3)) return CINEMA_ERR_BAD_ARG;
  return layernorm_fwd_impl(x, x_is_bf16, ldx, gamma, beta, rows, c, eps, act, y_bf16, y_f32, ldy, mean, rstd, y_fp8, row_scale, stream);
}
CINEMA_API int cinema_layernorm_fwd_q8(const void* x, int x_is_bf16, int ldx, const float* gamma, const float* beta, int rows, int c, float eps, int act,
                                       uint16_t* y_bf16, float* y_f32, int ldy, float* mean, float* rstd, const cinema_q8_out* q8, void* stream) {
  if (!q8 || !q8->amax_slots || (q8->data && !q8->inv_scale) || (c & 3)) return CINEMA_ERR_BAD_ARG;
  return layernorm_fwd_impl(x, x_is_bf16, ldx, gamma, beta, rows, c, eps, act, y_bf16, y_f32, ldy, mean, rstd, q8->data, nullptr, stream, q8);
}
static int layernorm_fwd_impl(const void* x, int x_is_bf16, int ldx, const float* gamma, const float* beta, int rows, int c, float eps, int act, uint16_t* y_bf16,
                              float* y_f32, int ldy, float* mean, float* rstd, uint8_t* y_fp8, float* row_scale, void* stream, const cinema_q8_out* q8) {
  if (!x || !gamma || !beta || rows <= 0 || c <= 0 || (!y_bf16 && !y_f32)) return CINEMA_ERR_BAD_ARG;
  LnFwdP p{x, x_is_bf16, ldx, gamma, beta, rows, c, eps, act, y_bf16, y_f32, ldy, mean, rstd, pick_lpr(c), y_fp8, row_scale, q8 ? q8->inv_scale : nullptr,
           q8 ? q8->amax_slots : nullptr};
  if ((c & 3) || (ldx & 3) || (ldy & 3)) {
    if (c > 64 || y_fp8 || q8) return CINEMA_ERR_UNSUPPORTED;
    CINEMA_LAUNCH(ln_fwd_small_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status();
  }
  const int cpl = ((c >> 2) + p.lpr - 1) / p.lpr;
  const int rows_per_block = 4 * (64 / p.lpr);
  int grid = (rows + rows_per_block - 1) / rows_per_block;
  if (grid > 8192) grid = 8192;
  return dispatch_cpl<LnFwdP>(cpl, [&](auto tag) {
    launch_lanes(ln_fwd_kernel<decltype(tag)::value>, ln_fwd_lanes_kernel<decltype(tag)::value>, 1, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status();
  });
}

static int layernorm_bwd_impl(const void* dy, int dy_is_bf16, int lddy, const void* x, int x_is_bf16, int ldx, const float* gamma,
                              const float* beta, const float* mean, const float* rstd, int rows, int c, int act,
                              const float* dx_residual, float* dx_f32, uint16_t* dx_bf16, int lddx, float* dgamma, float* dbeta,
                              float* workspace, long long workspace_bytes, int* deferred_partials, void* stream, const cinema_q8_out* q8 = nullptr) {
  if (deferred_partials) *deferred_partials = 0;
  if (!dy || !x || !gamma || !mean || !rstd || rows <= 0 || c <= 0) return CINEMA_ERR_BAD_ARG;
  if (act == 1 && !beta) return CINEMA_ERR_BAD_ARG;
  LnBwdP p{dy, dy_is_bf16, lddy, x, x_is_bf16, ldx, gamma, beta, mean, rstd, rows, c, act, dx_residual, dx_f32, dx_bf16, lddx, dgamma, dbeta,
           pick_lpr(c), nullptr, q8 ? q8->data : nullptr, q8 ? q8->inv_scale : nullptr, q8 ? q8->amax_slots : nullptr, q8 ? q8->colsum : nullptr};
  const int nr = p.dcol ? 3 : 2;
  if ((c & 3) || (ldx & 3) || (lddy & 3) || (lddx & 3)) {
    if (c > 64 || q8) return CINEMA_ERR_UNSUPPORTED;
    CINEMA_LAUNCH(ln_bwd_small_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status();
  }
  const int cpl = ((c >> 2) + p.lpr - 1) / p.lpr;
  const size_t smem = (size_t)4 * nr * c * sizeof(float);
  return dispatch_cpl<LnBwdP>(cpl, [&](auto tag) {
    constexpr int CPL = decltype(tag)::value;
    constexpr int RG = CPL == 1 ? 4 : 1;  // RG = 2 at CPL 2-3 measured 30 % slower (registers)
    const int rows_per_block = 4 * (64 / p.lpr) * RG;
    int grid = (rows + rows_per_block - 1) / rows_per_block;
    const int cap = CPL == 1 ? 2048 : 1024;
    if (grid > cap) grid = cap;
    const bool two_pass = (dgamma || dbeta || p.dcol) && workspace && workspace_bytes >= (long long)grid * nr * c * 4 && grid >= 64;
    if (two_pass) p.ws = workspace;
    if (p.dcol) CINEMA_LAUNCH((ln_bwd_dcol_kernel<CPL, RG>), dim3(grid), dim3(256), smem, (hipStream_t)stream, p);
    else launch_lanes(ln_bwd_kernel<CPL, RG>, ln_bwd_lanes_kernel<CPL, RG>, 1, dim3(grid), dim3(256), smem, (hipStream_t)stream, p);
    if (two_pass && deferred_partials) *deferred_partials = grid;  // the caller reduces `grid` partial rows later (cinema_ln_param_reduce_batched)
    else if (two_pass && p.dcol) return CINEMA_ERR_UNSUPPORTED;  // (column sums of dx: deferred form only)
    else if (two_pass) CINEMA_LAUNCH(ln_param_reduce_kernel, dim3((2 * c + 63) / 64, 16), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, grid, c, dgamma, dbeta);
    return launch_status();
  });
}

CINEMA_API int cinema_layernorm_bwd(const void* dy, int dy_is_bf16, int lddy, const void* x, int x_is_bf16, int ldx, const float* gamma,
                                    const float* beta, const float* mean, const float* rstd, int rows, int c, int act,
                                    const float* dx_residual, float* dx_f32, uint16_t* dx_bf16, int lddx, float* dgamma, float* dbeta,
                                    float* workspace, long long workspace_bytes, void* stream) {
  return layernorm_bwd_impl(dy, dy_is_bf16, lddy, x, x_is_bf16, ldx, gamma, beta, mean, rstd, rows, c, act, dx_residual, dx_f32, dx_bf16, lddx, dgamma, dbeta,
                            workspace, workspace_bytes, nullptr, stream);
}

// Number of per-block partial rows ([2][c] fp32 each) a backward launch over (rows, c) writes: sizes the workspace of the deferred form exactly
// (the same grid computation as layernorm_bwd_impl; 0 = the small-channel kernel, which takes no workspace).
CINEMA_API long long cinema_layernorm_bwd_workspace_bytes(int rows, int c) {
  if (rows <= 0 || c <= 0 || (c & 3)) return 0;
  const int lpr = pick_lpr(c);
  const int cpl = ((c >> 2) + lpr - 1) / lpr;
  const int rg = cpl <= 1 ? 4 : 1;
  const int rows_per_block = 4 * (64 / lpr) * rg;
  int grid = (rows + rows_per_block - 1) / rows_per_block;
  const int cap = cpl <= 1 ? 2048 : 1024;
  if (grid > cap) grid = cap;
  return (long long)grid * 2 * c * 4;
}

CINEMA_API int cinema_layernorm_bwd_deferred(const void* dy, int dy_is_bf16, int lddy, const void* x, int x_is_bf16, int ldx, const float* gamma,
                                             const float* beta, const float* mean, const float* rstd, int rows, int c, int act,
                                             const float* dx_residual, float* dx_f32, uint16_t* dx_bf16, int lddx, float* dgamma, float* dbeta,
                                             float* workspace, long long workspace_bytes, int* n_partials_out, void* stream) {
  if (!n_partials_out) return CINEMA_ERR_BAD_ARG;
  return layernorm_bwd_impl(dy, dy_is_bf16, lddy, x, x_is_bf16, ldx, gamma, beta, mean, rstd, rows, c, act, dx_residual, dx_f32, dx_bf16, lddx, dgamma, dbeta,
                            workspace, workspace_bytes, n_partials_out, stream);
}

CINEMA_API int cinema_layernorm_bwd_deferred_q8(const void* dy, int dy_is_bf16, int lddy, const void* x, int x_is_bf16, int ldx, const float* gamma, const float* beta,
                                                const float* mean, const float* rstd, int rows, int c, int act, const float* dx_residual, float* dx_f32,
                                                uint16_t* dx_bf16, int lddx, float* dgamma, float* dbeta, float* workspace, long long workspace_bytes,
                                                int* n_partials_out, const cinema_q8_out* q8, void* stream) {
  if (!n_partials_out || !q8 || !q8->amax_slots || (q8->data && !q8->inv_scale)) return CINEMA_ERR_BAD_ARG;
  return layernorm_bwd_impl(dy, dy_is_bf16, lddy, x, x_is_bf16, ldx, gamma, beta, mean, rstd, rows, c, act, dx_residual, dx_f32, dx_bf16, lddx, dgamma, dbeta,
                            workspace, workspace_bytes, n_partials_out, stream, q8);
}

namespace {
struct LnReduceBatch { cinema_ln_reduce_item it[48]; };
// the per-block partial sums of up to 48 LayerNorm backward launches in one grid: blockIdx.z = item, same tiling as ln_param_reduce_kernel
__global__ __launch_bounds__(256) void ln_param_reduce_batched_kernel(LnReduceBatch b) {
  __shared__ float red[4][64];
  const cinema_ln_reduce_item& e = b.it[blockIdx.z];
  const int c = e.c, nblocks = e.n_partials;
  const int nr = e.dcol ? 3 : 2;  // rows per partial: d gamma, d beta[, column sums of dx]
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  if (blockIdx.x * 64 >= nr * c) return;
  const int slices = gridDim.y * 4;
  const int per = (nblocks + slices - 1) / slices;
  const int b0 = (blockIdx.y * 4 + sub) * per, b1 = min(nblocks, b0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < nr * c) {
    const float* src = e.partials + col;
    const size_t ld = (size_t)nr * c;
    int r = b0;
    for (; r + 3 < b1; r += 4) { s0 += src[r * ld]; s1 += src[(r + 1) * ld]; s2 += src[(r + 2) * ld]; s3 += src[(r + 3) * ld]; }
    for (; r < b1; r++) s0 += src[r * ld];
  }
  red[sub][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sub == 0 && col < nr * c) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    float* dst = col < c ? e.dgamma : (col < 2 * c ? e.dbeta : e.dcol);
    if (dst) unsafeAtomicAdd(dst + col % c, t);
  }
}
}  // namespace

CINEMA_API int cinema_ln_param_reduce_batched(const cinema_ln_reduce_item* items_host, int count, void* stream) {
  if (!items_host || count <= 0) return CINEMA_ERR_BAD_ARG;
  for (int i0 = 0; i0 < count; i0 += 48) {
    LnReduceBatch b;
    const int n = count - i0 < 48 ? count - i0 : 48;
    int cmax = 0;
    for (int i = 0; i < n; i++) {
      b.it[i] = items_host[i0 + i];
      if (!b.it[i].partials || b.it[i].n_partials <= 0 || b.it[i].c <= 0) return CINEMA_ERR_BAD_ARG;
      if (b.it[i].c > cmax) cmax = b.it[i].c;
    }
    CINEMA_LAUNCH(ln_param_reduce_batched_kernel, dim3((3 * cmax + 63) / 64, 16, n), dim3(256), 0, (hipStream_t)stream, b);
  }
  return launch_status();
}
