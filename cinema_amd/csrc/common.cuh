// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the CineMA MAE hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CINEMA_API extern "C" __attribute__((visibility("default")))

// Error codes returned by every C-ABI entry point (0 = success; >0 = hipError_t of the launch).
#define CINEMA_ERR_BAD_ARG (-1)
#define CINEMA_ERR_UNSUPPORTED (-2)

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) short short8v;
typedef __attribute__((ext_vector_type(4))) float float4v;
typedef __attribute__((ext_vector_type(16))) float float16v;

static inline int launch_status() { return (int)hipGetLastError(); }

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even, NaN preserved as quiet NaN
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }

// exact (erf) GELU and its derivative, as nn.GELU() (reference cinema/conv.py:271-272, timm Mlp)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// wave64 reductions (all 64 lanes participate)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// 16-byte vector of 8 bf16
struct __attribute__((aligned(16))) bf16x8 { bf16_t v[8]; };
struct __attribute__((aligned(8))) bf16x4 { bf16_t v[4]; };

// Byte offset of 16-byte chunk `c` of row `r` in a K-major LDS tile whose rows are RB bytes (64 or 128).
// XOR swizzle so that a ds_read_b128 lane group (16 distinct rows, same chunk) touches 16 distinct
// 16-byte slots of the 256-byte bank row (cdna guide T2).
template <int RB>
__device__ __forceinline__ int swz_off(int r, int c) {
  constexpr int ROWS_PER_BANKROW = 256 / RB;
  constexpr int CHUNKS = RB / 16;
  return r * RB + ((c ^ ((r / ROWS_PER_BANKROW) & (CHUNKS - 1))) << 4);
}

__device__ __forceinline__ short4v lds_tr16_b64(const void* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(lds_ptr));
}
