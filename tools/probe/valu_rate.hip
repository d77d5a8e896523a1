// dev probe: issue rate of scalar vs packed fp32 FMA, v_exp_f32, v_cvt_pk_bf16_f32 and their overlap with MFMA on one SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef short short8v __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float b = 1.0001f, c = 1e-4f;
  float16v acc = {0};
  short8v fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {  // 8 scalar FMAs
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (MODE == 1) {  // 4 packed FMAs (the same 8 results)
      float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, bb = {b, b}, cc = {c, c};
      asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb), "v"(cc));
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    } else if (MODE == 2) {  // 8 v_exp_f32
      asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 3) {  // 1 MFMA 32x32x16
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    } else if (MODE == 4) {  // 1 MFMA + 8 scalar FMAs (independent)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc[0];
}
// inter-wave overlap on one SIMD: waves 0-3 (one per SIMD) run 4 independent MFMAs per iteration, waves 4-7 run 32 independent-chain FMAs per iteration
// (each alone: ~128 clk per iteration); which = 1: MFMA waves only, 2: VALU waves only, 3: both
__global__ __launch_bounds__(512) void mixed(float* out, int iters, int which) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float b = 1.0001f, c = 1e-4f;
  float16v acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  short8v fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (w < 4) {
    if (which & 1)
      for (int i = 0; i < iters; i++) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc3, 0, 0, 0);
      }
  } else if (which & 2) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int r = 0; r < 4; r++)
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc1[0] + acc2[0] + acc3[0];
}
template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 512 * 4);
  const int iters = 20000, blocks = 1024;  // 4 blocks of 4 waves per CU: one wave per SIMD per block -> 4 waves per SIMD
  const char* names[] = {"8 x v_fma_f32", "4 x v_pk_fma_f32", "8 x v_exp_f32", "1 x mfma 32x32x16", "1 mfma + 8 v_fma (same wave)"};
  float t[5];
  t[0] = timeit([&] { k<0><<<blocks, 256>>>(out, iters); });
  t[1] = timeit([&] { k<1><<<blocks, 256>>>(out, iters); });
  t[2] = timeit([&] { k<2><<<blocks, 256>>>(out, iters); });
  t[3] = timeit([&] { k<3><<<blocks, 256>>>(out, iters); });
  t[4] = timeit([&] { k<4><<<blocks, 256>>>(out, iters); });
  // waves per SIMD = blocks * 4 waves / (256 CUs * 4 SIMDs) = 4; per-SIMD cycles per loop iteration of ONE wave at 2.4 GHz:
  for (int i = 0; i < 5; i++) printf("%-30s %8.3f ms  -> %6.1f clk per iteration per SIMD-wave (4 waves/SIMD, 2.4 GHz)\n", names[i], t[i], t[i] * 1e-3 * 2.4e9 / iters / 4);
  float t2m = timeit([&] { k<3><<<512, 256>>>(out, iters); });
  float t2v = timeit([&] { k<0><<<512, 256>>>(out, iters); });
  printf("2 MFMA waves per SIMD: %.1f clk per iteration; 2 VALU waves per SIMD: %.1f clk per iteration\n", t2m * 1e-3 * 2.4e9 / iters, t2v * 1e-3 * 2.4e9 / iters);
  for (int which = 1; which <= 3; which++) {
    float tm = timeit([&] { mixed<<<256, 512>>>(out, iters, which); });
    printf("one MFMA wave (4 mfma/iter) and/or one VALU wave (32 v_fma/iter) per SIMD, which=%d: %6.1f clk per iteration\n", which, tm * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
