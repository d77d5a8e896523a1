// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the CineMA MAE hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
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

// ---- lane groups (stream.hip): independent, identically shaped launch sequences zipped into merged launches -------------------------------------
// The three long-axis views of an MAE step run the same ~45 (forward) / ~95 (backward) tiny stem kernels on different pointers: 550 launches of 5-10 us
// whose cost is launch latency, not work.  Between cinema_lanes_begin(n) and cinema_lanes_end() the launches of the library are not issued but
// recorded per lane (cinema_lanes_select(i) switches the lane); at the end the n sequences are walked in lock step and every position whose n
// launches are the same kernel with the same launch geometry becomes ONE launch of the kernel's *_lanes form (grid x n in the dimension the
// kernel does not use; block (.., lane, ..) reads parameter block `lane`).  Positions that do not match (or kernels without a lanes form) are
// issued one by one in lane order - each lane keeps its own order, and lanes are independent by contract, so any interleaving is valid.
constexpr int MAX_LANES = 4;
template <typename P>
struct Lanes { P p[MAX_LANES]; };
bool lanes_active();
// kernels actually handed to the HIP runtime by this library (cinema_kernel_launch_count: a merged lane-group launch counts once, stream forks not at all)
inline std::atomic<long long> g_kernel_launches{0};
// Record (inside a lane group) or issue (outside) one launch.  `relaunch(fn_single, grid, block, smem, stream, params)` issues the single form from the
// stored parameter bytes; fn_lanes (may be NULL) is the merged form taking Lanes<P> when the parameter bytes are ONE struct P.
typedef int (*lane_relaunch_t)(const void* fn, dim3 grid, dim3 block, size_t smem, hipStream_t st, const void* params);
int lane_submit(lane_relaunch_t relaunch, const void* fn_single, const void* fn_lanes, int lane_dim, dim3 grid, dim3 block, size_t smem, hipStream_t st,
                const void* params, size_t psize);
inline int lane_relaunch_struct(const void* fn, dim3 grid, dim3 block, size_t smem, hipStream_t st, const void* params) {
  void* args[] = {const_cast<void*>(params)};
  g_kernel_launches.fetch_add(1, std::memory_order_relaxed);
  return (int)hipLaunchKernel(fn, grid, block, args, smem, st);
}
// a kernel with ONE struct parameter and a lanes form: lane_dim 1 = blockIdx.y, 2 = blockIdx.z selects the lane (that grid dimension must be 1)
template <typename P>
inline void launch_lanes(void (*single)(P), void (*lanes)(Lanes<P>), int lane_dim, dim3 g, dim3 b, size_t smem, hipStream_t st, const P& p) {
  (void)lane_submit(lane_relaunch_struct, (const void*)single, (const void*)lanes, lane_dim, g, b, smem, st, &p, sizeof(P));
}
// any other kernel: the arguments are packed into a POD tuple so that the launch can be replayed in order at the end of a lane group
template <typename... A> struct ArgPack;
template <> struct ArgPack<> {};
template <typename H, typename... T> struct ArgPack<H, T...> { H h; ArgPack<T...> t; };
template <typename... KA, typename... Done>
inline void argpack_launch(void (*k)(KA...), dim3 g, dim3 b, size_t smem, hipStream_t st, const ArgPack<>&, Done... d) {
  g_kernel_launches.fetch_add(1, std::memory_order_relaxed);
  hipLaunchKernelGGL(k, g, b, smem, st, d...);
}
template <typename... KA, typename H, typename... T, typename... Done>
inline void argpack_launch(void (*k)(KA...), dim3 g, dim3 b, size_t smem, hipStream_t st, const ArgPack<H, T...>& p, Done... d) {
  argpack_launch(k, g, b, smem, st, p.t, d..., p.h);
}
template <typename... KA>
inline int lane_relaunch_pack(const void* fn, dim3 grid, dim3 block, size_t smem, hipStream_t st, const void* params) {
  argpack_launch(reinterpret_cast<void (*)(KA...)>(const_cast<void*>(fn)), grid, block, smem, st, *reinterpret_cast<const ArgPack<KA...>*>(params));
  return (int)hipPeekAtLastError();
}
template <typename... KA> inline void argpack_fill(ArgPack<KA...>&) {}
template <typename H, typename... T, typename A0, typename... A>
inline void argpack_fill(ArgPack<H, T...>& p, A0&& a0, A&&... a) { p.h = static_cast<H>(a0); argpack_fill(p.t, static_cast<A&&>(a)...); }
template <typename... KA, typename... A>
inline void launch_any(void (*kernel)(KA...), dim3 g, dim3 b, size_t smem, hipStream_t st, A&&... a) {
  static_assert(sizeof...(KA) == sizeof...(A), "argument count");
  if (!lanes_active()) {
    g_kernel_launches.fetch_add(1, std::memory_order_relaxed);
    hipLaunchKernelGGL(kernel, g, b, smem, st, static_cast<KA>(a)...);
    return;
  }
  ArgPack<KA...> pk;
  argpack_fill(pk, static_cast<A&&>(a)...);
  (void)lane_submit(lane_relaunch_pack<KA...>, (const void*)kernel, nullptr, 0, g, b, smem, st, &pk, sizeof(pk));
}
#define CINEMA_LAUNCH(kernel, grid, block, smem, st, ...) launch_any(kernel, grid, block, smem, st, __VA_ARGS__)

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16, round-to-nearest-even, on the gfx950 hardware converter (v_cvt_pk_bf16_f32: one instruction per PAIR;
// a bit-twiddled software rounding costs ~8 VALU instructions per element and showed up in every epilogue)
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const hw_f32x2 f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, hw_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// raw v_exp_f32 (exp2f() adds ~4 range-handling instructions per call)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// erf GELU (nn.GELU(), reference cinema/conv.py:271-272, timm Mlp) and its derivative.
// libm's erff costs ~30 VALU instructions with branches and made the fused fc1 / fc2-dgrad epilogues VALU-bound
// (dec fc1 211 us in-step vs 134 us without the activation).  Phi(x) is evaluated with Abramowitz-Stegun 7.1.26,
//   erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2),  t = 1/(1 + p z),  |error| <= 1.5e-7  (z = |x|/sqrt(2)),
// i.e. one exp (shared with the Gaussian pdf of the derivative), one reciprocal and a degree-5 Horner chain; the absolute
// error is 4 orders of magnitude below the bf16 rounding of the result (this is NOT the tanh approximation).
__device__ __forceinline__ void gelu_terms(float x, float& cdf, float& gauss) {
  const float ax = fabsf(x);
  gauss = __expf(-0.5f * x * x);  // exp(-z^2)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));  // v_rcp_f32 (1 ulp): __frcp_rn expands to the 11-instruction IEEE division sequence, and the epilogue GELU is VALU-bound (72 clk per element measured)
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float half_erfc = 0.5f * poly * t * gauss;  // 0.5 * (1 - erf(|x|/sqrt2))
  cdf = x >= 0.f ? 1.0f - half_erfc : half_erfc;
}
__device__ __forceinline__ float gelu_f(float x) {
  float cdf, g;
  gelu_terms(x, cdf, g);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf, g;
  gelu_terms(x, cdf, g);
  return fmaf(x * 0.39894228040143268f, g, cdf);
}

// Two elements at a time on packed fp32 (v_pk_fma_f32 / v_pk_mul_f32): the fused GEMM epilogues are VALU-bound on these terms.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_terms2(f32x2 x, f32x2& cdf, f32x2& gauss) {
  f32x2 ax;
  ax.x = fabsf(x.x); ax.y = fabsf(x.y);
  const f32x2 e = x * x * (-0.5f * 1.4426950408889634f);
  gauss.x = __builtin_amdgcn_exp2f(e.x); gauss.y = __builtin_amdgcn_exp2f(e.y);
  const f32x2 d = ax * (0.3275911f * 0.70710678118654752f) + 1.0f;
  f32x2 t;
  t.x = __builtin_amdgcn_rcpf(d.x); t.y = __builtin_amdgcn_rcpf(d.y);
  f32x2 poly = t * 1.061405429f + (-1.453152027f);
  poly = poly * t + 1.421413741f;
  poly = poly * t + (-0.284496736f);
  poly = poly * t + 0.254829592f;
  const f32x2 half_erfc = poly * t * gauss * 0.5f;
  cdf.x = x.x >= 0.f ? 1.0f - half_erfc.x : half_erfc.x;
  cdf.y = x.y >= 0.f ? 1.0f - half_erfc.y : half_erfc.y;
}
__device__ __forceinline__ void gelu2(float& a, float& b) {
  f32x2 x = {a, b}, cdf, g;
  gelu_terms2(x, cdf, g);
  x = x * cdf;
  a = x.x; b = x.y;
}
// (a, b) <- gelu(a), gelu(b) and (da, db) = gelu'(a), gelu'(b) from one evaluation of the erf terms
__device__ __forceinline__ void gelu_both2(float& a, float& b, float& da, float& db) {
  f32x2 x = {a, b}, cdf, g;
  gelu_terms2(x, cdf, g);
  const f32x2 d = x * 0.39894228040143268f * g + cdf;
  x = x * cdf;
  a = x.x; b = x.y; da = d.x; db = d.y;
}
// (ga, gb) = gelu'(a), gelu'(b)
__device__ __forceinline__ void gelu_grad2(float a, float b, float& ga, float& gb) {
  f32x2 x = {a, b}, cdf, g;
  gelu_terms2(x, cdf, g);
  const f32x2 r = x * 0.39894228040143268f * g + cdf;
  ga = r.x; gb = r.y;
}

// GELU'(x) of the exact (erf) GELU lies in [-0.1290, 1.1290]; the 8-bit code of the derivative tensor (GemmP::gelu_deriv == 2) is the affine map of
// [-0.13, 1.13] onto 0..255: step 0.00494, error <= 0.00247 - the size of a bf16 rounding error at |GELU'| >= 0.5 (bf16 ulp 2^-8 there), i.e. where the
// derivative matters; half the bytes of the bf16 tensor fc1 writes and fc2's data gradient reads back (the K = 512 / 768 GEMMs around it run on the HBM side
// of the ridge).  Uncorrelated with the gradient it multiplies: the noise averages out in every sum downstream.
constexpr float GELU8_LO = -0.13f, GELU8_STEP = 1.26f / 255.0f, GELU8_INV = 255.0f / 1.26f;
__device__ __forceinline__ uint32_t gelu8_pack4(float a, float b, float c, float d) {   // v_cvt_pk_u8_f32: round to nearest, saturate, byte select
  uint32_t r = 0;
  r = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(a, GELU8_INV, -GELU8_LO * GELU8_INV), 0, r);
  r = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(b, GELU8_INV, -GELU8_LO * GELU8_INV), 1, r);
  r = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(c, GELU8_INV, -GELU8_LO * GELU8_INV), 2, r);
  r = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(d, GELU8_INV, -GELU8_LO * GELU8_INV), 3, r);
  return r;
}
__device__ __forceinline__ float gelu8_dec(uint32_t word, int byte) { return fmaf((float)((word >> (8 * byte)) & 0xffu), GELU8_STEP, GELU8_LO); }  // (v_cvt_f32_ubyteN)

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

// One LDS-DMA instruction (16 B per lane -> 1 KiB lane-linear at the wave-uniform LDS byte address `lds_addr`), issued
// from inline asm so that hipcc does NOT see a pending LDS write: with the builtin it inserts s_waitcnt vmcnt(0) in front
// of the first ds_read_b64_tr_b16 of every k-step (observed in the .s of the dgrad/wgrad variants), which serialises
// the pipeline.  The loop below counts vmcnt by hand instead.  M0 is saved/restored inside the statement (cdna guide 5.7).
__device__ __forceinline__ void glds16(uint32_t lds_addr, const void* gsrc) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
}
// four LDS-DMA pieces in one statement: M0 is saved / restored once and the loads are issued back to back
__device__ __forceinline__ void glds16x4(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, const void* g0, const void* g1, const void* g2, const void* g3) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
      "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
      "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(a0), "s"(a1), "s"(a2), "s"(a3)
      : "memory");
}
__device__ __forceinline__ float bf_lo16(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi16(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
// ---- 8-bit output copies with per-tensor delayed scaling (cinema_q8_out): four values -> four e4m3 bytes, saturated (the scale comes from the previous step's
// maximum, so this step's values may exceed it); the wave's maximum goes into one of Q8_SLOTS slots (device-scope atomics on ONE address serialise: 64 slots still cost 20-40 us per launch of ~14 k waves)
__device__ __forceinline__ int q8_pack4(float a, float b, float c, float d, float inv) {
  const float lim = 448.0f;
  a = fminf(fmaxf(a * inv, -lim), lim); b = fminf(fmaxf(b * inv, -lim), lim); c = fminf(fmaxf(c * inv, -lim), lim); d = fminf(fmaxf(d * inv, -lim), lim);
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return w;
}
constexpr int Q8_SLOTS = 4096;  // = CINEMA_Q8_SLOTS of include/cinema_hip.h
__device__ __forceinline__ void q8_amax_commit(unsigned int* slots, float lane_amax, int slot) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lane_amax = fmaxf(lane_amax, __shfl_xor(lane_amax, o, 64));
  // non-negative floats order like their bit patterns.  A wave first LOOKS at its slot (one L2 load) and only raises it when it has to
  if ((threadIdx.x & 63) == 0 && lane_amax > 0.f) {
    unsigned int* s = slots + (slot & (Q8_SLOTS - 1));
    const unsigned int bits = __float_as_uint(lane_amax);
    if (__hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bits) atomicMax(s, bits);
  }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: set it once per (kernel instantiation, device ordinal), not once per process
// (a second device of the same process would otherwise launch with more than 64 KiB of dynamic LDS without it and fail).  `flags` = one static array per call site.
inline hipError_t dyn_lds_attr_once(bool (&flags)[16], const void* fn, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (flags[dev]) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) flags[dev] = true;
  return e;
}

__device__ __forceinline__ uint32_t lds_address(const void* p) { return (uint32_t)(uintptr_t)p; }  // low 32 bits of a flat LDS pointer = LDS byte address

// Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with a private 4 MiB L2.  xcd_remap() is a bijection
// of [0, n) that hands XCD x the CONTIGUOUS logical range [~x*n/8, ~(x+1)*n/8), so that neighbouring work items (shared
// operand rows, stencil halos) meet in one L2.  Placement is a speed hint only, never a correctness assumption.
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
  return x * q + min(x, r) + i;
}
