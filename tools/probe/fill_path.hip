// Fill-path probe for gfx950 (dev tooling): how many bytes per clock a CU can pull from an L2-resident working set, by path.  The 128x128 GEMM loop was measured
// bound by its global -> LDS stream at ~32 B/clk per CU; this asks whether that is the LDS-DMA path or the L2 -> CU path itself.
//   dma      : global_load_lds_dwordx4 (1 KiB per wave instruction, lane-linear LDS image), two 32 KiB stages in flight per workgroup
//   reg      : global_load_dwordx4 into VGPRs, 8 loads per lane in flight, values xor-ed into an accumulator
//   reg_lds  : reg + ds_write_b128 of every loaded value (the classic register-staged tile fill)
//   reg_half : reg, but each lane's 16 bytes sit in 32-byte row pieces (the direct-to-fragment pattern of a k-major MFMA operand: lane l -> row l & 31, 16 B at 16 (l >> 5))
// Working set: 8 regions of REGION bytes per XCD (workgroup w runs on XCD w % 8 and reads region (w / 8) % 8 of that XCD's share), every region shared by the
// workgroups of one XCD that map to it, as the 8 x 8 tile patch of the GEMM does.  Build: hipcc --offload-arch=gfx950 -O2 fill_path.hip -o fill_path
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int REGION = 384 * 1024;  // one A + one B panel of a 128 x 128 x 768 tile
constexpr int STEP = 32 * 1024;     // bytes per workgroup per iteration (one 128 x 64 A tile + one B tile)
constexpr int NSTEP = REGION / STEP;

__device__ __forceinline__ void glds16(uint32_t lds_addr, const void* gsrc) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ const char* region_of(const char* buf) {
  const int w = blockIdx.x, xcd = w & 7, r = (w >> 3) & 7;
  return buf + ((size_t)(xcd * 8 + r)) * REGION;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_dma(const char* buf, int reps, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const char* reg = region_of(buf);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr int PER_WAVE = STEP / WAVES / 1024;  // 1 KiB pieces per wave per step
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)lds);
  int it = 0;
  for (int rep = 0; rep < reps; rep++)
    for (int s = 0; s < NSTEP; s++, it++) {
      const uint32_t stage = lds0 + (it & 1) * STEP;
#pragma unroll
      for (int i = 0; i < PER_WAVE; i++) {
        const int piece = wave * PER_WAVE + i;
        glds16(stage + piece * 1024, reg + (size_t)s * STEP + piece * 1024 + lane * 16);
      }
      if (PER_WAVE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && reps < 0) sink[0] = lds[5];
}

template <int WAVES, int MODE>  // MODE 0 reg, 1 reg + ds_write_b128, 2 reg with 32-byte row pieces
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_reg(const char* buf, int reps, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const char* reg = region_of(buf);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int PER_WAVE = STEP / WAVES / 1024;
  uint4 acc = make_uint4(0, 0, 0, 0);
  uint4 va[PER_WAVE], vb[PER_WAVE];
  // MODE 2: a piece = 32 rows of 128 B (a k-major [128 rows][64 k] bf16 tile); load i takes bytes [32 j, 32 j + 32) of each row, j = i & 3
  const int lane_off = MODE == 2 ? (lane & 31) * 128 + (lane >> 5) * 16 : lane * 16;
  const char* base = reg + wave * PER_WAVE * 1024 + lane_off;
#define ADDR(s, i) reinterpret_cast<const uint4*>(base + (size_t)(s) * STEP + (MODE == 2 ? ((i) >> 2) * 4096 + ((i) & 3) * 32 : (i) * 1024))
#define USE(v, stage)                                                                                                        \
  _Pragma("unroll") for (int i = 0; i < PER_WAVE; i++) {                                                                     \
    if (MODE == 1) *reinterpret_cast<uint4*>(lds + (stage) * STEP + (wave * PER_WAVE + i) * 1024 + lane * 16) = v[i];        \
    else { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }                                            \
  }
#pragma unroll
  for (int i = 0; i < PER_WAVE; i++) va[i] = *ADDR(0, i);
  for (int rep = 0; rep < reps; rep++)
#pragma unroll 1
    for (int s = 0; s < NSTEP; s += 2) {
#pragma unroll
      for (int i = 0; i < PER_WAVE; i++) vb[i] = *ADDR(s + 1, i);
      USE(va, 0)
      const int sn = (s + 2 == NSTEP) ? 0 : s + 2;
#pragma unroll
      for (int i = 0; i < PER_WAVE; i++) va[i] = *ADDR(sn, i);
      USE(vb, 1)
    }
  __syncthreads();
  if (MODE == 1) acc = *reinterpret_cast<uint4*>(lds + threadIdx.x * 16);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345 && reps < 0) sink[0] = 1;
}

template <typename F>
static void run(const char* name, F launch, int wgs, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(2);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int t = 0; t < 5; t++) {
    CK(hipEventRecord(e0));
    launch(reps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double bytes = (double)wgs * reps * REGION;
  const double tbs = bytes / (best * 1e-3) / 1e12;
  printf("%-28s %4d wgs  %8.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz, 256 CUs)\n", name, wgs, best * 1e3, tbs, tbs * 1e12 / 256 / 2.4e9);
}

int main() {
  char* buf; int* sink;
  const size_t total = (size_t)64 * REGION;
  CK(hipMalloc(&buf, total)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, total));
  const int reps = 60;
  CK(hipFuncSetAttribute((const void*)k_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  CK(hipFuncSetAttribute((const void*)k_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  CK(hipFuncSetAttribute((const void*)k_reg<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  for (int wgs : {256, 512, 1024}) {
    run("dma, 4 waves / wg", [&](int r) { hipLaunchKernelGGL(k_dma<4>, dim3(wgs), dim3(256), 64 * 1024, 0, buf, r, sink); }, wgs, reps);
    run("dma, 8 waves / wg", [&](int r) { hipLaunchKernelGGL(k_dma<8>, dim3(wgs), dim3(512), 64 * 1024, 0, buf, r, sink); }, wgs, reps);
    run("reg, 4 waves / wg", [&](int r) { hipLaunchKernelGGL((k_reg<4, 0>), dim3(wgs), dim3(256), 64 * 1024, 0, buf, r, sink); }, wgs, reps);
    run("reg, 8 waves / wg", [&](int r) { hipLaunchKernelGGL((k_reg<8, 0>), dim3(wgs), dim3(512), 64 * 1024, 0, buf, r, sink); }, wgs, reps);
    run("reg + ds_write, 4 waves / wg", [&](int r) { hipLaunchKernelGGL((k_reg<4, 1>), dim3(wgs), dim3(256), 64 * 1024, 0, buf, r, sink); }, wgs, reps);
    run("reg 32-B pieces, 4 waves", [&](int r) { hipLaunchKernelGGL((k_reg<4, 2>), dim3(wgs), dim3(256), 64 * 1024, 0, buf, r, sink); }, wgs, reps);
  }
  return 0;
}
