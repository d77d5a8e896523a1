// Fill-rate probe 2 for gfx950 (dev tooling): LDS-DMA throughput of a CU when the operand stream does NOT sit in L2 - the regime of the GEMM loops (the persistent
// 256x256 kernel stages 22-30 B/clk per CU against 56 B/clk for the same instruction on an L2-resident set, tools/probe/fill_path.hip) - and whether a run-ahead L2 prefetch changes it.
// Work: 256 workgroups of 8 DMA waves (one per CU), workgroup w runs on XCD w % 8 and streams region (w / 8) % NREG of that XCD's share from begin to end, 32 KiB per step
// (= one phase of the 256x256 loop: 4 pieces of 1 KiB per wave, counted vmcnt, two steps in flight); the 32 / NREG workgroups of an XCD that map to one region read the SAME bytes in
// lockstep, as the tiles of one panel do.  Regions are far larger than L2 (4 MiB per XCD), the whole set fits the 256 MiB MALL or not (argument).
//   pf 0: no prefetch
//   pf 1: a ninth wave per workgroup touches one dword of every 128-byte line of the step D ahead (4 wave-instructions per step), every sharer the whole step
//   pf 2: the same, but the sharers split the lines (sharer j takes lines j, j + S, ...: one wave-instruction per step when S = 4)
// Build: hipcc --offload-arch=gfx950 -O2 fill_stream.hip -o fill_stream ; ./fill_stream [region MiB] [nreg] [D]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int STEP = 32 * 1024;

__device__ __forceinline__ void glds16(uint32_t lds_addr, const void* gsrc) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

template <int PF>
__global__ __launch_bounds__(PF ? 576 : 512) void k_stream(const char* buf, long long region_bytes, int nreg, int dist, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int w = blockIdx.x, xcd = w & 7, r = (w >> 3) % nreg, sharer = (w >> 3) / nreg, nshare = 32 / nreg;
  const char* reg = buf + ((size_t)(xcd * nreg + r)) * region_bytes;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nstep = (int)(region_bytes / STEP);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)lds);
  if (wave < 8) {
    for (int s = 0; s < nstep; s++) {
      const uint32_t stage = lds0 + (s & 3) * STEP;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int piece = wave * 4 + i;
        glds16(stage + piece * 1024, reg + (size_t)s * STEP + piece * 1024 + lane * 16);
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (PF) __syncthreads();  // paces the prefetch wave
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    int acc = 0;
    for (int s = 0; s < nstep; s++) {
      const int t = s + dist;
      if (t < nstep) {
        if (PF == 1) {
#pragma unroll
          for (int i = 0; i < 4; i++) asm volatile("global_load_dword %0, %1, off" : "+v"(acc) : "v"(reg + (size_t)t * STEP + (i * 64 + lane) * 128) : "memory");
        } else {
          for (int l = sharer * 64 + lane; l < 256; l += nshare * 64) asm volatile("global_load_dword %0, %1, off" : "+v"(acc) : "v"(reg + (size_t)t * STEP + l * 128) : "memory");
        }
      }
      __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678) sink[0] = acc;
  }
  if (threadIdx.x == 0 && region_bytes < 0) sink[1] = lds[5];
}

template <typename F>
static void run(const char* name, F launch, double bytes) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int t = 0; t < 5; t++) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double tbs = bytes / (best * 1e-3) / 1e12;
  printf("%-44s %9.1f us  %6.2f TB/s staged  %5.1f B/clk/CU (2.4 GHz, 256 CUs)\n", name, best * 1e3, tbs, tbs * 1e12 / 256 / 2.4e9);
}

int main(int argc, char** argv) {
  const long long region = (argc > 1 ? atoll(argv[1]) : 4) << 20;
  const int nreg = argc > 2 ? atoi(argv[2]) : 8;
  const int dist = argc > 3 ? atoi(argv[3]) : 4;
  char* buf; int* sink;
  const size_t total = (size_t)8 * nreg * region;
  CK(hipMalloc(&buf, total)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, total));
  printf("regions of %lld MiB, %d per XCD (%d workgroups share one), %.0f MiB in all, prefetch distance %d steps\n", region >> 20, nreg, 32 / nreg, total / 1048576.0, dist);
  CK(hipFuncSetAttribute((const void*)k_stream<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  CK(hipFuncSetAttribute((const void*)k_stream<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  CK(hipFuncSetAttribute((const void*)k_stream<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  const double bytes = 256.0 * region;
  run("LDS-DMA stream, no prefetch", [&]() { hipLaunchKernelGGL(k_stream<0>, dim3(256), dim3(512), 128 * 1024, 0, buf, region, nreg, dist, sink); }, bytes);
  run("  + L2 prefetch, every sharer every line", [&]() { hipLaunchKernelGGL(k_stream<1>, dim3(256), dim3(576), 128 * 1024, 0, buf, region, nreg, dist, sink); }, bytes);
  run("  + L2 prefetch, sharers split the lines", [&]() { hipLaunchKernelGGL(k_stream<2>, dim3(256), dim3(576), 128 * 1024, 0, buf, region, nreg, dist, sink); }, bytes);
  return 0;
}
