// Stream-ordering helpers of the C-ABI (see include/cinema_hip.h): the weight-gradient GEMMs of the backward pass run on a second
// HIP stream, one fork (event record + wait) per launch, ~200 per step.  The events come from a per-device ring created once, so a
// fork is two HIP calls and no allocation; a host that drives the library from Python would otherwise pay an event object, a
// stream-context switch and three interpreter round trips per launch, which made the step launch-bound on slower hosts.
#include <hip/hip_runtime.h>

#include <mutex>

#include "common.cuh"

namespace {

constexpr int MAX_DEVICES = 16;
constexpr int RING = 4096;

struct DeviceRing {
  hipEvent_t fork_events[RING];
  hipEvent_t markers[RING];
  bool ready = false;
  unsigned next_fork = 0;
  long long next_marker = 0;
};

DeviceRing g_rings[MAX_DEVICES];
std::mutex g_mutex;

DeviceRing* ring_of_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
  DeviceRing* r = &g_rings[dev];
  if (!r->ready) {
    for (int i = 0; i < RING; ++i) {
      if (hipEventCreateWithFlags(&r->fork_events[i], hipEventDisableTiming) != hipSuccess) return nullptr;
      if (hipEventCreateWithFlags(&r->markers[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    r->ready = true;
  }
  return r;
}

}  // namespace

CINEMA_API int cinema_stream_fork(void* from_stream, void* to_stream) {
  if (from_stream == to_stream) return 0;
  std::lock_guard<std::mutex> lock(g_mutex);
  DeviceRing* r = ring_of_current_device();
  if (!r) return CINEMA_ERR_UNSUPPORTED;
  hipEvent_t ev = r->fork_events[r->next_fork++ % RING];
  hipError_t e = hipEventRecord(ev, (hipStream_t)from_stream);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)to_stream, ev, 0);
  return (int)e;
}

CINEMA_API long long cinema_marker_record(void* stream) {
  std::lock_guard<std::mutex> lock(g_mutex);
  DeviceRing* r = ring_of_current_device();
  if (!r) return CINEMA_ERR_UNSUPPORTED;
  const long long ticket = r->next_marker++;
  if (hipEventRecord(r->markers[ticket % RING], (hipStream_t)stream) != hipSuccess) return CINEMA_ERR_BAD_ARG;
  return ticket;
}

CINEMA_API int cinema_marker_done(long long ticket) {
  std::lock_guard<std::mutex> lock(g_mutex);
  DeviceRing* r = ring_of_current_device();
  if (!r || ticket < 0 || ticket >= r->next_marker) return CINEMA_ERR_BAD_ARG;
  if (r->next_marker - ticket > RING) return CINEMA_ERR_UNSUPPORTED;  // its ring slot has been recorded again: the caller held too many tickets
  const hipError_t e = hipEventQuery(r->markers[ticket % RING]);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  return CINEMA_ERR_BAD_ARG;
}

// Host-side cost of a kernel launch on this machine: n back-to-back launches of an empty kernel from one C loop (tools/launch_rate.py).
namespace {
__global__ void empty_kernel(int) {}
}  // namespace

CINEMA_API int cinema_launch_probe(int n, void* stream) {
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, i);
  return launch_status();
}

// Sustained rate of the matrix pipe alone on this device (tools/mfma_peak.py): every wave issues `iters` x 16 independent
// v_mfma_f32_32x32x16_bf16 from registers (no memory traffic in the loop).  FLOPs per launch = grid * 4 waves * iters * 16 * 32768.
namespace {
typedef short mp_short8 __attribute__((ext_vector_type(8)));
typedef float mp_float16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void mfma_probe_kernel(int iters, float* out) {
  mp_short8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (short)(0x3c00 + threadIdx.x + i); b[i] = (short)(0x3c00 + 2 * threadIdx.x + i); }
  mp_float16 acc[4];
  for (int j = 0; j < 4; j++)
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < 4; j++)
    for (int r = 0; r < 16; r++) s += acc[j][r];
  if (s == 123.456f) out[0] = s;  // keep the accumulators alive
}
}  // namespace

CINEMA_API int cinema_mfma_probe(int grid, int iters, float* out, void* stream) {
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, iters, out);
  return launch_status();
}
