// Stream-ordering helpers of the C-ABI (see include/cinema_hip.h): the weight-gradient GEMMs of the backward pass run on a second
// HIP stream, one fork (event record + wait) per launch, ~200 per step.  The events come from a per-device ring created once, so a
// fork is two HIP calls and no allocation; a host that drives the library from Python would otherwise pay an event object, a
// stream-context switch and three interpreter round trips per launch, which made the step launch-bound on slower hosts.
#include <cstdio>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <vector>

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

// ---- lane groups (see common.cuh) --------------------------------------------------------------------------------------------------------------
namespace {
struct Deferred {
  int kind = 0;  // 0 kernel, 1 stream fork
  lane_relaunch_t relaunch = nullptr;
  const void* fn_single = nullptr; const void* fn_lanes = nullptr; int lane_dim = 0;
  dim3 grid, block; size_t smem = 0; hipStream_t st = nullptr, st2 = nullptr;
  std::vector<char> params;
};
struct LaneGroup {
  int n = 0, cur = 0;
  std::vector<Deferred> seq[MAX_LANES];
};
LaneGroup g_lanes;

int launch_one(const Deferred& d) { return d.relaunch(d.fn_single, d.grid, d.block, d.smem, d.st, d.params.data()); }
int fork_now(hipStream_t from, hipStream_t to);
}  // namespace

bool lanes_active() { return g_lanes.n > 0; }

int lane_submit(lane_relaunch_t relaunch, const void* fn_single, const void* fn_lanes, int lane_dim, dim3 grid, dim3 block, size_t smem, hipStream_t st,
                const void* params, size_t psize) {
  if (g_lanes.n == 0) return relaunch(fn_single, grid, block, smem, st, params);
  Deferred d;
  d.kind = 0; d.relaunch = relaunch; d.fn_single = fn_single; d.fn_lanes = fn_lanes; d.lane_dim = lane_dim; d.grid = grid; d.block = block; d.smem = smem; d.st = st;
  d.params.assign((const char*)params, (const char*)params + psize);
  g_lanes.seq[g_lanes.cur].push_back(std::move(d));
  return 0;
}

CINEMA_API int cinema_lanes_begin(int n) {
  if (n < 1 || n > MAX_LANES) return CINEMA_ERR_BAD_ARG;
  // a group left open by a caller that died between begin and end holds launches that were never issued: drop them (their buffers are gone) instead
  // of refusing every later group - and every later launch - for the rest of the process; a begin inside an open group is still a programming error
  // (hosts close or cinema_lanes_abort() their groups), so it does not pass silently
  if (g_lanes.n != 0) {
    size_t dropped = 0;
    for (int i = 0; i < MAX_LANES; i++) dropped += g_lanes.seq[i].size();
    fprintf(stderr, "cinema_lanes_begin: a lane group of %d was still open, %zu recorded launches dropped\n", g_lanes.n, dropped);
  }
  g_lanes.n = n; g_lanes.cur = 0;
  for (int i = 0; i < MAX_LANES; i++) g_lanes.seq[i].clear();
  return 0;
}

// Drop an open group without issuing anything (error paths: the recorded launches reference buffers the failed caller no longer owns).
CINEMA_API int cinema_lanes_abort(void) {
  g_lanes.n = 0; g_lanes.cur = 0;
  for (int i = 0; i < MAX_LANES; i++) g_lanes.seq[i].clear();
  return 0;
}

CINEMA_API int cinema_lanes_select(int lane) {
  if (g_lanes.n == 0 || lane < 0 || lane >= g_lanes.n) return CINEMA_ERR_BAD_ARG;
  g_lanes.cur = lane;
  return 0;
}

// Issue the recorded sequences: merged where the lanes agree, one by one otherwise.  merged_out / single_out (may be NULL) count the launches issued.
CINEMA_API int cinema_lanes_end(int* merged_out, int* single_out) {
  if (g_lanes.n == 0) return CINEMA_ERR_BAD_ARG;
  const int n = g_lanes.n;
  g_lanes.n = 0;  // launches below go straight out
  int merged = 0, single = 0, rc = 0;
  size_t len = g_lanes.seq[0].size();
  bool lockstep = n > 1;
  for (int i = 1; i < n; i++) lockstep = lockstep && g_lanes.seq[i].size() == len;
  auto issue = [&](const Deferred& d) {
    const int e = d.kind == 1 ? fork_now(d.st, d.st2) : launch_one(d);
    if (e != 0 && rc == 0) rc = e;
    single += d.kind == 0;
  };
  if (!lockstep) {
    for (int i = 0; i < n; i++)
      for (const Deferred& d : g_lanes.seq[i]) issue(d);
  } else {
    std::vector<char> buf;
    for (size_t j = 0; j < len; j++) {
      const Deferred& d0 = g_lanes.seq[0][j];
      bool same = true;
      for (int i = 1; i < n && same; i++) {
        const Deferred& d = g_lanes.seq[i][j];
        same = d.kind == d0.kind && d.fn_single == d0.fn_single && d.grid.x == d0.grid.x && d.grid.y == d0.grid.y && d.grid.z == d0.grid.z &&
               d.block.x == d0.block.x && d.block.y == d0.block.y && d.smem == d0.smem && d.st == d0.st && d.st2 == d0.st2 && d.params.size() == d0.params.size();
      }
      if (same && d0.kind == 1) {  // the same fork in every lane: once
        const int e = fork_now(d0.st, d0.st2);
        if (e != 0 && rc == 0) rc = e;
        continue;
      }
      const bool can_merge = same && d0.fn_lanes != nullptr && (d0.lane_dim == 1 ? d0.grid.y == 1 : d0.grid.z == 1);
      if (!can_merge) {
        for (int i = 0; i < n; i++) issue(g_lanes.seq[i][j]);
        continue;
      }
      const size_t ps = d0.params.size();
      buf.assign(MAX_LANES * ps, 0);
      for (int i = 0; i < n; i++) memcpy(buf.data() + i * ps, g_lanes.seq[i][j].params.data(), ps);
      dim3 grid = d0.grid;
      if (d0.lane_dim == 1) grid.y = n; else grid.z = n;
      void* args[] = {(void*)buf.data()};
      g_kernel_launches.fetch_add(1, std::memory_order_relaxed);
      const int e = (int)hipLaunchKernel(d0.fn_lanes, grid, d0.block, args, d0.smem, d0.st);
      if (e != 0 && rc == 0) rc = e;
      merged++;
    }
  }
  for (int i = 0; i < MAX_LANES; i++) g_lanes.seq[i].clear();
  if (merged_out) *merged_out = merged;
  if (single_out) *single_out = single;
  return rc;
}

namespace {
int fork_now(hipStream_t from, hipStream_t to) {
  std::lock_guard<std::mutex> lock(g_mutex);
  DeviceRing* r = ring_of_current_device();
  if (!r) return CINEMA_ERR_UNSUPPORTED;
  hipEvent_t ev = r->fork_events[r->next_fork++ % RING];
  hipError_t e = hipEventRecord(ev, from);
  if (e == hipSuccess) e = hipStreamWaitEvent(to, ev, 0);
  return (int)e;
}
}  // namespace

CINEMA_API int cinema_stream_fork(void* from_stream, void* to_stream) {
  if (from_stream == to_stream) return 0;
  if (g_lanes.n > 0) {  // inside a lane group the fork keeps its place in the lane's sequence
    Deferred d;
    d.kind = 1;
    d.st = (hipStream_t)from_stream; d.st2 = (hipStream_t)to_stream;
    g_lanes.seq[g_lanes.cur].push_back(std::move(d));
    return 0;
  }
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

CINEMA_API long long cinema_kernel_launch_count(void) { return g_kernel_launches.load(std::memory_order_relaxed); }

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
