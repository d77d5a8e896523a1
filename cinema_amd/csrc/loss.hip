// Masked-patch MSE (reference mse_loss, cinema/mae/mae.py:107-152) with the target patch gathered on the fly
// from the fp32 image (no materialised patchify), per-patch normalisation option, and the metric reductions.
#include "common.cuh"
#include "../../include/cinema_hip.h"

namespace {

struct PatchG {
  int b, c, gx, gy, gz, px, py, pz;
  long long sb, sc, sx, sy, sz;
  int n_rows;
  const int* token_idx;
};

__device__ __forceinline__ float target_at(const float* image, const PatchG& g, int tok, int f) {
  const int G = g.gx * g.gy * g.gz;
  const int bb = tok / G, gi = tok % G;
  const int iz = gi % g.gz, iy = (gi / g.gz) % g.gy, ix = gi / (g.gz * g.gy);
  const int cc = f % g.c, pf = f / g.c;
  const int kz = pf % g.pz, ky = (pf / g.pz) % g.py, kx = pf / (g.pz * g.py);
  return image[(long long)bb * g.sb + (long long)(ix * g.px + kx) * g.sx + (long long)(iy * g.py + ky) * g.sy +
               (long long)(iz * g.pz + kz) * g.sz + (long long)cc * g.sc];
}

__device__ __forceinline__ void patch_moments(const float* image, const PatchG& g, int tok, int F, int lane, float& mean, float& stdv) {
  float s = 0.f;
  for (int f = lane; f < F; f += 64) s += target_at(image, g, tok, f);
  mean = wave_sum(s) / (float)F;
  float q = 0.f;
  for (int f = lane; f < F; f += 64) { const float d = target_at(image, g, tok, f) - mean; q += d * d; }
  stdv = sqrtf(wave_sum(q) / (float)(F - 1));  // unbiased, as torch.var (mae.py:130)
}

__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  unsigned int* a = reinterpret_cast<unsigned int*>(addr);
  unsigned int old = *a;
  while (__uint_as_float(old) < v) {
    const unsigned int prev = atomicCAS(a, old, __float_as_uint(v));
    if (prev == old) break;
    old = prev;
  }
}

__device__ __forceinline__ float pred_at(const void* pred, int dtype, size_t off) {
  return dtype == 0 ? bf2f(reinterpret_cast<const bf16_t*>(pred)[off]) : reinterpret_cast<const float*>(pred)[off];
}

__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* image, PatchG g, const void* pred, int pdt, int ldp, int norm_target, float eps,
                                                      float inv_count, float* loss_out, float* max_out) {
  const int lane = threadIdx.x & 63;
  const int F = g.px * g.py * g.pz * g.c;
  const int nw = gridDim.x * 4;
  float acc = 0.f, tmax = -INFINITY, pmax = -INFINITY;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < g.n_rows; row += nw) {
    const int tok = g.token_idx ? g.token_idx[row] : row;
    float mean = 0.f, stdv = 1.f;
    if (norm_target) patch_moments(image, g, tok, F, lane, mean, stdv);
    for (int f = lane; f < F; f += 64) {
      float t = target_at(image, g, tok, f);
      if (norm_target) t = (t - mean) / (stdv + eps);
      const float pv = pred_at(pred, pdt, (size_t)row * ldp + f);
      const float d = pv - t;
      acc += d * d;
      tmax = fmaxf(tmax, t);
      pmax = fmaxf(pmax, pv);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) unsafeAtomicAdd(loss_out, acc * inv_count);
  if (max_out) {  // metrics `normed_target_max` / `pred_max` (mae.py:146-150); max_out is pre-filled with -inf by the caller
    tmax = wave_max(tmax); pmax = wave_max(pmax);
    if (lane == 0) { atomic_max_f32(max_out, tmax); atomic_max_f32(max_out + 1, pmax); }
  }
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* image, PatchG g, const void* pred, int pdt, int ldp, int norm_target, float eps,
                                                      const float* upstream, float host_scale, bf16_t* dpred, int ldd) {
  const int lane = threadIdx.x & 63;
  const int F = g.px * g.py * g.pz * g.c;
  const int nw = gridDim.x * 4;
  const float coef = 2.f * host_scale * (upstream ? upstream[0] : 1.f);
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < g.n_rows; row += nw) {
    const int tok = g.token_idx ? g.token_idx[row] : row;
    float mean = 0.f, stdv = 1.f;
    if (norm_target) patch_moments(image, g, tok, F, lane, mean, stdv);
    for (int f = lane; f < F; f += 64) {
      float t = target_at(image, g, tok, f);
      if (norm_target) t = (t - mean) / (stdv + eps);
      dpred[(size_t)row * ldd + f] = f2bf(coef * (pred_at(pred, pdt, (size_t)row * ldp + f) - t));
    }
  }
}

__global__ __launch_bounds__(256) void patch_stats_kernel(const float* image, PatchG g, float inv_n, float* out2) {
  const int lane = threadIdx.x & 63;
  const int F = g.px * g.py * g.pz * g.c;
  const int nw = gridDim.x * 4;
  float am = 0.f, as = 0.f;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < g.n_rows; row += nw) {
    float mean, stdv;
    patch_moments(image, g, g.token_idx ? g.token_idx[row] : row, F, lane, mean, stdv);
    am += mean; as += stdv;
  }
  if (lane == 0) { unsafeAtomicAdd(out2, am * inv_n); unsafeAtomicAdd(out2 + 1, as * inv_n); }
}

// mean over the finite entries (reference: `if torch.isfinite(loss_view)` mae.py:604-608); coef[i] = d mean / d vals[i]
__global__ void mean_finite_kernel(const float* vals, int n, float* mean_out, float* coef_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int cnt = 0; float s = 0.f;
  for (int i = 0; i < n; i++) if (isfinite(vals[i])) { cnt++; s += vals[i]; }
  mean_out[0] = cnt > 0 ? s / (float)cnt : __uint_as_float(0x7fc00000u);
  if (coef_out) for (int i = 0; i < n; i++) coef_out[i] = (cnt > 0 && isfinite(vals[i])) ? 1.f / (float)cnt : 0.f;
}

PatchG to_dev(const cinema_patch_geom* g) {
  PatchG p;
  p.b = g->b; p.c = g->c; p.gx = g->gx; p.gy = g->gy; p.gz = g->gz; p.px = g->px; p.py = g->py; p.pz = g->pz;
  p.sb = g->sb; p.sc = g->sc; p.sx = g->sx; p.sy = g->sy; p.sz = g->sz; p.n_rows = g->n_rows; p.token_idx = g->token_idx;
  return p;
}
int rows_grid(int n_rows) { int g = (n_rows + 3) / 4; return g > 4096 ? 4096 : (g < 1 ? 1 : g); }

}  // namespace

CINEMA_API int cinema_mse_fwd(const float* image, const cinema_patch_geom* geom, const void* pred, int pred_dtype, int ld_pred, int norm_target,
                              float eps, float inv_count, float* loss_out, float* max_out, void* stream) {
  if (!image || !geom || !pred || !loss_out || geom->n_rows <= 0) return CINEMA_ERR_BAD_ARG;
  hipLaunchKernelGGL(mse_fwd_kernel, dim3(rows_grid(geom->n_rows)), dim3(256), 0, (hipStream_t)stream, image, to_dev(geom), pred, pred_dtype, ld_pred,
                     norm_target, eps, inv_count, loss_out, max_out);
  return launch_status();
}

CINEMA_API int cinema_mse_bwd(const float* image, const cinema_patch_geom* geom, const void* pred, int pred_dtype, int ld_pred, int norm_target,
                              float eps, const float* upstream, float host_scale, uint16_t* dpred, int ld_dpred, void* stream) {
  if (!image || !geom || !pred || !dpred || geom->n_rows <= 0) return CINEMA_ERR_BAD_ARG;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(rows_grid(geom->n_rows)), dim3(256), 0, (hipStream_t)stream, image, to_dev(geom), pred, pred_dtype, ld_pred,
                     norm_target, eps, upstream, host_scale, dpred, ld_dpred);
  return launch_status();
}

CINEMA_API int cinema_patch_stats(const float* image, const cinema_patch_geom* geom_all, float* out2, void* stream) {
  if (!image || !geom_all || !out2 || geom_all->n_rows <= 0) return CINEMA_ERR_BAD_ARG;
  hipLaunchKernelGGL(patch_stats_kernel, dim3(rows_grid(geom_all->n_rows)), dim3(256), 0, (hipStream_t)stream, image, to_dev(geom_all),
                     1.f / (float)geom_all->n_rows, out2);
  return launch_status();
}

CINEMA_API int cinema_mean_finite(const float* vals, int n, float* mean_out, float* coef_out, void* stream) {
  if (!vals || !mean_out || n <= 0) return CINEMA_ERR_BAD_ARG;
  hipLaunchKernelGGL(mean_finite_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, vals, n, mean_out, coef_out);
  return launch_status();
}
