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

// Fast path (patch features F <= 64 * MAXI): the feature -> image offset of a lane's features does not depend on the token, so
// it is decomposed ONCE per thread (the per-element div/mod chain of target_at() made these kernels ALU-bound: 157 us to
// read 38 MB); a wave then holds its whole patch in registers for the moments, the loss and the gradient.
constexpr int MAXI = 8;
__device__ __forceinline__ void feat_offsets(const PatchG& g, int F, int lane, long long (&o)[MAXI]) {
#pragma unroll
  for (int i = 0; i < MAXI; i++) {
    const int f = lane + 64 * i;
    const int cc = f % g.c, pf = f / g.c;
    const int kz = pf % g.pz, ky = (pf / g.pz) % g.py, kx = pf / (g.pz * g.py);
    o[i] = f < F ? (long long)kx * g.sx + (long long)ky * g.sy + (long long)kz * g.sz + (long long)cc * g.sc : 0;
  }
}
__device__ __forceinline__ long long tok_base(const PatchG& g, int tok) {
  const int G = g.gx * g.gy * g.gz;
  const int bb = tok / G, gi = tok % G;
  const int iz = gi % g.gz, iy = (gi / g.gz) % g.gy, ix = gi / (g.gz * g.gy);
  return (long long)bb * g.sb + (long long)(ix * g.px) * g.sx + (long long)(iy * g.py) * g.sy + (long long)(iz * g.pz) * g.sz;
}
__device__ __forceinline__ void load_patch(const float* image, long long base, const long long (&o)[MAXI], int F, int lane, float (&t)[MAXI]) {
#pragma unroll
  for (int i = 0; i < MAXI; i++) t[i] = (lane + 64 * i < F) ? image[base + o[i]] : 0.f;
}
__device__ __forceinline__ void reg_moments(const float (&t)[MAXI], int F, int lane, float& mean, float& stdv) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; i++) s += t[i];
  mean = wave_sum(s) / (float)F;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; i++) { const float d = t[i] - mean; q += (lane + 64 * i < F) ? d * d : 0.f; }
  stdv = sqrtf(wave_sum(q) / (float)(F - 1));  // unbiased, as torch.var (mae.py:130)
}

// per-block combine of the 4 waves' partials -> ONE atomic per block and output (16k waves hammering three addresses
// serialised the forward kernel: 74 us for 11 MB)
__device__ __forceinline__ void block_add_max(float acc, float tmax, float pmax, float* loss_out, float* max_out) {
  __shared__ float red[3][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  acc = wave_sum(acc); tmax = wave_max(tmax); pmax = wave_max(pmax);
  if (lane == 0) { red[0][wave] = acc; red[1][wave] = tmax; red[2][wave] = pmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(loss_out, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    if (max_out) {  // metrics `normed_target_max` / `pred_max` (mae.py:146-150); max_out is pre-filled with -inf by the caller
      atomic_max_f32(max_out, fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3])));
      atomic_max_f32(max_out + 1, fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3])));
    }
  }
}

template <bool FAST>
__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* image, PatchG g, const void* pred, int pdt, int ldp, int norm_target, float eps,
                                                      float inv_count, float* loss_out, float* max_out) {
  const int lane = threadIdx.x & 63;
  const int F = g.px * g.py * g.pz * g.c;
  const int nw = gridDim.x * 4;
  float acc = 0.f, tmax = -INFINITY, pmax = -INFINITY;
  long long offs[MAXI];
  if (FAST) feat_offsets(g, F, lane, offs);
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < g.n_rows; row += nw) {
    const int tok = g.token_idx ? g.token_idx[row] : row;
    float mean = 0.f, stdv = 1.f;
    if (FAST) {
      float t[MAXI];
      load_patch(image, tok_base(g, tok), offs, F, lane, t);
      if (norm_target) reg_moments(t, F, lane, mean, stdv);
#pragma unroll
      for (int i = 0; i < MAXI; i++) {
        const int f = lane + 64 * i;
        if (f < F) {
          const float tt = norm_target ? (t[i] - mean) / (stdv + eps) : t[i];
          const float pv = pred_at(pred, pdt, (size_t)row * ldp + f);
          const float d = pv - tt;
          acc += d * d;
          tmax = fmaxf(tmax, tt);
          pmax = fmaxf(pmax, pv);
        }
      }
    } else {
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
  }
  block_add_max(acc * inv_count, tmax, pmax, loss_out, max_out);
}

template <bool FAST>
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* image, PatchG g, const void* pred, int pdt, int ldp, int norm_target, float eps,
                                                      const float* upstream, float host_scale, bf16_t* dpred, int ldd) {
  const int lane = threadIdx.x & 63;
  const int F = g.px * g.py * g.pz * g.c;
  const int nw = gridDim.x * 4;
  const float coef = 2.f * host_scale * (upstream ? upstream[0] : 1.f);
  long long offs[MAXI];
  if (FAST) feat_offsets(g, F, lane, offs);
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < g.n_rows; row += nw) {
    const int tok = g.token_idx ? g.token_idx[row] : row;
    float mean = 0.f, stdv = 1.f;
    if (FAST) {
      float t[MAXI];
      load_patch(image, tok_base(g, tok), offs, F, lane, t);
      if (norm_target) reg_moments(t, F, lane, mean, stdv);
#pragma unroll
      for (int i = 0; i < MAXI; i++) {
        const int f = lane + 64 * i;
        if (f < F) {
          const float tt = norm_target ? (t[i] - mean) / (stdv + eps) : t[i];
          dpred[(size_t)row * ldd + f] = f2bf(coef * (pred_at(pred, pdt, (size_t)row * ldp + f) - tt));
        }
      }
    } else {
      if (norm_target) patch_moments(image, g, tok, F, lane, mean, stdv);
      for (int f = lane; f < F; f += 64) {
        float t = target_at(image, g, tok, f);
        if (norm_target) t = (t - mean) / (stdv + eps);
        dpred[(size_t)row * ldd + f] = f2bf(coef * (pred_at(pred, pdt, (size_t)row * ldp + f) - t));
      }
    }
  }
}

template <bool FAST>
__global__ __launch_bounds__(256) void patch_stats_kernel(const float* image, PatchG g, float inv_n, float* out2) {
  const int lane = threadIdx.x & 63;
  const int F = g.px * g.py * g.pz * g.c;
  const int nw = gridDim.x * 4;
  float am = 0.f, as = 0.f;
  long long offs[MAXI];
  if (FAST) feat_offsets(g, F, lane, offs);
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < g.n_rows; row += nw) {
    const int tok = g.token_idx ? g.token_idx[row] : row;
    float mean, stdv;
    if (FAST) {
      float t[MAXI];
      load_patch(image, tok_base(g, tok), offs, F, lane, t);
      reg_moments(t, F, lane, mean, stdv);
    } else {
      patch_moments(image, g, tok, F, lane, mean, stdv);
    }
    am += mean; as += stdv;
  }
  __shared__ float red[2][4];
  if (lane == 0) { red[0][threadIdx.x >> 6] = am; red[1][threadIdx.x >> 6] = as; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(out2, ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * inv_n);
    unsafeAtomicAdd(out2 + 1, ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * inv_n);
  }
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
int rows_grid(int n_rows) { int g = (n_rows + 3) / 4; return g > 2048 ? 2048 : (g < 1 ? 1 : g); }
bool fast_patch(const cinema_patch_geom* g) { return (long long)g->px * g->py * g->pz * g->c <= 64 * MAXI; }


// ---- losses of the ConvViT heads (a handful of rows): one workgroup, forward and gradient in one launch
// cross entropy with label smoothing (F.cross_entropy(logits, label, label_smoothing = eps), mean over the batch):
//   loss_i = (1 - eps) * (lse_i - z_i[y_i]) + eps / c * sum_j (lse_i - z_i[j]);   d loss / d z_i[j] = (softmax_i[j] - (1 - eps) [j == y_i] - eps / c) / b
__global__ __launch_bounds__(256) void head_ce_kernel(const float* logits, const int* labels, int b, int c, float eps, float* out, float* dlogits) {
  __shared__ float red[256];
  float part = 0.f;
  for (int i = threadIdx.x; i < b; i += 256) {
    const float* z = logits + (size_t)i * c;
    float mx = z[0];
    for (int j = 1; j < c; j++) mx = fmaxf(mx, z[j]);
    float se = 0.f, sz = 0.f;
    for (int j = 0; j < c; j++) { se += expf(z[j] - mx); sz += z[j]; }
    const float lse = mx + logf(se);
    const int y = labels[i];
    if (y < 0 || y >= c) {  // F.cross_entropy raises for such a label: the loss becomes NaN (the step's non-finite guard skips the update) instead of
      part = NAN;           // reading z[y] out of bounds
      for (int j = 0; j < c; j++) dlogits[(size_t)i * c + j] = NAN;
      continue;
    }
    part += (1.f - eps) * (lse - z[y]) + eps * (lse - sz / c);
    const float inv = 1.f / se;
    for (int j = 0; j < c; j++) dlogits[(size_t)i * c + j] = (expf(z[j] - mx) * inv - (j == y ? 1.f - eps : 0.f) - eps / c) / b;
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] / b;
}
// mean squared error over all n = rows * cols elements with the reported values of the reference's regression_loss:
//   out = {mse, mae, max label, min label, max pred, min pred};  d mse / d pred = 2 (pred - label) / n
__global__ __launch_bounds__(256) void head_mse_kernel(const float* pred, const float* label, int n, float* out, float* dpred) {
  __shared__ float red[6][256];
  float s2 = 0.f, s1 = 0.f, mxl = -INFINITY, mnl = INFINITY, mxp = -INFINITY, mnp = INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = pred[i] - label[i];
    s2 += d * d; s1 += fabsf(d);
    mxl = fmaxf(mxl, label[i]); mnl = fminf(mnl, label[i]); mxp = fmaxf(mxp, pred[i]); mnp = fminf(mnp, pred[i]);
    dpred[i] = 2.f * d / n;
  }
  red[0][threadIdx.x] = s2; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = mxl; red[3][threadIdx.x] = mnl; red[4][threadIdx.x] = mxp; red[5][threadIdx.x] = mnp;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const int t = threadIdx.x;
      red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o];
      red[2][t] = fmaxf(red[2][t], red[2][t + o]); red[3][t] = fminf(red[3][t], red[3][t + o]);
      red[4][t] = fmaxf(red[4][t], red[4][t + o]); red[5][t] = fminf(red[5][t], red[5][t + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = red[0][0] / n; out[1] = red[1][0] / n; out[2] = red[2][0]; out[3] = red[3][0]; out[4] = red[4][0]; out[5] = red[5][0];
  }
}

}  // namespace

CINEMA_API int cinema_mse_fwd(const float* image, const cinema_patch_geom* geom, const void* pred, int pred_dtype, int ld_pred, int norm_target,
                              float eps, float inv_count, float* loss_out, float* max_out, void* stream) {
  if (!image || !geom || !pred || !loss_out || geom->n_rows <= 0) return CINEMA_ERR_BAD_ARG;
  if (fast_patch(geom))
    CINEMA_LAUNCH(mse_fwd_kernel<true>, dim3(rows_grid(geom->n_rows)), dim3(256), 0, (hipStream_t)stream, image, to_dev(geom), pred, pred_dtype, ld_pred,
                       norm_target, eps, inv_count, loss_out, max_out);
  else
    CINEMA_LAUNCH(mse_fwd_kernel<false>, dim3(rows_grid(geom->n_rows)), dim3(256), 0, (hipStream_t)stream, image, to_dev(geom), pred, pred_dtype, ld_pred,
                       norm_target, eps, inv_count, loss_out, max_out);
  return launch_status();
}

CINEMA_API int cinema_mse_bwd(const float* image, const cinema_patch_geom* geom, const void* pred, int pred_dtype, int ld_pred, int norm_target,
                              float eps, const float* upstream, float host_scale, uint16_t* dpred, int ld_dpred, void* stream) {
  if (!image || !geom || !pred || !dpred || geom->n_rows <= 0) return CINEMA_ERR_BAD_ARG;
  if (fast_patch(geom))
    CINEMA_LAUNCH(mse_bwd_kernel<true>, dim3(rows_grid(geom->n_rows)), dim3(256), 0, (hipStream_t)stream, image, to_dev(geom), pred, pred_dtype, ld_pred,
                       norm_target, eps, upstream, host_scale, dpred, ld_dpred);
  else
    CINEMA_LAUNCH(mse_bwd_kernel<false>, dim3(rows_grid(geom->n_rows)), dim3(256), 0, (hipStream_t)stream, image, to_dev(geom), pred, pred_dtype, ld_pred,
                       norm_target, eps, upstream, host_scale, dpred, ld_dpred);
  return launch_status();
}

CINEMA_API int cinema_patch_stats(const float* image, const cinema_patch_geom* geom_all, float* out2, void* stream) {
  if (!image || !geom_all || !out2 || geom_all->n_rows <= 0) return CINEMA_ERR_BAD_ARG;
  if (fast_patch(geom_all))
    CINEMA_LAUNCH(patch_stats_kernel<true>, dim3(rows_grid(geom_all->n_rows)), dim3(256), 0, (hipStream_t)stream, image, to_dev(geom_all),
                       1.f / (float)geom_all->n_rows, out2);
  else
    CINEMA_LAUNCH(patch_stats_kernel<false>, dim3(rows_grid(geom_all->n_rows)), dim3(256), 0, (hipStream_t)stream, image, to_dev(geom_all),
                       1.f / (float)geom_all->n_rows, out2);
  return launch_status();
}

CINEMA_API int cinema_mean_finite(const float* vals, int n, float* mean_out, float* coef_out, void* stream) {
  if (!vals || !mean_out || n <= 0) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(mean_finite_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, vals, n, mean_out, coef_out);
  return launch_status();
}

// ------------------------------------------------------------------------------------------------
// Segmentation loss of one view (reference _segmentation_loss, cinema/segmentation/train.py:77-103):
//   cross_entropy(logits, labels, ignore_index = -1)  +  monai DiceLoss(include_background=False, softmax=True) against one_hot(max(labels, 0))
// (monai defaults: smooth_nr = smooth_dr = 1e-5, reduced over the spatial axes per (sample, class), mean over samples x foreground classes).
// logits: fp32 channels-last rows [b * vox][c] (c <= 16), labels int32 [b * vox].
//   pass 1  per (sample, class) sums of p*t, p, t  + CE sum / count            -> acc[b][c][3], acc_ce[2]
//   finish  loss, metrics and the coefficients d dice / d (sum p*t), d dice / d (sum p)   -> out[3], coef[b][c][2], inv_count
//   pass 2  d loss / d logits (softmax Jacobian applied per voxel)
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int SEG_MAXC = 16;

__device__ __forceinline__ void row_softmax(const float* row, int c, float (&p)[SEG_MAXC], float& lse) {
  float mx = -INFINITY;
  for (int j = 0; j < c; j++) mx = fmaxf(mx, row[j]);
  float s = 0.f;
  for (int j = 0; j < c; j++) { p[j] = __expf(row[j] - mx); s += p[j]; }
  const float inv = 1.f / s;
  for (int j = 0; j < c; j++) p[j] *= inv;
  lse = mx + __logf(s);
}

__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const float* logits, const int* labels, int vox, int c, float* acc, float* acc_ce) {
  __shared__ float red[SEG_MAXC * 3 + 2];
  for (int i = threadIdx.x; i < c * 3 + 2; i += 256) red[i] = 0.f;
  __syncthreads();
  const int b = blockIdx.y;
  float li[SEG_MAXC], lp[SEG_MAXC], lt[SEG_MAXC], ce = 0.f, cnt = 0.f;
  for (int j = 0; j < c; j++) { li[j] = 0.f; lp[j] = 0.f; lt[j] = 0.f; }
  for (int v = blockIdx.x * 256 + threadIdx.x; v < vox; v += gridDim.x * 256) {
    const size_t r = (size_t)b * vox + v;
    float p[SEG_MAXC], lse;
    row_softmax(logits + r * c, c, p, lse);
    const int lab = labels[r];
    const int tcls = lab < 0 ? 0 : lab;  // one_hot(labels.clamp(min=0))
    for (int j = 0; j < c; j++) { lp[j] += p[j]; if (j == tcls) { li[j] += p[j]; lt[j] += 1.f; } }
    if (lab >= 0) { ce += lse - logits[r * c + lab]; cnt += 1.f; }
  }
  for (int j = 0; j < c; j++) {
    const float a = wave_sum(li[j]), bb = wave_sum(lp[j]), t = wave_sum(lt[j]);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&red[j * 3], a); atomicAdd(&red[j * 3 + 1], bb); atomicAdd(&red[j * 3 + 2], t); }
  }
  ce = wave_sum(ce); cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) { atomicAdd(&red[c * 3], ce); atomicAdd(&red[c * 3 + 1], cnt); }
  __syncthreads();
  for (int i = threadIdx.x; i < c * 3; i += 256) unsafeAtomicAdd(acc + (size_t)b * c * 3 + i, red[i]);
  if (threadIdx.x < 2) unsafeAtomicAdd(acc_ce + threadIdx.x, red[c * 3 + threadIdx.x]);
}

// out[0] = loss, out[1] = cross entropy, out[2] = mean dice loss; coef[b][c][0] = d dice / d I, coef[b][c][1] = d dice / d P; out[3] = 1 / count
__global__ void seg_loss_finish_kernel(const float* acc, const float* acc_ce, int b, int c, float smooth_nr, float smooth_dr, float* out, float* coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float n = (float)(b * (c - 1));
  float dice = 0.f;
  for (int i = 0; i < b; i++)
    for (int j = 0; j < c; j++) {
      const float I = acc[(i * c + j) * 3], P = acc[(i * c + j) * 3 + 1], G = acc[(i * c + j) * 3 + 2];
      const float den = G + P + smooth_dr;
      float dI = 0.f, dP = 0.f;
      if (j >= 1) {
        dice += 1.f - (2.f * I + smooth_nr) / den;
        dI = -2.f / den / n;
        dP = (2.f * I + smooth_nr) / (den * den) / n;
      }
      coef[(i * c + j) * 2] = dI;
      coef[(i * c + j) * 2 + 1] = dP;
    }
  dice /= n;
  const float ce = acc_ce[0] / acc_ce[1];
  out[0] = dice + ce; out[1] = ce; out[2] = dice; out[3] = 1.f / acc_ce[1];
}

__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* logits, const int* labels, int vox, int c, const float* coef, const float* out,
                                                           const float* upstream, float* dlogits) {
  const int b = blockIdx.y;
  const float up = upstream ? upstream[0] : 1.f, inv_cnt = out[3];
  for (int v = blockIdx.x * 256 + threadIdx.x; v < vox; v += gridDim.x * 256) {
    const size_t r = (size_t)b * vox + v;
    float p[SEG_MAXC], lse;
    row_softmax(logits + r * c, c, p, lse);
    const int lab = labels[r];
    const int tcls = lab < 0 ? 0 : lab;
    float g[SEG_MAXC], dot = 0.f;
    for (int j = 0; j < c; j++) {
      g[j] = coef[(b * c + j) * 2 + 1] + (j == tcls ? coef[(b * c + j) * 2] : 0.f);  // d dice / d p_j
      dot += g[j] * p[j];
    }
    for (int j = 0; j < c; j++) {
      float d = p[j] * (g[j] - dot);
      if (lab >= 0) d += (p[j] - (j == lab ? 1.f : 0.f)) * inv_cnt;
      dlogits[r * c + j] = d * up;
    }
  }
}

// ---- evaluation path (cinema/segmentation/train.py:148-286): sliding-window aggregation and segmentation metrics ------------------------------
// One window: softmax over the classes of every window voxel (channels-last fp32 rows of the window's logits), added into the channels-last
// probability volume at the window's offset; count += 1 there (aggregate_patches, cinema/transform.py:86-124, after F.softmax at train.py:212).
// Windows are accumulated one launch after the other in grid order, so the sums are formed in the reference's order (deterministic).
__global__ __launch_bounds__(256) void seg_window_accumulate_kernel(const float* logits, int c, int px, int py, int pz, int sx, int sy, int sz, int X, int Y, int Z,
                                                                    float* prob_sum, float* count) {
  const int nvox = px * py * pz;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < nvox; v += gridDim.x * 256) {
    const int z = v % pz, y = (v / pz) % py, x = v / (pz * py);
    float p[SEG_MAXC], lse;
    row_softmax(logits + (size_t)v * c, c, p, lse);
    const size_t o = ((size_t)(x + sx) * Y + (y + sy)) * Z + (z + sz);
    for (int j = 0; j < c; j++) prob_sum[o * c + j] += p[j];
    count[o] += 1.f;
  }
  (void)X;
}

// out[(j * nvox) + v] = log(prob_sum[v][j] / count[v])  (channels-first logits of the aggregated probabilities, train.py:213-214)
__global__ __launch_bounds__(256) void seg_window_finish_kernel(const float* prob_sum, const float* count, int c, long long nvox, float* out) {
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (long long)gridDim.x * 256) {
    const float inv = 1.f / count[v];
    for (int j = 0; j < c; j++) out[(size_t)j * nvox + v] = logf(prob_sum[v * c + j] * inv);
  }
}

// Per (sample, class) voxel counts for segmentation_metrics (train.py:224-286): [0] predicted (argmax, first maximum like torch.argmax), [1] true,
// [2] both (Dice / IoU / volumes); stability_score (cinema/metric.py:21-45): [3] logit - mean >= +1, [4] >= -1, [5] both.  Channels-first logits.
__global__ __launch_bounds__(256) void seg_metric_counts_kernel(const float* logits, const int* labels, int vox, int c, unsigned int* counts) {
  __shared__ unsigned int part[SEG_MAXC * 6];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < c * 6; i += 256) part[i] = 0;
  __syncthreads();
  const float* base = logits + (size_t)b * c * vox;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < vox; v += gridDim.x * 256) {
    float l[SEG_MAXC], mean = 0.f;
    int arg = 0;
    for (int j = 0; j < c; j++) {
      l[j] = base[(size_t)j * vox + v];
      mean += l[j];
      if (l[j] > l[arg]) arg = j;
    }
    mean /= (float)c;
    const int lab = labels[(size_t)b * vox + v];
    atomicAdd(&part[arg * 6 + 0], 1u);
    if (lab >= 0 && lab < c) {
      atomicAdd(&part[lab * 6 + 1], 1u);
      if (lab == arg) atomicAdd(&part[lab * 6 + 2], 1u);
    }
    for (int j = 0; j < c; j++) {
      const float n = l[j] - mean;
      const bool hi = n >= 1.f, lo = n >= -1.f;
      if (hi) atomicAdd(&part[j * 6 + 3], 1u);
      if (lo) atomicAdd(&part[j * 6 + 4], 1u);
      if (hi && lo) atomicAdd(&part[j * 6 + 5], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < c * 6; i += 256)
    if (part[i]) atomicAdd(&counts[(size_t)b * c * 6 + i], part[i]);
}

// Surface voxels of every class of a label map (monai get_mask_edges as compute_hausdorff_distance calls it: mask XOR binary_erosion(mask) with scipy's
// default cross-shaped structure and border value 0): a voxel of class k is an edge voxel of k unless all of its 2 * ndim face neighbours exist and carry k.
// label int32 [b][X][Y][Z] (ndim = 2: X == 1 and the x axis does not exist), edges uint8 [b][c][X*Y*Z].
__global__ __launch_bounds__(256) void mask_edges_kernel(const int* label, int X, int Y, int Z, int c, int ndim, unsigned char* edges) {
  const int b = blockIdx.y, vox = X * Y * Z;
  const int* lab = label + (size_t)b * vox;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < vox; v += gridDim.x * 256) {
    const int z = v % Z, y = (v / Z) % Y, x = v / (Z * Y);
    const int k = lab[v];
    bool inner = true;
    inner = inner && z > 0 && lab[v - 1] == k && z + 1 < Z && lab[v + 1] == k;
    inner = inner && y > 0 && lab[v - Z] == k && y + 1 < Y && lab[v + Z] == k;
    if (ndim == 3) inner = inner && x > 0 && lab[v - Z * Y] == k && x + 1 < X && lab[v + Z * Y] == k;
    for (int j = 0; j < c; j++) edges[((size_t)b * c + j) * vox + v] = (j == k && !inner) ? 1 : 0;
  }
}

// out[i] = min_j |a_i - b_j| (Euclidean, coordinates already in physical units): the value scipy's distance_transform_edt of the complement of set B takes at
// point a_i.  Brute force with B streamed through LDS: surface sets are a few thousand points.
__global__ __launch_bounds__(256) void min_dist_kernel(const float* a, const float* bpts, int na, int nb, float* out) {
  __shared__ float sb[256 * 3];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float ax = 0.f, ay = 0.f, az = 0.f;
  if (i < na) { ax = a[3 * i]; ay = a[3 * i + 1]; az = a[3 * i + 2]; }
  float best = INFINITY;
  for (int j0 = 0; j0 < nb; j0 += 256) {
    __syncthreads();
    const int j = j0 + threadIdx.x;
    if (j < nb) { sb[3 * threadIdx.x] = bpts[3 * j]; sb[3 * threadIdx.x + 1] = bpts[3 * j + 1]; sb[3 * threadIdx.x + 2] = bpts[3 * j + 2]; }
    __syncthreads();
    const int n = min(256, nb - j0);
    for (int t = 0; t < n; t++) {
      const float dx = ax - sb[3 * t], dy = ay - sb[3 * t + 1], dz = az - sb[3 * t + 2];
      best = fminf(best, dx * dx + dy * dy + dz * dz);
    }
  }
  if (i < na) out[i] = sqrtf(best);
}

}  // namespace

CINEMA_API int cinema_mask_edges(const int* label, int b, int X, int Y, int Z, int c, int ndim, unsigned char* edges, void* stream) {
  if (!label || !edges || b <= 0 || X <= 0 || Y <= 0 || Z <= 0 || c < 1 || (ndim != 2 && ndim != 3) || (ndim == 2 && X != 1)) return CINEMA_ERR_BAD_ARG;
  const long long vox = (long long)X * Y * Z;
  if (vox >= (1LL << 31)) return CINEMA_ERR_UNSUPPORTED;
  int gx = (int)((vox + 255) / 256);
  if (gx > 4096) gx = 4096;
  CINEMA_LAUNCH(mask_edges_kernel, dim3(gx, b), dim3(256), 0, (hipStream_t)stream, label, X, Y, Z, c, ndim, edges);
  return launch_status();
}

CINEMA_API int cinema_min_dist(const float* a, const float* bpts, int na, int nb, float* out, void* stream) {
  if (!a || !bpts || !out || na <= 0 || nb <= 0) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(min_dist_kernel, dim3((na + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, bpts, na, nb, out);
  return launch_status();
}

CINEMA_API int cinema_seg_window_accumulate(const float* window_logits, int c, int px, int py, int pz, int sx, int sy, int sz, int X, int Y, int Z,
                                            float* prob_sum, float* count, void* stream) {
  if (!window_logits || !prob_sum || !count || c < 2 || px <= 0 || py <= 0 || pz <= 0 || sx < 0 || sy < 0 || sz < 0 || sx + px > X || sy + py > Y || sz + pz > Z)
    return CINEMA_ERR_BAD_ARG;
  if (c > SEG_MAXC) return CINEMA_ERR_UNSUPPORTED;
  int gx = (px * py * pz + 255) / 256;
  if (gx > 4096) gx = 4096;
  CINEMA_LAUNCH(seg_window_accumulate_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, window_logits, c, px, py, pz, sx, sy, sz, X, Y, Z, prob_sum, count);
  return launch_status();
}

CINEMA_API int cinema_seg_window_finish(const float* prob_sum, const float* count, int c, long long n_voxels, float* logits_out, void* stream) {
  if (!prob_sum || !count || !logits_out || c < 2 || n_voxels <= 0) return CINEMA_ERR_BAD_ARG;
  long long gx = (n_voxels + 255) / 256;
  if (gx > 4096) gx = 4096;
  CINEMA_LAUNCH(seg_window_finish_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, prob_sum, count, c, n_voxels, logits_out);
  return launch_status();
}

CINEMA_API int cinema_seg_metric_counts(const float* logits, const int* labels, int b, int vox, int c, unsigned int* counts, void* stream) {
  if (!logits || !labels || !counts || b <= 0 || vox <= 0 || c < 2) return CINEMA_ERR_BAD_ARG;
  if (c > SEG_MAXC) return CINEMA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(counts, 0, (size_t)b * c * 6 * sizeof(unsigned int), st) != hipSuccess) return CINEMA_ERR_BAD_ARG;
  int gx = (vox + 255) / 256;
  if (gx > 1024) gx = 1024;
  CINEMA_LAUNCH(seg_metric_counts_kernel, dim3(gx, b), dim3(256), 0, st, logits, labels, vox, c, counts);
  return launch_status();
}

CINEMA_API int cinema_seg_loss_fwd(const float* logits, const int* labels, int b, int vox, int c, float* acc, float* out4, float* coef, void* stream) {
  if (!logits || !labels || !acc || !out4 || !coef || b <= 0 || vox <= 0 || c < 2) return CINEMA_ERR_BAD_ARG;
  if (c > SEG_MAXC) return CINEMA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(acc, 0, ((size_t)b * c * 3 + 2) * sizeof(float), st) != hipSuccess) return CINEMA_ERR_BAD_ARG;
  int gx = (vox + 255) / 256;
  if (gx > 512) gx = 512;
  CINEMA_LAUNCH(seg_loss_fwd_kernel, dim3(gx, b), dim3(256), 0, st, logits, labels, vox, c, acc, acc + (size_t)b * c * 3);
  CINEMA_LAUNCH(seg_loss_finish_kernel, dim3(1), dim3(64), 0, st, (const float*)acc, (const float*)(acc + (size_t)b * c * 3), b, c, 1e-5f, 1e-5f, out4, coef);
  return launch_status();
}

CINEMA_API int cinema_seg_loss_bwd(const float* logits, const int* labels, int b, int vox, int c, const float* coef, const float* out4, const float* upstream,
                                   float* dlogits, void* stream) {
  if (!logits || !labels || !coef || !out4 || !dlogits || b <= 0 || vox <= 0 || c < 2) return CINEMA_ERR_BAD_ARG;
  if (c > SEG_MAXC) return CINEMA_ERR_UNSUPPORTED;
  int gx = (vox + 255) / 256;
  if (gx > 2048) gx = 2048;
  CINEMA_LAUNCH(seg_loss_bwd_kernel, dim3(gx, b), dim3(256), 0, (hipStream_t)stream, logits, labels, vox, c, coef, out4, upstream, dlogits);
  return launch_status();
}

CINEMA_API int cinema_head_ce(const float* logits, const int* labels, int b, int c, float label_smoothing, float* out1, float* dlogits, void* stream) {
  if (!logits || !labels || !out1 || !dlogits || b <= 0 || c < 2 || !(label_smoothing >= 0.f && label_smoothing < 1.f)) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(head_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, b, c, label_smoothing, out1, dlogits);
  return launch_status();
}

CINEMA_API int cinema_head_mse(const float* pred, const float* label, int n, float* out6, float* dpred, void* stream) {
  if (!pred || !label || !out6 || !dpred || n <= 0) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(head_mse_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, label, n, out6, dpred);
  return launch_status();
}
