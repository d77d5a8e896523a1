// Optimiser kernels for gfx950 (flat fp32 buffers, HBM-bound): global grad-norm, clip coefficient, fused AdamW
// with optional bf16 shadow-weight emission.  Reference semantics: torch.optim.AdamW + clip_grad_norm_ as driven by
// cinema/optim.py:204-215 and cinema/mae/pretrain.py:365-366.
#include "common.cuh"
#include "../../include/cinema_hip.h"

namespace {

// Squared L2 norm in two deterministic passes: every block writes ONE partial (fixed thread -> element map, fixed in-block tree), a single block
// adds the partials in a fixed order.  (fp32 atomics gave run-to-run / rank-to-rank differences in the 7th digit of the norm: data-parallel
// replicas that clip with slightly different coefficients drift apart bit by bit.)
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* g, long long n, float* partial) {
  __shared__ float part[4];
  float s = 0.f;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(256) void sqnorm_finish_kernel(const float* partial, int n_partial, float* out) {
  __shared__ float part[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n_partial; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] += (part[0] + part[1]) + (part[2] + part[3]);
}

// torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (norm + 1e-6)).
// Non-finite guard (cinema/mae/pretrain.py:255-257 skips a NaN loss, torch's GradScaler.step skips inf/NaN gradients): a non-finite norm
// gives coef = 0, which adamw_kernel reads as "leave parameters, moments and shadows alone"; state[0] counts the updates applied (the Adam
// step used for the bias corrections), state[1] the updates skipped.  The decision stays on the device - no host round trip.
__global__ void clip_coef_kernel(const float* sq, float max_norm, float* coef, float* norm_out, int* state) {
  if (threadIdx.x || blockIdx.x) return;
  const float nrm = sqrtf(sq[0]);
  const bool ok = isfinite(nrm);
  if (norm_out) norm_out[0] = nrm;
  if (coef) coef[0] = !ok ? 0.f : (max_norm > 0.f ? fminf(1.f, max_norm / (nrm + 1e-6f)) : 1.f);
  if (state) state[ok ? 0 : 1] += 1;
}

struct AdamP {
  float* p; const float* g; float* m; float* v; long long n;
  float lr, b1, b2, eps, wd, bc1, bc2;
  const float* clip; bf16_t* shadow; const int* state;
};

__device__ __forceinline__ float adam1(float& p, float g, float& m, float& v, const AdamP& a, float cc) {
  g *= cc;
  p *= (1.f - a.lr * a.wd);
  m = a.b1 * m + (1.f - a.b1) * g;
  v = a.b2 * v + (1.f - a.b2) * g * g;
  const float denom = sqrtf(v) / sqrtf(a.bc2) + a.eps;
  p -= (a.lr / a.bc1) * (m / denom);
  return p;
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamP a) {
  const float cc = a.clip ? a.clip[0] : 1.f;
  if (a.state) {  // guarded update: coef == 0 marks a non-finite gradient norm (skip); the step count lives on the device
    if (!(cc > 0.f)) return;
    const double step = (double)a.state[0];
    a.bc1 = (float)(1.0 - pow((double)a.b1, step));
    a.bc2 = (float)(1.0 - pow((double)a.b2, step));
  }
  const long long n4 = a.n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 p = reinterpret_cast<float4*>(a.p)[i];
    const float4 g = reinterpret_cast<const float4*>(a.g)[i];
    float4 m = reinterpret_cast<float4*>(a.m)[i];
    float4 v = reinterpret_cast<float4*>(a.v)[i];
    adam1(p.x, g.x, m.x, v.x, a, cc); adam1(p.y, g.y, m.y, v.y, a, cc);
    adam1(p.z, g.z, m.z, v.z, a, cc); adam1(p.w, g.w, m.w, v.w, a, cc);
    reinterpret_cast<float4*>(a.p)[i] = p;
    reinterpret_cast<float4*>(a.m)[i] = m;
    reinterpret_cast<float4*>(a.v)[i] = v;
    if (a.shadow) {
      uint2 u; u.x = pack_bf2(p.x, p.y); u.y = pack_bf2(p.z, p.w);
      reinterpret_cast<uint2*>(a.shadow)[i] = u;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
    const long long j = n4 * 4 + threadIdx.x;
    const float r = adam1(a.p[j], a.g[j], a.m[j], a.v[j], a, cc);
    if (a.shadow) a.shadow[j] = f2bf(r);
  }
}

struct AdamGroupsP {
  float* p; const float* g; float* m; float* v;
  float b1, b2, eps;
  const float* clip; bf16_t* shadow; const int* state;
  int n_groups;
  long long total4;                       // float4 count of all groups together
  long long first4[CINEMA_ADAMW_MAX_GROUPS + 1];  // prefix sums of the groups' float4 counts
  long long begin4[CINEMA_ADAMW_MAX_GROUPS];
  float lr[CINEMA_ADAMW_MAX_GROUPS], wd[CINEMA_ADAMW_MAX_GROUPS];
};

// several parameter groups (own lr / weight decay) of one flat buffer in one launch: the float4s of all groups are numbered consecutively, a thread walks its
// grid-stride sequence and moves on through the (ascending) groups as it goes
__global__ __launch_bounds__(256) void adamw_groups_kernel(AdamGroupsP q) {
  const float cc = q.clip ? q.clip[0] : 1.f;
  if (!(cc > 0.f)) return;  // non-finite gradient norm: skip (clip_coef_kernel counted it)
  AdamP a{q.p, q.g, q.m, q.v, 0, 0.f, q.b1, q.b2, q.eps, 0.f, 1.f, 1.f, q.clip, q.shadow, q.state};
  const double step = (double)q.state[0];
  a.bc1 = (float)(1.0 - pow((double)a.b1, step));
  a.bc2 = (float)(1.0 - pow((double)a.b2, step));
  int gi = 0;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < q.total4; t += (long long)gridDim.x * blockDim.x) {
    while (t >= q.first4[gi + 1]) gi++;
    a.lr = q.lr[gi]; a.wd = q.wd[gi];
    const long long i = q.begin4[gi] + (t - q.first4[gi]);
    float4 p = reinterpret_cast<float4*>(a.p)[i];
    const float4 g = reinterpret_cast<const float4*>(a.g)[i];
    float4 m = reinterpret_cast<float4*>(a.m)[i];
    float4 v = reinterpret_cast<float4*>(a.v)[i];
    adam1(p.x, g.x, m.x, v.x, a, cc); adam1(p.y, g.y, m.y, v.y, a, cc);
    adam1(p.z, g.z, m.z, v.z, a, cc); adam1(p.w, g.w, m.w, v.w, a, cc);
    reinterpret_cast<float4*>(a.p)[i] = p;
    reinterpret_cast<float4*>(a.m)[i] = m;
    reinterpret_cast<float4*>(a.v)[i] = v;
    if (a.shadow) {
      uint2 u; u.x = pack_bf2(p.x, p.y); u.y = pack_bf2(p.z, p.w);
      reinterpret_cast<uint2*>(a.shadow)[i] = u;
    }
  }
}

}  // namespace

CINEMA_API int cinema_adamw_groups_grid(float* p, const float* g, float* m, float* v, const cinema_adamw_group* groups, int n_groups, float beta1, float beta2,
                                        float eps, const float* clip_coef, uint16_t* p_bf16, const int* step_state, int max_blocks, void* stream) {
  if (!p || !g || !m || !v || !groups || n_groups < 1 || n_groups > CINEMA_ADAMW_MAX_GROUPS || !clip_coef || !step_state) return CINEMA_ERR_BAD_ARG;
  if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return CINEMA_ERR_UNSUPPORTED;
  if (p_bf16 && (((uintptr_t)p_bf16) & 7)) return CINEMA_ERR_UNSUPPORTED;
  AdamGroupsP q{};
  q.p = p; q.g = g; q.m = m; q.v = v; q.b1 = beta1; q.b2 = beta2; q.eps = eps; q.clip = clip_coef; q.shadow = p_bf16; q.state = step_state; q.n_groups = n_groups;
  long long prev_end = 0;
  q.first4[0] = 0;
  for (int i = 0; i < n_groups; i++) {
    const cinema_adamw_group& gr = groups[i];
    if (gr.begin < prev_end || gr.end < gr.begin || (gr.begin & 3) || (gr.end & 3)) return CINEMA_ERR_BAD_ARG;
    q.begin4[i] = gr.begin >> 2;
    q.first4[i + 1] = q.first4[i] + ((gr.end - gr.begin) >> 2);
    q.lr[i] = gr.lr; q.wd[i] = gr.weight_decay;
    prev_end = gr.end;
  }
  q.total4 = q.first4[n_groups];
  if (q.total4 == 0) return 0;
  long long grid = (q.total4 + 255) / 256;
  const long long cap = max_blocks > 0 ? max_blocks : 4096;
  if (grid > cap) grid = cap;
  CINEMA_LAUNCH(adamw_groups_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, q);
  return launch_status();
}

CINEMA_API int cinema_adamw_groups(float* p, const float* g, float* m, float* v, const cinema_adamw_group* groups, int n_groups, float beta1, float beta2, float eps,
                                   const float* clip_coef, uint16_t* p_bf16, const int* step_state, void* stream) {
  return cinema_adamw_groups_grid(p, g, m, v, groups, n_groups, beta1, beta2, eps, clip_coef, p_bf16, step_state, 0, stream);
}

CINEMA_API int cinema_sqnorm_f32(const float* g, long long n, float* out, float* workspace, void* stream) {
  if (!g || !out || !workspace || n <= 0) return CINEMA_ERR_BAD_ARG;
  if (((uintptr_t)g) & 15) return CINEMA_ERR_UNSUPPORTED;
  long long grid = (n / 4 + 255) / 256;
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  CINEMA_LAUNCH(sqnorm_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, g, n, workspace);
  CINEMA_LAUNCH(sqnorm_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (int)grid, out);
  return launch_status();
}

CINEMA_API int cinema_clip_coef(const float* sqnorm, float max_norm, float* coef_out, float* norm_out, int* step_state, void* stream) {
  if (!sqnorm) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(clip_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sqnorm, max_norm, coef_out, norm_out, step_state);
  return launch_status();
}

CINEMA_API int cinema_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                            float weight_decay, float bias_corr1, float bias_corr2, const float* clip_coef, uint16_t* p_bf16, const int* step_state,
                            void* stream) {
  if (!p || !g || !m || !v || n <= 0) return CINEMA_ERR_BAD_ARG;
  if (step_state && !clip_coef) return CINEMA_ERR_BAD_ARG;
  if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return CINEMA_ERR_UNSUPPORTED;
  if (p_bf16 && (((uintptr_t)p_bf16) & 7)) return CINEMA_ERR_UNSUPPORTED;
  AdamP a{p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, clip_coef, p_bf16, step_state};
  long long grid = (n / 4 + 255) / 256;
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  CINEMA_LAUNCH(adamw_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}
