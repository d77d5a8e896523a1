// Depthwise 5^n convolution of the conv stem on the visible voxels, TOKEN-PAIR form (reference: cinema/conv.py:385,410-413 - dw_conv over mask * conv1(...),
// evaluated at the visible voxels; see csrc/sparse_conv.hip for why that is exact).
//
// The mask is constant over a token and a token is a B x B x 1 block of stage voxels (B = 4 at stage 1, 2 at stage 2; the long-axis views are the same with the
// thin axis of extent 1), stored as B*B consecutive compact rows.  So instead of a per-VOXEL neighbour list (sparse_nbr_build_kernel: 125 rank lookups per voxel,
// 512 B of list per voxel, a gather driven by it) a wave walks the 3 x 3 x 5 neighbour TOKENS of a kept token: one rank lookup per neighbour, a visible
// neighbour's whole block (B*B rows, one channel per lane: 128 contiguous bytes per row and wave) goes into registers, and the taps that connect the two blocks
// are compile-time constants - straight-line packed FMAs, every weight read once per (neighbour, tap) from the LDS slab.  No list, no barrier in the loop.
//   forward / data gradient (flipped taps): thread = one channel of one token, B*B fp32 outputs in registers.
//   weight gradient: thread = (channel, thin-axis tap), 25 in-plane tap accumulators in registers over a chunk of tokens, one slab per workgroup,
//   ordered reduce (bit-identical run to run).
#include "common.cuh"
#include "../../include/cinema_hip.h"
#include <utility>

namespace {

// compile-time loops: the register arrays below are only ever subscripted with constants IN THE SOURCE.  With `#pragma unroll` loops the first scalar-replacement
// pass of hipcc 7.2 still sees run-time subscripts, turns float[16] / float[25] into <16 x float> / <25 x float> VALUES and the register allocator then moves whole
// 32-register tuples around (600-700 bytes of scratch per lane under an 80-register budget).
template <int... I, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(std::make_integer_sequence<int, N>{}, f); }
#define SV(x) decltype(x)::value

struct DwP {
  const bf16_t* x;      // compact rows [n_tok * B*B][c] (forward: input; data gradient: dy; weight gradient: input)
  const bf16_t* dy;     // weight gradient only
  const float* w;       // [c][taps]
  const float* bias;    // [c] or null
  bf16_t* y;
  float* slab;          // weight gradient: [workgroups][c * taps + c]
  const int* keep; const int* rank; const int* pos;
  int c, n_tok, gu, gv, gw;  // channels (row pitch); token grid per sample in canonical axes (u, v in-plane, w thin)
  int flip, tok_per_wg;
};

// A lane owns ONE channel and a workgroup 64 of them (blockIdx.y = channel group): the loop over the neighbour tokens is a chain of dependent loads (rank ->
// rows) per visible neighbour, so what it needs is many waves per CU, i.e. few registers per lane (B*B outputs + B*B neighbour values) and a small weight slab.
struct Tok { int bb, tu, tv, tw; };
__device__ __forceinline__ Tok token_of(const DwP& p, int r) {
  const int T = p.gu * p.gv * p.gw;
  const int id = p.keep[r];
  const int bb = id / T, t = id - bb * T;
  return {bb, t / (p.gw * p.gv), (t / p.gw) % p.gv, t % p.gw};
}

constexpr int FWD_WAVES = 8;
// y[token block] = bias + sum over visible neighbour tokens of the taps that connect the two blocks
template <int B, int KW>
__device__ __forceinline__ void dw_fwd_body(const DwP& p) {
  constexpr int BV = B * B, TAPS = 25 * KW, RW = KW / 2, NTH = FWD_WAVES * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WP = 65;                       // slab pitch in floats: the transposing store below (lanes = consecutive taps) spreads over the banks
  float* wl = reinterpret_cast<float*>(smem);  // [TAPS][WP]
  const int tid = threadIdx.x, c0 = blockIdx.y * 64;
  for (int i0 = tid; i0 < TAPS * 64; i0 += NTH * 4) {  // coalesced read of w[c0 + c][taps], four loads in flight per thread
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const int i = i0 + q * NTH; v[q] = i < TAPS * 64 ? p.w[(size_t)c0 * TAPS + i] : 0.f; }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int i = i0 + q * NTH;
      if (i < TAPS * 64) { const int c = i / TAPS, t = i - c * TAPS; wl[(p.flip ? TAPS - 1 - t : t) * WP + c] = v[q]; }
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, ch = c0 + lane;
  const int T = p.gu * p.gv * p.gw;
  int pos_s[BV];
  sfor<BV>([&](auto v) { pos_s[SV(v)] = p.pos[SV(v)] * p.c; });
  const float bias = p.bias ? p.bias[ch] : 0.f;
  for (int r = blockIdx.x * FWD_WAVES + wave; r < p.n_tok; r += gridDim.x * FWD_WAVES) {
    const Tok tk = token_of(p, r);
    float out[BV];
    sfor<BV>([&](auto v) { out[SV(v)] = bias; });
    // the rank of every neighbour token in ONE load (lane j looks up neighbour j): 45 dependent lookups, each in front of a branch, were 20-40 us per token
    int rkv = -1;
    if (lane < 9 * KW) {
      const int nw = tk.tw + lane / 9 - RW, nu = tk.tu + (lane % 9) / 3 - 1, nv = tk.tv + lane % 3 - 1;
      if (nw >= 0 && nw < p.gw && nu >= 0 && nu < p.gu && nv >= 0 && nv < p.gv) rkv = p.rank[tk.bb * T + (nu * p.gv + nv) * p.gw + nw];
    }
    const unsigned long long vis = __ballot(rkv >= 0);
    sfor<KW>([&](auto kwi) {
      constexpr int dw = SV(kwi) - RW;
      sfor<9>([&](auto cell) {
        constexpr int du = SV(cell) / 3 - 1, dv = SV(cell) % 3 - 1, j = SV(kwi) * 9 + SV(cell);
        if (!((vis >> j) & 1ull)) return;
        const int rk = __builtin_amdgcn_readlane(rkv, j);
        const bf16_t* src = p.x + (size_t)rk * BV * p.c + ch;
        float h[BV];
        sfor<BV>([&](auto v) { h[SV(v)] = bf2f(src[pos_s[SV(v)]]); });
        sfor<25>([&](auto tap) {
          // output voxel (vu, vv) reads neighbour voxel (vu + a - 2 - du B, vv + b - 2 - dv B) of this neighbour token, if that lies inside its block
          constexpr int a = SV(tap) / 5, b = SV(tap) % 5, ou = a - 2 - du * B, ov = b - 2 - dv * B;
          if constexpr (ou > -B && ou < B && ov > -B && ov < B) {
            const float wv = wl[(SV(tap) * KW + dw + RW) * WP + lane];
            sfor<BV>([&](auto o) {
              constexpr int vu = SV(o) / B, vv = SV(o) % B;
              if constexpr (vu + ou >= 0 && vu + ou < B && vv + ov >= 0 && vv + ov < B) out[SV(o)] = fmaf(wv, h[(vu + ou) * B + vv + ov], out[SV(o)]);
            });
          }
        });
      });
    });
    bf16_t* dst = p.y + (size_t)r * BV * p.c + ch;
    sfor<BV>([&](auto v) { dst[pos_s[SV(v)]] = f2bf(out[SV(v)]); });
  }
}
template <int B, int KW> __global__ __launch_bounds__(FWD_WAVES * 64, B == 4 ? 4 : 6) void stem_dw_fwd_kernel(DwP p) { dw_fwd_body<B, KW>(p); }
template <int B, int KW> __global__ __launch_bounds__(FWD_WAVES * 64, B == 4 ? 4 : 6) void stem_dw_fwd_lanes_kernel(Lanes<DwP> L) { dw_fwd_body<B, KW>(L.p[blockIdx.z]); }

// dw[c][tap] += sum over visible (voxel, neighbour voxel) pairs dy[voxel][c] * x[neighbour][c];  db[c] += sum dy.  wave = (thin-axis tap, token group)
template <int B, int KW, int TG>
__device__ __forceinline__ void dw_wgrad_body(const DwP& p) {
  constexpr int BV = B * B, TAPS = 25 * KW, RW = KW / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [TG - 1][KW][25 + 1][64]  (token groups 1.. hand their sums to group 0)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c0 = blockIdx.y * 64, ch = c0 + lane;
  const int kwi = wave % KW, tg = wave / KW, dw = kwi - RW;
  const int T = p.gu * p.gv * p.gw;
  int pos_s[BV];
  sfor<BV>([&](auto v) { pos_s[SV(v)] = p.pos[SV(v)] * p.c; });
  float acc[25];
  sfor<25>([&](auto t) { acc[SV(t)] = 0.f; });
  float accb = 0.f;
  const int r_begin = blockIdx.x * p.tok_per_wg, r_end = min(p.n_tok, r_begin + p.tok_per_wg);
  for (int r = r_begin + tg; r < r_end; r += TG) {
    const Tok tk = token_of(p, r);
    const int nw = tk.tw + dw;
    if (nw < 0 || nw >= p.gw) continue;
    float d[BV];
    {
      const bf16_t* src = p.dy + (size_t)r * BV * p.c + ch;
      sfor<BV>([&](auto v) { d[SV(v)] = bf2f(src[pos_s[SV(v)]]); });
    }
    if (kwi == RW) sfor<BV>([&](auto v) { accb += d[SV(v)]; });
    int rkv = -1;  // ranks of the nine in-plane neighbours in one load (lane j looks up neighbour j)
    if (lane < 9) {
      const int nu = tk.tu + lane / 3 - 1, nv = tk.tv + lane % 3 - 1;
      if (nu >= 0 && nu < p.gu && nv >= 0 && nv < p.gv) rkv = p.rank[tk.bb * T + (nu * p.gv + nv) * p.gw + nw];
    }
    const unsigned long long vis = __ballot(rkv >= 0);
    sfor<9>([&](auto cell) {
      constexpr int du = SV(cell) / 3 - 1, dv = SV(cell) % 3 - 1;
      if (!((vis >> SV(cell)) & 1ull)) return;
      const int rk = __builtin_amdgcn_readlane(rkv, SV(cell));
      const bf16_t* src = p.x + (size_t)rk * BV * p.c + ch;
      float h[BV];
      sfor<BV>([&](auto v) { h[SV(v)] = bf2f(src[pos_s[SV(v)]]); });
      sfor<25>([&](auto tap) {
        constexpr int a = SV(tap) / 5, b = SV(tap) % 5, ou = a - 2 - du * B, ov = b - 2 - dv * B;
        if constexpr (ou > -B && ou < B && ov > -B && ov < B) {
          sfor<BV>([&](auto o) {
            constexpr int vu = SV(o) / B, vv = SV(o) % B;
            if constexpr (vu + ou >= 0 && vu + ou < B && vv + ov >= 0 && vv + ov < B) acc[SV(tap)] = fmaf(d[SV(o)], h[(vu + ou) * B + vv + ov], acc[SV(tap)]);
          });
        }
      });
    });
  }
  if (TG > 1) {
    if (tg > 0) {
      float* dst = red + (size_t)((tg - 1) * KW + kwi) * 26 * 64;
      sfor<25>([&](auto t) { dst[SV(t) * 64 + lane] = acc[SV(t)]; });
      dst[25 * 64 + lane] = accb;
    }
    __syncthreads();
    if (tg == 0) {
      for (int g = 1; g < TG; g++) {
        const float* src = red + (size_t)((g - 1) * KW + kwi) * 26 * 64;
        sfor<25>([&](auto t) { acc[SV(t)] += src[SV(t) * 64 + lane]; });
        accb += src[25 * 64 + lane];
      }
    }
  }
  if (tg == 0) {
    float* slab = p.slab + (size_t)blockIdx.x * (p.c * TAPS + p.c);
    sfor<25>([&](auto t) { slab[(size_t)ch * TAPS + SV(t) * KW + kwi] = acc[SV(t)]; });
    if (kwi == RW) slab[(size_t)p.c * TAPS + ch] = accb;
  }
}
template <int B, int KW, int TG> __global__ __launch_bounds__(KW* TG * 64, B == 4 ? 4 : 6) void stem_dw_wgrad_kernel(DwP p) { dw_wgrad_body<B, KW, TG>(p); }
template <int B, int KW, int TG> __global__ __launch_bounds__(KW* TG * 64, B == 4 ? 4 : 6) void stem_dw_wgrad_lanes_kernel(Lanes<DwP> L) { dw_wgrad_body<B, KW, TG>(L.p[blockIdx.z]); }

// dw[i] += sum_b slab[b][i] (i < c taps), db[j] += sum_b slab[b][c taps + j]: one thread per element, slabs in order
struct DwRedP { const float* slab; int n_slabs, n_w, n_b; float* dw; float* db; };
__device__ __forceinline__ void dw_reduce_body(const DwRedP& q) {  // block = 32 elements x 8 slab groups
  __shared__ float red[8][32];
  const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e, total = q.n_w + q.n_b;
  const int per = (q.n_slabs + 7) / 8, b0 = grp * per, b1 = min(q.n_slabs, b0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (i < total) {
    int b = b0;
    for (; b + 3 < b1; b += 4) {
      a0 += q.slab[(size_t)b * total + i]; a1 += q.slab[(size_t)(b + 1) * total + i]; a2 += q.slab[(size_t)(b + 2) * total + i]; a3 += q.slab[(size_t)(b + 3) * total + i];
    }
    for (; b < b1; b++) a0 += q.slab[(size_t)b * total + i];
  }
  red[grp][e] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && i < total) {
    const float t = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) + ((red[4][e] + red[5][e]) + (red[6][e] + red[7][e]));
    if (i < q.n_w) q.dw[i] += t;
    else if (q.db) q.db[i - q.n_w] += t;
  }
}
__global__ __launch_bounds__(256) void stem_dw_reduce_kernel(DwRedP q) { dw_reduce_body(q); }
__global__ __launch_bounds__(256) void stem_dw_reduce_lanes_kernel(Lanes<DwRedP> L) { dw_reduce_body(L.p[blockIdx.y]); }

int cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  return n;
}
// canonical geometry: two in-plane axes with a B x B block, one thin axis with block 1
struct Canon { int ok, B, KW, gu, gv, gw; };
Canon canon(const cinema_sparse_geom* g, int kx, int ky, int kz) {
  Canon c{};
  if (!g || !g->keep || !g->rank || !g->pos || g->n_tok <= 0) return c;
  if (g->bz == 1 && g->bx == g->by && kx == 5 && ky == 5 && (kz == 5 || kz == 1)) c = Canon{1, g->bx, kz, g->tx, g->ty, g->tz};         // short-axis volume
  else if (g->bx == 1 && g->tx == 1 && g->by == g->bz && kx == 1 && ky == 5 && kz == 5) c = Canon{1, g->by, 1, g->ty, g->tz, 1};       // 2-D view (leading axis 1)
  if (c.ok && c.B != 2 && c.B != 4) c.ok = 0;
  if (c.ok && c.KW == 1 && c.gw != 1) c.ok = 0;
  return c;
}
template <typename K1, typename K2>
int lds_attr(bool (&flags)[16], K1 single, K2 lanes, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (flags[dev]) return 0;
  if (hipFuncSetAttribute((const void*)single, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)lanes, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 1;
  flags[dev] = true;
  return 0;
}
template <int B, int KW>
int launch_fwd(const DwP& p, hipStream_t st) {
  static bool f[16] = {};
  constexpr int lds = 25 * KW * 65 * 4;
  if (lds_attr(f, stem_dw_fwd_kernel<B, KW>, stem_dw_fwd_lanes_kernel<B, KW>, lds)) return CINEMA_ERR_UNSUPPORTED;
  const int groups = p.c / 64;
  int grid = (p.n_tok + FWD_WAVES - 1) / FWD_WAVES;
  const int cap = (cus() * (B == 4 ? 2 : 4) + groups - 1) / groups;   // 8-wave workgroups resident per CU (registers; 33 KB slab each): every one stages its slab once and walks its tokens
  if (grid > cap) grid = cap;
  launch_lanes(stem_dw_fwd_kernel<B, KW>, stem_dw_fwd_lanes_kernel<B, KW>, 2, dim3(grid, groups), dim3(FWD_WAVES * 64), lds, st, p);
  return launch_status();
}
template <int B, int KW, int TG>
int launch_wgrad(DwP p, int n_wg, hipStream_t st) {
  static bool f[16] = {};
  constexpr int lds = (TG - 1) * KW * 26 * 64 * 4 + 16;
  if (lds_attr(f, stem_dw_wgrad_kernel<B, KW, TG>, stem_dw_wgrad_lanes_kernel<B, KW, TG>, lds)) return CINEMA_ERR_UNSUPPORTED;
  launch_lanes(stem_dw_wgrad_kernel<B, KW, TG>, stem_dw_wgrad_lanes_kernel<B, KW, TG>, 2, dim3(n_wg, p.c / 64), dim3(KW * TG * 64), lds, st, p);
  return launch_status();
}
int wgrad_workgroups(int n_tok, int c) {
  int n = (n_tok + 15) / 16;  // >= 16 tokens per workgroup
  const int cap = 2 * cus() * 64 / c;  // two 15-16-wave workgroups per CU over the channel groups (each leaves a slab row of 64 (taps + 1) floats)
  return n < 1 ? 1 : (n > cap ? cap : n);
}

}  // namespace

CINEMA_API int cinema_stem_dw_supported(const cinema_sparse_geom* geom, int c, int kx, int ky, int kz) {
  return (c == 64 || c == 128) && canon(geom, kx, ky, kz).ok;
}

CINEMA_API int cinema_stem_dw_fwd(const uint16_t* x, const float* w, const float* bias, uint16_t* y, const cinema_sparse_geom* geom, int c, int kx, int ky, int kz, int flip,
                                  void* stream) {
  if (!x || !w || !y) return CINEMA_ERR_BAD_ARG;
  const Canon cn = canon(geom, kx, ky, kz);
  if (!cn.ok || (c != 64 && c != 128)) return CINEMA_ERR_UNSUPPORTED;
  DwP p{};
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.keep = geom->keep; p.rank = geom->rank; p.pos = geom->pos; p.c = c;
  p.n_tok = geom->n_tok; p.gu = cn.gu; p.gv = cn.gv; p.gw = cn.gw; p.flip = flip;
  hipStream_t st = (hipStream_t)stream;
#define DW_FWD(BB, KK) if (cn.B == BB && cn.KW == KK) return launch_fwd<BB, KK>(p, st)
  DW_FWD(4, 5); DW_FWD(2, 5); DW_FWD(4, 1); DW_FWD(2, 1);
#undef DW_FWD
  return CINEMA_ERR_UNSUPPORTED;
}

CINEMA_API long long cinema_stem_dw_wgrad_workspace_bytes(int n_tok, int c, int kx, int ky, int kz) {
  return (long long)wgrad_workgroups(n_tok, c) * ((long long)c * kx * ky * kz + c) * 4;
}

CINEMA_API int cinema_stem_dw_bwd_weight(const uint16_t* x, const uint16_t* dy, float* dw, float* dbias, float* workspace, long long workspace_bytes,
                                         const cinema_sparse_geom* geom, int c, int kx, int ky, int kz, void* stream) {
  if (!x || !dy || !dw || !workspace) return CINEMA_ERR_BAD_ARG;
  const Canon cn = canon(geom, kx, ky, kz);
  if (!cn.ok || (c != 64 && c != 128)) return CINEMA_ERR_UNSUPPORTED;
  const int taps = kx * ky * kz;
  DwP p{};
  p.x = x; p.dy = dy; p.slab = workspace; p.keep = geom->keep; p.rank = geom->rank; p.pos = geom->pos; p.c = c;
  p.n_tok = geom->n_tok; p.gu = cn.gu; p.gv = cn.gv; p.gw = cn.gw;
  int n_wg = wgrad_workgroups(p.n_tok, c);
  p.tok_per_wg = (p.n_tok + n_wg - 1) / n_wg;
  n_wg = (p.n_tok + p.tok_per_wg - 1) / p.tok_per_wg;
  if (workspace_bytes < (long long)n_wg * ((long long)c * taps + c) * 4) return CINEMA_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  int rc = CINEMA_ERR_UNSUPPORTED;
#define DW_WG(BB, KK, TT) if (cn.B == BB && cn.KW == KK) rc = launch_wgrad<BB, KK, TT>(p, n_wg, st)
  DW_WG(4, 5, 3); DW_WG(2, 5, 3); DW_WG(4, 1, 8); DW_WG(2, 1, 8);
#undef DW_WG
  if (rc) return rc;
  const int total = c * taps + c;
  launch_lanes(stem_dw_reduce_kernel, stem_dw_reduce_lanes_kernel, 1, dim3((total + 31) / 32), dim3(256), 0, st, DwRedP{workspace, n_wg, c * taps, c, dw, dbias});
  return launch_status();
}
