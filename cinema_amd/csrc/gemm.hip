// bf16 GEMM with fused epilogues for gfx950: MFMA 32x32x16 tiles, register-staged double-buffered LDS,
// XOR-swizzled K-major tiles, ds_read_b64_tr_b16 transpose reads for reduction-strided operands
// (dgrad / wgrad), fp32-atomic split-K for the weight gradients.  See include/cinema_hip.h for the contract.
#include "common.cuh"
#include "../../include/cinema_hip.h"

namespace {

struct GemmP {
  const bf16_t* a; const bf16_t* b; void* d;
  int m, n, k, lda, ldb, ldd;
  float alpha;
  const float* bias;
  const float* res_f32; const bf16_t* res_bf16; int ld_res;
  const bf16_t* gelu_in; int ld_gelu;
  const uint8_t* row_mask;
  bf16_t* aux_out; int ld_aux;
  int act, out_f32, accumulate;
  int ktiles_per_split;
};

// ---- epilogue on 4 consecutive columns (n0..n0+3) of row m; n0 % 4 == 0 and n0+3 < N guaranteed by the caller
__device__ __forceinline__ void epilogue4(const GemmP& p, int m, int n0, float v0, float v1, float v2, float v3, bool add_bias) {
  float v[4] = {v0 * p.alpha, v1 * p.alpha, v2 * p.alpha, v3 * p.alpha};
  if (p.bias && add_bias) {
    const float4 bv = *reinterpret_cast<const float4*>(p.bias + n0);
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
  }
  if (p.aux_out) {
    uint2 pk; pk.x = pack_bf2(v[0], v[1]); pk.y = pack_bf2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p.aux_out + (size_t)m * p.ld_aux + n0) = pk;
  }
  if (p.act == 1) {
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = gelu_f(v[i]);
  }
  if (p.gelu_in) {
    const uint2 gi = *reinterpret_cast<const uint2*>(p.gelu_in + (size_t)m * p.ld_gelu + n0);
    v[0] *= gelu_grad_f(bf2f((bf16_t)(gi.x & 0xffff))); v[1] *= gelu_grad_f(bf2f((bf16_t)(gi.x >> 16)));
    v[2] *= gelu_grad_f(bf2f((bf16_t)(gi.y & 0xffff))); v[3] *= gelu_grad_f(bf2f((bf16_t)(gi.y >> 16)));
  }
  if (p.row_mask) {
    const float s = p.row_mask[m] ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] *= s;
  }
  if (p.res_f32) {
    const float4 rv = *reinterpret_cast<const float4*>(p.res_f32 + (size_t)m * p.ld_res + n0);
    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
  } else if (p.res_bf16) {
    const uint2 rv = *reinterpret_cast<const uint2*>(p.res_bf16 + (size_t)m * p.ld_res + n0);
    v[0] += bf2f((bf16_t)(rv.x & 0xffff)); v[1] += bf2f((bf16_t)(rv.x >> 16));
    v[2] += bf2f((bf16_t)(rv.y & 0xffff)); v[3] += bf2f((bf16_t)(rv.y >> 16));
  }
  if (p.out_f32) {
    float* dp = reinterpret_cast<float*>(p.d) + (size_t)m * p.ldd + n0;
    if (p.accumulate) {
#pragma unroll
      for (int i = 0; i < 4; i++) unsafeAtomicAdd(dp + i, v[i]);
    } else {
      *reinterpret_cast<float4*>(dp) = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else {
    uint2 pk; pk.x = pack_bf2(v[0], v[1]); pk.y = pack_bf2(v[2], v[3]);
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.d) + (size_t)m * p.ldd + n0) = pk;
  }
}

// scalar epilogue for the generic kernel
__device__ __forceinline__ void epilogue1(const GemmP& p, int m, int n, float acc, bool add_bias) {
  float v = acc * p.alpha;
  if (p.bias && add_bias) v += p.bias[n];
  if (p.aux_out) p.aux_out[(size_t)m * p.ld_aux + n] = f2bf(v);
  if (p.act == 1) v = gelu_f(v);
  if (p.gelu_in) v *= gelu_grad_f(bf2f(p.gelu_in[(size_t)m * p.ld_gelu + n]));
  if (p.row_mask) v *= p.row_mask[m] ? 1.f : 0.f;
  if (p.res_f32) v += p.res_f32[(size_t)m * p.ld_res + n];
  else if (p.res_bf16) v += bf2f(p.res_bf16[(size_t)m * p.ld_res + n]);
  if (p.out_f32) {
    float* dp = reinterpret_cast<float*>(p.d) + (size_t)m * p.ldd + n;
    if (p.accumulate) unsafeAtomicAdd(dp, v); else *dp = v;
  } else {
    reinterpret_cast<bf16_t*>(p.d)[(size_t)m * p.ldd + n] = f2bf(v);
  }
}

// ------------------------------------------------------------------------------------------------
// MFMA kernel: 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32x16 tiles.
// The MFMA is issued with the B-tile fragment as its first operand (rows = n) and the A-tile fragment
// as its second (cols = m): accumulator reg r of lane l holds D[m = l&31][n = (r&3)+8*(r>>2)+4*(l>>5)],
// i.e. 4 consecutive n per register quad -> 8/16-byte epilogue accesses.
// ------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int KMAJ_BYTES = 128 * BK * 2;        // [128 rows][64 k] bf16, 128-byte rows, XOR swizzled
constexpr int MNMAJ_STRIDE = 128 * 2 + 64;      // [64 k rows][128 cols] bf16 + 64 B pad (tr-read conflict-free)
constexpr int MNMAJ_BYTES = BK * MNMAJ_STRIDE;

template <bool KMAJ>
struct TileIO {
  // global -> registers (4 x 16 B per thread) for the 128(rows of M or N) x 64(k) operand tile
  static __device__ __forceinline__ void load(uint4 (&r)[4], const bf16_t* base, int ld, int row0, int nrows, int k0, int kdim, int tid) {
    if (KMAJ) {
      const int chunk = tid & 7;
      const int kk = k0 + chunk * 8;
#pragma unroll
      for (int pss = 0; pss < 4; pss++) {
        int row = row0 + pss * 32 + (tid >> 3);
        row = row < nrows ? row : nrows - 1;
        if (kk < kdim) r[pss] = *reinterpret_cast<const uint4*>(base + (size_t)row * ld + kk);
        else r[pss] = make_uint4(0, 0, 0, 0);
      }
    } else {
      const int chunk = tid & 15;
      int col = row0 + chunk * 8;
      col = col < nrows ? col : 0;
#pragma unroll
      for (int pss = 0; pss < 4; pss++) {
        const int kr = k0 + pss * 16 + (tid >> 4);
        if (kr < kdim) r[pss] = *reinterpret_cast<const uint4*>(base + (size_t)kr * ld + col);
        else r[pss] = make_uint4(0, 0, 0, 0);
      }
    }
  }
  // registers -> LDS
  static __device__ __forceinline__ void store(const uint4 (&r)[4], char* lds, int tid) {
    if (KMAJ) {
      const int chunk = tid & 7;
#pragma unroll
      for (int pss = 0; pss < 4; pss++) {
        const int row = pss * 32 + (tid >> 3);
        *reinterpret_cast<uint4*>(lds + swz_off<128>(row, chunk)) = r[pss];
      }
    } else {
      const int chunk = tid & 15;
#pragma unroll
      for (int pss = 0; pss < 4; pss++) {
        const int kr = pss * 16 + (tid >> 4);
        *reinterpret_cast<uint4*>(lds + kr * MNMAJ_STRIDE + chunk * 16) = r[pss];
      }
    }
  }
  // MFMA operand fragment: 32 rows starting at `base`, 16 k starting at ks*16; lane l -> row base+(l&31), k-group l>>5
  static __device__ __forceinline__ short8v frag(const char* lds, int base, int ks, int lane) {
    if (KMAJ) {
      const int row = base + (lane & 31);
      return *reinterpret_cast<const short8v*>(lds + swz_off<128>(row, ks * 2 + (lane >> 5)));
    } else {
      const int q4 = lane >> 4, t = lane & 15;
      const int col = base + 16 * (q4 & 1) + 4 * (t & 3);
      const int kr = ks * 16 + 8 * (q4 >> 1) + (t >> 2);
      const short4v lo = lds_tr16_b64(lds + kr * MNMAJ_STRIDE + col * 2);
      const short4v hi = lds_tr16_b64(lds + (kr + 4) * MNMAJ_STRIDE + col * 2);
      short8v out;
      out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; out[3] = lo[3];
      out[4] = hi[0]; out[5] = hi[1]; out[6] = hi[2]; out[7] = hi[3];
      return out;
    }
  }
  static constexpr int BYTES = KMAJ ? KMAJ_BYTES : MNMAJ_BYTES;
};

template <bool A_KMAJ, bool B_KMAJ>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(GemmP p) {
  using AIO = TileIO<A_KMAJ>;
  using BIO = TileIO<B_KMAJ>;
  constexpr int STAGE = AIO::BYTES + BIO::BYTES;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int tiles_n = (p.n + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int nkt = (p.k + BK - 1) / BK;
  const int kt_begin = blockIdx.z * p.ktiles_per_split;
  const int kt_end = min(nkt, kt_begin + p.ktiles_per_split);
  if (kt_begin >= kt_end) return;

  float16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  uint4 ra[4], rb[4];
  AIO::load(ra, p.a, p.lda, m0, p.m, kt_begin * BK, p.k, tid);
  BIO::load(rb, p.b, p.ldb, n0, p.n, kt_begin * BK, p.k, tid);
  AIO::store(ra, smem, tid);
  BIO::store(rb, smem + AIO::BYTES, tid);
  __syncthreads();

  for (int kt = kt_begin; kt < kt_end; kt++) {
    const int cur = (kt - kt_begin) & 1;
    const char* sa = smem + cur * STAGE;
    const char* sb = sa + AIO::BYTES;
    const bool more = kt + 1 < kt_end;
    if (more) {
      AIO::load(ra, p.a, p.lda, m0, p.m, (kt + 1) * BK, p.k, tid);
      BIO::load(rb, p.b, p.ldb, n0, p.n, (kt + 1) * BK, p.k, tid);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      short8v fa[2], fb[2];
      fa[0] = AIO::frag(sa, wm, ks, lane);
      fa[1] = AIO::frag(sa, wm + 32, ks, lane);
      fb[0] = BIO::frag(sb, wn, ks, lane);
      fb[1] = BIO::frag(sb, wn + 32, ks, lane);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    if (more) {
      char* na = smem + (cur ^ 1) * STAGE;
      AIO::store(ra, na, tid);
      BIO::store(rb, na + AIO::BYTES, tid);
    }
    __syncthreads();
  }

  const bool add_bias = blockIdx.z == 0;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int m = m0 + wm + i * 32 + (lane & 31);
    if (m >= p.m) continue;
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int n = n0 + wn + j * 32 + 8 * q + 4 * (lane >> 5);
        if (n < p.n) epilogue4(p, m, n, acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3], add_bias);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Generic kernel: any shape / alignment, fp32 FMA on bf16 inputs, 64x64 tile, 16x16 threads x (4x4).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmP p, int a_rs, int a_cs, int b_rs, int b_cs) {
  __shared__ float sa[16][65];
  __shared__ float sb[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int tiles_n = (p.n + 63) / 64;
  const int m0 = (blockIdx.x / tiles_n) * 64, n0 = (blockIdx.x % tiles_n) * 64;
  const int nkt = (p.k + 15) / 16;
  const int kt_begin = blockIdx.z * p.ktiles_per_split, kt_end = min(nkt, kt_begin + p.ktiles_per_split);
  if (kt_begin >= kt_end) return;
  float acc[4][4] = {};
  for (int kt = kt_begin; kt < kt_end; kt++) {
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      const int r = i >> 4, kk = i & 15;
      const int m = m0 + r, n = n0 + r, k = kt * 16 + kk;
      sa[kk][r] = (m < p.m && k < p.k) ? bf2f(p.a[(size_t)m * a_rs + (size_t)k * a_cs]) : 0.f;
      sb[kk][r] = (n < p.n && k < p.k) ? bf2f(p.b[(size_t)k * b_rs + (size_t)n * b_cs]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { av[i] = sa[kk][ty * 4 + i]; bv[i] = sb[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m < p.m && n < p.n) epilogue1(p, m, n, acc[i][j], blockIdx.z == 0);
    }
}

__global__ void colsum_kernel(const void* x, int is_f32, const int* row_idx, int m, int n, int ldx, float* out, int rows_per_block) {
  // block: 64 columns x 4 row-lanes; grid.x = column groups, grid.y = row chunks
  __shared__ float part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(m, r0 + rows_per_block);
  float s = 0.f;
  if (col < n)
    for (int r = r0 + rl; r < r1; r += 4) {
      const size_t off = (size_t)(row_idx ? row_idx[r] : r) * ldx + col;
      s += is_f32 ? reinterpret_cast<const float*>(x)[off] : bf2f(reinterpret_cast<const bf16_t*>(x)[off]);
    }
  part[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && col < n) unsafeAtomicAdd(out + col, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

}  // namespace

CINEMA_API int cinema_gemm_bf16(cinema_gemm_args* a, void* stream) {
  if (!a || !a->a || !a->b || !a->d || a->m <= 0 || a->n <= 0 || a->k <= 0) return CINEMA_ERR_BAD_ARG;
  if (a->accumulate && !a->out_f32) return CINEMA_ERR_BAD_ARG;
  const int split = a->split_k < 1 ? 1 : a->split_k;
  if (split > 1 && !a->accumulate) return CINEMA_ERR_BAD_ARG;
  GemmP p;
  p.a = (const bf16_t*)a->a; p.b = (const bf16_t*)a->b; p.d = a->d;
  p.m = a->m; p.n = a->n; p.k = a->k; p.lda = a->lda; p.ldb = a->ldb; p.ldd = a->ldd;
  p.alpha = a->alpha;
  p.bias = a->bias; p.res_f32 = a->residual_f32; p.res_bf16 = a->residual_bf16; p.ld_res = a->ld_res;
  p.gelu_in = a->gelu_in; p.ld_gelu = a->ld_gelu; p.row_mask = a->row_mask; p.aux_out = a->aux_out; p.ld_aux = a->ld_aux;
  p.act = a->act; p.out_f32 = a->out_f32; p.accumulate = a->accumulate;
  hipStream_t st = (hipStream_t)stream;

  auto al8 = [](int v) { return (v & 7) == 0; };
  auto ptr16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  bool fast = !a->force_generic && al8(a->lda) && al8(a->ldb) && al8(a->ldd) && al8(a->n) && ptr16(a->a) && ptr16(a->b) && ptr16(a->d);
  fast = fast && (a->a_kmajor ? al8(a->k) : al8(a->m)) && (a->b_kmajor ? al8(a->k) : true);
  fast = fast && (!a->bias || ptr16(a->bias)) && (!a->residual_f32 || (al8(a->ld_res) && ptr16(a->residual_f32)));
  fast = fast && (!a->residual_bf16 || (al8(a->ld_res) && ptr16(a->residual_bf16))) && (!a->gelu_in || (al8(a->ld_gelu) && ptr16(a->gelu_in)));
  fast = fast && (!a->aux_out || (al8(a->ld_aux) && ptr16(a->aux_out)));
  fast = fast && !(a->a_kmajor == 0 && a->b_kmajor == 1);  // (M-major A, K-major B) is not used by the path
  if (fast) {
    const int nkt = (a->k + BK - 1) / BK;
    const int sp = split > nkt ? nkt : split;
    p.ktiles_per_split = (nkt + sp - 1) / sp;
    const int gz = (nkt + p.ktiles_per_split - 1) / p.ktiles_per_split;
    dim3 grid(((a->m + BM - 1) / BM) * ((a->n + BN - 1) / BN), 1, gz);
    a->kernel_used = (a->a_kmajor && a->b_kmajor) ? 1 : (a->a_kmajor ? 2 : 3);
    if (a->a_kmajor && a->b_kmajor) hipLaunchKernelGGL((gemm_mfma_kernel<true, true>), grid, dim3(256), 0, st, p);
    else if (a->a_kmajor && !a->b_kmajor) hipLaunchKernelGGL((gemm_mfma_kernel<true, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_mfma_kernel<false, false>), grid, dim3(256), 0, st, p);
    return launch_status();
  }
  const int nkt = (a->k + 15) / 16;
  const int sp = split > nkt ? nkt : split;
  p.ktiles_per_split = (nkt + sp - 1) / sp;
  const int gz = (nkt + p.ktiles_per_split - 1) / p.ktiles_per_split;
  dim3 grid(((a->m + 63) / 64) * ((a->n + 63) / 64), 1, gz);
  const int a_rs = a->a_kmajor ? a->lda : 1, a_cs = a->a_kmajor ? 1 : a->lda;   // element (m,k) = a[m*a_rs + k*a_cs]
  const int b_rs = a->b_kmajor ? 1 : a->ldb, b_cs = a->b_kmajor ? a->ldb : 1;   // element (k,n) = b[k*b_rs + n*b_cs]
  a->kernel_used = 0;
  hipLaunchKernelGGL(gemm_generic_kernel, grid, dim3(256), 0, st, p, a_rs, a_cs, b_rs, b_cs);
  return launch_status();
}

CINEMA_API int cinema_colsum(const void* x, int x_dtype, const int* row_idx, int m, int n, int ldx, float* out, void* stream) {
  if (!x || !out || m <= 0 || n <= 0) return CINEMA_ERR_BAD_ARG;
  int chunks = (m + 511) / 512;
  if (chunks > 256) chunks = 256;
  const int rpb = (m + chunks - 1) / chunks;
  dim3 grid((n + 63) / 64, (m + rpb - 1) / rpb);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, x_dtype, row_idx, m, n, ldx, out, rpb);
  return launch_status();
}
