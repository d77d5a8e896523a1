// Depthwise 5^n convolution of the conv stem restricted to the VISIBLE voxels of an MAE step.
//
// The reference runs its masked conv blocks densely (cinema/conv.py:349-415): y = dwconv(mask * conv1(LN(x))), and only the
// kept tokens are read downstream (cinema/mae/mae.py:548-550, cinema/convvit.py:186-207).  Every op of the block except the
// depthwise conv is per-voxel and the depthwise conv only sees zeros at masked voxels, so the outputs (and gradients) at
// visible voxels depend on visible voxels alone: the stem can be evaluated on the 25 % visible voxels with identical
// results.  Activations are "token-major compact rows": row = token_row * Bv + pos[voxel in token], Bv = voxels per kept
// token at this stage, tokens in the raster order of TokenSelection.keep.  A neighbour voxel is found through
// rank[b * T + token] (row of that token or -1 = masked = contributes zero).
//
// forward / data gradient (flipped taps): per-row lists of visible neighbours, built once per mask (sparse_nbr_build_kernel), drive a
// gather kernel with no barriers in its loop.  weight gradient: one workgroup walks a chunk of kept tokens; per token it gathers the
// (B + k - 1)^3 halo of the token's voxel block into LDS (zeros for masked / out-of-volume voxels) and accumulates tap-major.
#include "common.cuh"
#include "../../include/cinema_hip.h"

namespace {

struct SpP {
  const bf16_t* x;      // compact rows [n_tok * Bv][c] (forward: input; data-gradient: dy; weight-gradient: input)
  const bf16_t* dy;     // weight-gradient only
  const float* w;       // [c][taps]
  const float* bias;    // [c] or nullptr
  bf16_t* y;            // compact rows out
  float* dw; float* dbias; float* ws;
  int b, tx, ty, tz, bx, by, bz, n_tok;
  const int* keep; const int* rank; const int* pos;
  int c, kx, ky, kz, flip, tok_per_block;
};

struct TokCoord { int bb, x0, y0, z0; };  // sample and the dense coordinates of the token's first voxel
__device__ __forceinline__ TokCoord token_coord(const SpP& p, int r) {
  const int T = p.tx * p.ty * p.tz;
  const int tid = p.keep[r];
  const int bb = tid / T, t = tid - bb * T;
  const int tzz = t % p.tz, tyy = (t / p.tz) % p.ty, txx = t / (p.tz * p.ty);
  return {bb, txx * p.bx, tyy * p.by, tzz * p.bz};
}

// Halo geometry of one token block: the halo box is cut into token CELLS (the token itself and its neighbours); a cell is either
// fully visible or fully masked, which is a workgroup-uniform fact.
struct Halo {
  int Hx, Hy, Hz, Hvox;      // halo extent in voxels
  int rx, ry, rz;            // kernel radii
  int cxr, cyr, czr;         // neighbour cells on each side
  int ncx, ncy, ncz, ncells;
};
__device__ __forceinline__ Halo make_halo(const SpP& p) {
  Halo h;
  h.Hx = p.bx + p.kx - 1; h.Hy = p.by + p.ky - 1; h.Hz = p.bz + p.kz - 1; h.Hvox = h.Hx * h.Hy * h.Hz;
  h.rx = p.kx >> 1; h.ry = p.ky >> 1; h.rz = p.kz >> 1;
  h.cxr = (h.rx + p.bx - 1) / p.bx; h.cyr = (h.ry + p.by - 1) / p.by; h.czr = (h.rz + p.bz - 1) / p.bz;
  h.ncx = 2 * h.cxr + 1; h.ncy = 2 * h.cyr + 1; h.ncz = 2 * h.czr + 1; h.ncells = h.ncx * h.ncy * h.ncz;
  return h;
}

// step 1a: row-of-token of every cell (-1: masked / outside the grid) -> LDS: the ONLY dependent global lookups of a token, all in parallel
__device__ __forceinline__ void load_cells(const SpP& p, const Halo& h, const TokCoord& tc, int* cell_rank, int tid) {
  if (tid < h.ncells) {
    const int T = p.tx * p.ty * p.tz;
    const int cz = tid % h.ncz, cy = (tid / h.ncz) % h.ncy, cx = tid / (h.ncz * h.ncy);
    const int ttx = tc.x0 / p.bx + cx - h.cxr, tty = tc.y0 / p.by + cy - h.cyr, ttz = tc.z0 / p.bz + cz - h.czr;
    int rk = -1;
    if (ttx >= 0 && ttx < p.tx && tty >= 0 && tty < p.ty && ttz >= 0 && ttz < p.tz) rk = p.rank[tc.bb * T + (ttx * p.ty + tty) * p.tz + ttz];
    cell_rank[tid] = rk;
  }
}
// step 1b: source row of every halo voxel (-1: masked / outside) from the cell table
__device__ __forceinline__ void index_halo(const SpP& p, const Halo& h, const int* pos_l, const int* cell_rank, int* src_row, int tid) {
  const int Bv = p.bx * p.by * p.bz;
  for (int hv = tid; hv < h.Hvox; hv += 256) {
    const int hz = hv % h.Hz, hy = (hv / h.Hz) % h.Hy, hx = hv / (h.Hz * h.Hy);
    // offsets relative to the first voxel of the left-most cell (non-negative)
    const int ox = hx - h.rx + h.cxr * p.bx, oy = hy - h.ry + h.cyr * p.by, oz = hz - h.rz + h.czr * p.bz;
    const int rk = cell_rank[((ox / p.bx) * h.ncy + oy / p.by) * h.ncz + oz / p.bz];
    src_row[hv] = rk >= 0 ? rk * Bv + pos_l[((ox % p.bx) * p.by + (oy % p.by)) * p.bz + oz % p.bz] : -1;
  }
}
// step 2: copy the rows into the LDS tile [Hvox][64 channels] bf16 (zeros where masked / outside / beyond c); loads in batches of 4 per
// thread BEFORE the first LDS store (one load per iteration serialised ten global latencies per token)
__device__ __forceinline__ void fill_halo(const SpP& p, const Halo& h, int c0, const int* src_row, char* tile, int tid) {
  const int n = h.Hvox * 8;
  for (int base = tid; base < n; base += 256 * 4) {
    uint4 val[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ch = base + q * 256;
      val[q] = make_uint4(0u, 0u, 0u, 0u);
      if (ch < n) {
        const int row = src_row[ch >> 3], cg = ch & 7;
        if (row >= 0 && c0 + cg * 8 < p.c) val[q] = *reinterpret_cast<const uint4*>(p.x + (size_t)row * p.c + c0 + cg * 8);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ch = base + q * 256;
      if (ch < n) *reinterpret_cast<uint4*>(tile + (size_t)(ch >> 3) * 128 + (ch & 7) * 16) = val[q];
    }
  }
}

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}

// ---- neighbour lists: for every compact row the (tap, source row) pairs of its VISIBLE stencil neighbours, in tap order, packed as
// (tap << 24 | row); built once per mask and stage (one wave per row: 64 taps per pass, ballot-compacted) and reused by the forward and
// data-gradient passes of every conv block of the stage.  nbr: [rows][NBR_STRIDE], cnt: [rows].
constexpr int NBR_STRIDE = 128;
__device__ __forceinline__ void sparse_nbr_build_body(SpP p, int* nbr, int* cnt) {
  const int lane = threadIdx.x & 63;
  const int Bv = p.bx * p.by * p.bz, T = p.tx * p.ty * p.tz, taps = p.kx * p.ky * p.kz;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.n_tok * Bv) return;
  const int r = row / Bv, q = row - r * Bv;
  // raster voxel index stored at row offset q: u with pos[u] == q
  int u = 0;
  for (int base = 0; base < Bv; base += 64) {
    const bool hit = base + lane < Bv && p.pos[base + lane] == q;
    const unsigned long long m = __ballot(hit);
    if (m) u = base + __ffsll((long long)m) - 1;
  }
  const TokCoord tc = token_coord(p, r);
  const int uz = u % p.bz, uy = (u / p.bz) % p.by, ux = u / (p.bz * p.by);
  const int rx = p.kx >> 1, ry = p.ky >> 1, rz = p.kz >> 1;
  int total = 0;
  for (int t0 = 0; t0 < taps; t0 += 64) {
    const int t = t0 + lane;
    int idx = -1;
    if (t < taps) {
      const int k = t % p.kz, j = (t / p.kz) % p.ky, i = t / (p.kz * p.ky);
      const int X = tc.x0 + ux + i - rx, Y = tc.y0 + uy + j - ry, Z = tc.z0 + uz + k - rz;
      if (X >= 0 && X < p.tx * p.bx && Y >= 0 && Y < p.ty * p.by && Z >= 0 && Z < p.tz * p.bz) {
        const int rk = p.rank[tc.bb * T + ((X / p.bx) * p.ty + (Y / p.by)) * p.tz + Z / p.bz];
        if (rk >= 0) idx = rk * Bv + p.pos[((X % p.bx) * p.by + (Y % p.by)) * p.bz + Z % p.bz];
      }
    }
    const unsigned long long m = __ballot(idx >= 0);
    if (idx >= 0) nbr[(size_t)row * NBR_STRIDE + total + __popcll(m & ((1ull << lane) - 1ull))] = (t << 24) | idx;
    total += __popcll(m);
  }
  if (lane == 0) cnt[row] = total;
}
struct SpNbrP { SpP p; int* nbr; int* cnt; };
__global__ __launch_bounds__(256) void sparse_nbr_build_kernel(SpNbrP q) { sparse_nbr_build_body(q.p, q.nbr, q.cnt); }
__global__ __launch_bounds__(256) void sparse_nbr_build_lanes_kernel(Lanes<SpNbrP> L) { const SpNbrP& q = L.p[blockIdx.y]; sparse_nbr_build_body(q.p, q.nbr, q.cnt); }
// y[row] = bias + sum over the row's neighbour list of w[tap] * x[source row]: thread = (row, channel group of 8), a workgroup walks
// groups of 32 rows with the 64-channel weight slab in LDS as [tap][channel] fp32; loads of 4 list entries are issued together.
__device__ __forceinline__ void sparse_dwconv_list_body(SpP p, const int* nbr, const int* cnt, int n_rows) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  float* wl = reinterpret_cast<float*>(dyn_smem);
  const int taps = p.kx * p.ky * p.kz;
  const int tid = threadIdx.x, c0 = blockIdx.y * 64;
  for (int i = tid; i < taps * 64; i += 256) {
    const int t = i >> 6, cc = i & 63;
    const int ts = p.flip ? taps - 1 - t : t;
    wl[i] = (c0 + cc < p.c) ? p.w[(size_t)(c0 + cc) * taps + ts] : 0.f;
  }
  __syncthreads();
  const int cg = tid & 7, ch = c0 + cg * 8;
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.bias && ch < p.c) {
    const float4 a = *reinterpret_cast<const float4*>(p.bias + ch), b4 = *reinterpret_cast<const float4*>(p.bias + ch + 4);
    bv[0] = a.x; bv[1] = a.y; bv[2] = a.z; bv[3] = a.w; bv[4] = b4.x; bv[5] = b4.y; bv[6] = b4.z; bv[7] = b4.w;
  }
  const bf16_t* xc = p.x + ch;
  for (int row = blockIdx.x * 32 + (tid >> 3); row < n_rows; row += gridDim.x * 32) {
    if (ch >= p.c) continue;
    const int n = cnt[row];
    const int* lst = nbr + (size_t)row * NBR_STRIDE;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = bv[i];
    for (int e0 = 0; e0 < n; e0 += 4) {
      int ent[4];
      uint4 xv[4];
#pragma unroll
      for (int q = 0; q < 4; q++) ent[q] = e0 + q < n ? lst[e0 + q] : -1;
#pragma unroll
      for (int q = 0; q < 4; q++) xv[q] = ent[q] >= 0 ? *reinterpret_cast<const uint4*>(xc + (size_t)(ent[q] & 0xffffff) * p.c) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (ent[q] < 0) continue;
        const float* wt = wl + (ent[q] >> 24) * 64 + cg * 8;
        const float4 w0 = *reinterpret_cast<const float4*>(wt), w1 = *reinterpret_cast<const float4*>(wt + 4);
        float f[8];
        unpack8(xv[q], f);
        acc[0] = fmaf(w0.x, f[0], acc[0]); acc[1] = fmaf(w0.y, f[1], acc[1]); acc[2] = fmaf(w0.z, f[2], acc[2]); acc[3] = fmaf(w0.w, f[3], acc[3]);
        acc[4] = fmaf(w1.x, f[4], acc[4]); acc[5] = fmaf(w1.y, f[5], acc[5]); acc[6] = fmaf(w1.z, f[6], acc[6]); acc[7] = fmaf(w1.w, f[7], acc[7]);
      }
    }
    uint4 ov;
    ov.x = pack_bf2(acc[0], acc[1]); ov.y = pack_bf2(acc[2], acc[3]); ov.z = pack_bf2(acc[4], acc[5]); ov.w = pack_bf2(acc[6], acc[7]);
    *reinterpret_cast<uint4*>(p.y + (size_t)row * p.c + ch) = ov;
  }
}
struct SpListP { SpP p; const int* nbr; const int* cnt; int n_rows; };
__global__ __launch_bounds__(256) void sparse_dwconv_list_kernel(SpListP q) { sparse_dwconv_list_body(q.p, q.nbr, q.cnt, q.n_rows); }
__global__ __launch_bounds__(256) void sparse_dwconv_list_lanes_kernel(Lanes<SpListP> L) { const SpListP& q = L.p[blockIdx.z]; sparse_dwconv_list_body(q.p, q.nbr, q.cnt, q.n_rows); }
// Weight gradient: thread = (in-plane tap, channel group of 8) holds dw for its kz taps; a workgroup walks a chunk of kept
// tokens (halo of x and the token's dy block in LDS) and writes ONE partial slab [c][taps] + [c] at the end
// (reduced by the kernel below, deterministic).  LDS: tile [Hvox][64] bf16 | dyb [Bv][64] bf16 | pos [Bv] int
template <int KZ>
__device__ __forceinline__ void sparse_dwconv_wgrad_body(const SpP& p) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  const int Bv = p.bx * p.by * p.bz, nxy = p.kx * p.ky, taps = nxy * KZ;
  const Halo h = make_halo(p);
  const int Hy = h.Hy, Hz = h.Hz;
  char* tile = dyn_smem;
  char* dyb = tile + (size_t)h.Hvox * 128;
  int* src_row = reinterpret_cast<int*>(dyb + (size_t)Bv * 128);
  int* cell_vis = src_row + h.Hvox;
  int* pos_l = cell_vis + h.ncells;
  const int tid = threadIdx.x, c0 = blockIdx.y * 64;
  for (int i = tid; i < Bv; i += 256) pos_l[i] = p.pos[i];
  const int cg = tid & 7, txy = tid >> 3;
  const bool worker = txy < nxy && c0 + cg * 8 < p.c;
  const int ti = txy / p.ky, tj = txy % p.ky;
  const bool center = worker && ti == (p.kx >> 1) && tj == (p.ky >> 1);
  float acc[KZ][8];
  float accb[8];
#pragma unroll
  for (int k = 0; k < KZ; k++)
#pragma unroll
    for (int i = 0; i < 8; i++) acc[k][i] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) accb[i] = 0.f;
  const int r_begin = blockIdx.x * p.tok_per_block, r_end = min(p.n_tok, r_begin + p.tok_per_block);
  for (int r = r_begin; r < r_end; r++) {
    __syncthreads();
    const TokCoord tc = token_coord(p, r);
    load_cells(p, h, tc, cell_vis, tid);
    __syncthreads();
    index_halo(p, h, pos_l, cell_vis, src_row, tid);
    __syncthreads();
    fill_halo(p, h, c0, src_row, tile, tid);
    for (int ch = tid; ch < Bv * 8; ch += 256) {  // dy block in RASTER voxel order
      const int v = ch >> 3, g8 = ch & 7;
      uint4 val = make_uint4(0u, 0u, 0u, 0u);
      if (c0 + g8 * 8 < p.c) val = *reinterpret_cast<const uint4*>(p.dy + ((size_t)r * Bv + pos_l[v]) * p.c + c0 + g8 * 8);
      *reinterpret_cast<uint4*>(dyb + (size_t)v * 128 + g8 * 16) = val;
    }
    __syncthreads();
    if (worker) {
      for (int v = 0; v < Bv; v++) {
        const int uz = v % p.bz, uy = (v / p.bz) % p.by, ux = v / (p.bz * p.by);
        float d[8];
        unpack8(*reinterpret_cast<const uint4*>(dyb + (size_t)v * 128 + cg * 16), d);
        if (center) {
#pragma unroll
          for (int i = 0; i < 8; i++) accb[i] += d[i];
        }
        const int hbase = ((ux + ti) * Hy + (uy + tj)) * Hz + uz;
#pragma unroll
        for (int k = 0; k < KZ; k++) {
          const uint4 xv = *reinterpret_cast<const uint4*>(tile + (size_t)(hbase + k) * 128 + cg * 16);
          if ((xv.x | xv.y | xv.z | xv.w) == 0u) continue;
          float f[8];
          unpack8(xv, f);
#pragma unroll
          for (int i = 0; i < 8; i++) acc[k][i] = fmaf(d[i], f[i], acc[k][i]);
        }
      }
    }
  }
  if (!worker) return;
  const int ch = c0 + cg * 8;
  float* slab = p.ws + (size_t)blockIdx.x * p.c * (taps + 1);
#pragma unroll
  for (int k = 0; k < KZ; k++)
#pragma unroll
    for (int i = 0; i < 8; i++) slab[(size_t)(ch + i) * taps + txy * KZ + k] = acc[k][i];
  if (center) {
#pragma unroll
    for (int i = 0; i < 8; i++) slab[(size_t)p.c * taps + ch + i] = accb[i];
  }
}
template <int KZ>
__global__ __launch_bounds__(256) void sparse_dwconv_wgrad_kernel(SpP p) { sparse_dwconv_wgrad_body<KZ>(p); }
template <int KZ>
__global__ __launch_bounds__(256) void sparse_dwconv_wgrad_lanes_kernel(Lanes<SpP> L) { sparse_dwconv_wgrad_body<KZ>(L.p[blockIdx.z]); }
// ---- halo source rows of every kept token, once per mask / stage / kernel extent: hidx[r][hv] = compact row of halo voxel hv of token r, -1 = masked /
// outside the volume.  The weight-gradient kernel below reads them as one coalesced row per token instead of chasing keep -> rank -> pos per token.
__device__ __forceinline__ void sparse_halo_index_body(const SpP& p, int* hidx) {
  __shared__ int cell_rank[128];
  __shared__ int pos_l[512];
  const Halo h = make_halo(p);
  const int Bv = p.bx * p.by * p.bz, tid = threadIdx.x;
  for (int i = tid; i < Bv; i += 256) pos_l[i] = p.pos[i];
  const int r = blockIdx.x;
  const TokCoord tc = token_coord(p, r);
  load_cells(p, h, tc, cell_rank, tid);
  __syncthreads();
  index_halo(p, h, pos_l, cell_rank, hidx + (size_t)r * h.Hvox, tid);
}
struct SpHaloP { SpP p; int* hidx; };
__global__ __launch_bounds__(256) void sparse_halo_index_kernel(SpHaloP q) { sparse_halo_index_body(q.p, q.hidx); }
__global__ __launch_bounds__(256) void sparse_halo_index_lanes_kernel(Lanes<SpHaloP> L) { const SpHaloP& q = L.p[blockIdx.y]; sparse_halo_index_body(q.p, q.hidx); }

// Weight gradient, pipelined over the tokens of a workgroup (the first form above spends ~6 us per token in three dependent global round trips - keep ->
// rank -> rows - and four barriers, for ~1.3 us of FMAs): the halo source rows come from the table above, and while token r is accumulated from LDS the
// rows of token r + 1 are already in flight into registers (<= NG 16-byte loads per thread) and the index row of token r + 2 into one more register pair;
// two barriers per token.  Same arithmetic, same slab layout, same reduce kernel.
template <int KZ, int NG>
__device__ __forceinline__ void sparse_dwconv_wgrad_pipe_body(const SpP& p, const int* hidx) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  const int Bv = p.bx * p.by * p.bz, nxy = p.kx * p.ky, taps = nxy * KZ;
  const Halo h = make_halo(p);
  const int Hy = h.Hy, Hz = h.Hz, Hvox = h.Hvox;
  char* tile = dyn_smem;
  char* dyb = tile + (size_t)Hvox * 128;
  int* src_row = reinterpret_cast<int*>(dyb + (size_t)Bv * 128);
  int* pos_l = src_row + Hvox;
  const int tid = threadIdx.x, c0 = blockIdx.y * 64;
  for (int i = tid; i < Bv; i += 256) pos_l[i] = p.pos[i];
  const int cg = tid & 7, txy = tid >> 3;
  const bool worker = txy < nxy && c0 + cg * 8 < p.c;
  const int ti = txy / p.ky, tj = txy % p.ky;
  const bool center = worker && ti == (p.kx >> 1) && tj == (p.ky >> 1);
  float acc[KZ][8];
  float accb[8];
#pragma unroll
  for (int k = 0; k < KZ; k++)
#pragma unroll
    for (int i = 0; i < 8; i++) acc[k][i] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) accb[i] = 0.f;
  const int r_begin = blockIdx.x * p.tok_per_block, r_end = min(p.n_tok, r_begin + p.tok_per_block);
  const int n_chunks = Hvox * 8, n_dy = Bv * 8;
  const bool ch_ok = c0 + cg * 8 < p.c;  // (chunk & 7) == cg for every chunk of this thread: 256 is a multiple of 8

  int idx_reg[2];                        // index row of a later token: halo voxels tid and tid + 256 (Hvox <= 512)
  uint4 hv[NG], dv[2];                   // rows in flight: halo chunks tid + 256 q, dy chunks tid + 256 q
  auto load_idx = [&](int r) {
    const int* row = hidx + (size_t)r * Hvox;
    idx_reg[0] = tid < Hvox ? row[tid] : -1;
    idx_reg[1] = tid + 256 < Hvox ? row[tid + 256] : -1;
  };
  auto store_idx = [&]() {
    if (tid < Hvox) src_row[tid] = idx_reg[0];
    if (tid + 256 < Hvox) src_row[tid + 256] = idx_reg[1];
  };
  auto gather = [&](int r) {             // needs src_row (LDS) of token r
#pragma unroll
    for (int q = 0; q < NG; q++) {
      const int chk = tid + q * 256;
      hv[q] = make_uint4(0u, 0u, 0u, 0u);
      if (chk < n_chunks && ch_ok) {
        const int row = src_row[chk >> 3];
        if (row >= 0) hv[q] = *reinterpret_cast<const uint4*>(p.x + (size_t)row * p.c + c0 + cg * 8);
      }
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int chk = tid + q * 256;
      dv[q] = make_uint4(0u, 0u, 0u, 0u);
      if (chk < n_dy && ch_ok) dv[q] = *reinterpret_cast<const uint4*>(p.dy + ((size_t)r * Bv + pos_l[chk >> 3]) * p.c + c0 + cg * 8);
    }
  };
  auto store_rows = [&]() {
#pragma unroll
    for (int q = 0; q < NG; q++) {
      const int chk = tid + q * 256;
      if (chk < n_chunks) *reinterpret_cast<uint4*>(tile + (size_t)(chk >> 3) * 128 + cg * 16) = hv[q];
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int chk = tid + q * 256;
      if (chk < n_dy) *reinterpret_cast<uint4*>(dyb + (size_t)(chk >> 3) * 128 + cg * 16) = dv[q];
    }
  };

  if (r_begin >= r_end) return;
  load_idx(r_begin);
  store_idx();
  __syncthreads();                       // pos_l and the first index row
  gather(r_begin);
  if (r_begin + 1 < r_end) load_idx(r_begin + 1);
  store_rows();
  __syncthreads();                       // everyone has read src_row(r_begin)
  store_idx();
  __syncthreads();
  for (int r = r_begin; r < r_end; r++) {
    const bool more = r + 1 < r_end;
    if (more) gather(r + 1);             // in flight while token r is accumulated
    if (r + 2 < r_end) load_idx(r + 2);
    if (worker) {
      for (int v = 0; v < Bv; v++) {
        const int uz = v % p.bz, uy = (v / p.bz) % p.by, ux = v / (p.bz * p.by);
        const uint4 dq = *reinterpret_cast<const uint4*>(dyb + (size_t)v * 128 + cg * 16);
        if (center) {
          float d[8];
          unpack8(dq, d);
#pragma unroll
          for (int i = 0; i < 8; i++) accb[i] += d[i];
        }
        // acc[c] += x[c] * dy[c] per channel as v_dot2c_f32_bf16 on the packed pairs with the OTHER half of the dy pair zeroed: the bf16 products are exact in
        // fp32 and the zero term adds nothing, so this is the same fp32 FMA chain - without the 16 shift / mask instructions per 8 FMAs of the unpacked form
        const uint32_t dw_[4] = {dq.x, dq.y, dq.z, dq.w};
        uint32_t dlo[4], dhi[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { dlo[j] = dw_[j] & 0xffffu; dhi[j] = dw_[j] & 0xffff0000u; }
        const int hbase = ((ux + ti) * Hy + (uy + tj)) * Hz + uz;
#pragma unroll
        for (int k = 0; k < KZ; k++) {
          const uint4 xv = *reinterpret_cast<const uint4*>(tile + (size_t)(hbase + k) * 128 + cg * 16);
          const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
          for (int j = 0; j < 4; j++) {
            asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[k][2 * j]) : "v"(xw[j]), "v"(dlo[j]));
            asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[k][2 * j + 1]) : "v"(xw[j]), "v"(dhi[j]));
          }
        }
      }
    }
    __syncthreads();                     // tile / dyb / src_row of token r (and r + 1's index row) are no longer read
    if (more) { store_rows(); store_idx(); }
    __syncthreads();
  }
  if (!worker) return;
  const int ch = c0 + cg * 8;
  float* slab = p.ws + (size_t)blockIdx.x * p.c * (taps + 1);
#pragma unroll
  for (int k = 0; k < KZ; k++)
#pragma unroll
    for (int i = 0; i < 8; i++) slab[(size_t)(ch + i) * taps + txy * KZ + k] = acc[k][i];
  if (center) {
#pragma unroll
    for (int i = 0; i < 8; i++) slab[(size_t)p.c * taps + ch + i] = accb[i];
  }
}
struct SpPipeP { SpP p; const int* hidx; };
template <int KZ, int NG>
__global__ __launch_bounds__(256) void sparse_dwconv_wgrad_pipe_kernel(SpPipeP q) { sparse_dwconv_wgrad_pipe_body<KZ, NG>(q.p, q.hidx); }
template <int KZ, int NG>
__global__ __launch_bounds__(256) void sparse_dwconv_wgrad_pipe_lanes_kernel(Lanes<SpPipeP> L) { const SpPipeP& q = L.p[blockIdx.z]; sparse_dwconv_wgrad_pipe_body<KZ, NG>(q.p, q.hidx); }

// dw[i] += sum_blocks slab[block][i]  (i < c*taps), dbias[j] += sum_blocks slab[block][c*taps + j]
__device__ __forceinline__ void sparse_wgrad_reduce_body(const float* ws, int nblocks, int n_w, int n_b, float* dw, float* dbias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = n_w + n_b;
  if (i >= total) return;
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
  float s = 0.f;
  for (int b = b0; b < b1; b++) s += ws[(size_t)b * total + i];
  if (i < n_w) unsafeAtomicAdd(dw + i, s);
  else if (dbias) unsafeAtomicAdd(dbias + (i - n_w), s);
}
struct SpRedP { const float* ws; int nblocks; int n_w; int n_b; float* dw; float* dbias; };
__global__ __launch_bounds__(256) void sparse_wgrad_reduce_kernel(SpRedP q) { sparse_wgrad_reduce_body(q.ws, q.nblocks, q.n_w, q.n_b, q.dw, q.dbias); }
__global__ __launch_bounds__(256) void sparse_wgrad_reduce_lanes_kernel(Lanes<SpRedP> L) { const SpRedP& q = L.p[blockIdx.z]; sparse_wgrad_reduce_body(q.ws, q.nblocks, q.n_w, q.n_b, q.dw, q.dbias); }
int fill(SpP& p, const cinema_sparse_geom* g, int c, int kx, int ky, int kz) {
  if (!g || !g->keep || !g->rank || !g->pos || g->b <= 0 || g->tx <= 0 || g->ty <= 0 || g->tz <= 0 || g->bx <= 0 || g->by <= 0 || g->bz <= 0 || g->n_tok <= 0)
    return CINEMA_ERR_BAD_ARG;
  if (c <= 0 || (c & 7) || !(kx & 1) || !(ky & 1) || !(kz & 1)) return CINEMA_ERR_UNSUPPORTED;
  p.b = g->b; p.tx = g->tx; p.ty = g->ty; p.tz = g->tz; p.bx = g->bx; p.by = g->by; p.bz = g->bz; p.n_tok = g->n_tok;
  p.keep = g->keep; p.rank = g->rank; p.pos = g->pos; p.c = c; p.kx = kx; p.ky = ky; p.kz = kz;
  return 0;
}

}  // namespace

CINEMA_API long long cinema_sparse_dwconv_wgrad_workspace_bytes(int n_tok, int c, int kx, int ky, int kz) {
  const int blocks = n_tok < 1024 ? n_tok : 1024;
  return (long long)blocks * c * (kx * ky * kz + 1) * 4;
}

CINEMA_API long long cinema_sparse_nbr_ints(int n_rows) { return (long long)n_rows * (NBR_STRIDE + 1); }

CINEMA_API int cinema_sparse_nbr_build(const cinema_sparse_geom* geom, int kx, int ky, int kz, int* nbr, int* cnt, void* stream) {
  if (!nbr || !cnt) return CINEMA_ERR_BAD_ARG;
  SpP p{};
  if (int e = fill(p, geom, 8, kx, ky, kz)) return e;
  const long long rows = (long long)p.n_tok * p.bx * p.by * p.bz;
  if (kx * ky * kz > NBR_STRIDE || rows >= (1 << 24)) return CINEMA_ERR_UNSUPPORTED;
  launch_lanes(sparse_nbr_build_kernel, sparse_nbr_build_lanes_kernel, 1, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, SpNbrP{p, nbr, cnt});
  return launch_status();
}

CINEMA_API int cinema_sparse_dwconv_fwd(const uint16_t* x, const float* w, const float* bias, uint16_t* y, const cinema_sparse_geom* geom, const int* nbr,
                                        const int* cnt, int c, int kx, int ky, int kz, int flip, void* stream) {
  if (!x || !w || !y || !nbr || !cnt) return CINEMA_ERR_BAD_ARG;
  SpP p{};
  if (int e = fill(p, geom, c, kx, ky, kz)) return e;
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.flip = flip;
  const int taps = kx * ky * kz;
  if (taps > NBR_STRIDE) return CINEMA_ERR_UNSUPPORTED;
  const int n_rows = p.n_tok * p.bx * p.by * p.bz;
  int blocks = (n_rows + 31) / 32;
  if (blocks > 1024) blocks = 1024;
  launch_lanes(sparse_dwconv_list_kernel, sparse_dwconv_list_lanes_kernel, 2, dim3(blocks, (c + 63) / 64), dim3(256), (size_t)taps * 64 * 4, (hipStream_t)stream, SpListP{p, nbr, cnt, n_rows});
  return launch_status();
}

CINEMA_API long long cinema_sparse_halo_ints(const cinema_sparse_geom* geom, int kx, int ky, int kz) {
  if (!geom) return 0;
  return (long long)geom->n_tok * (geom->bx + kx - 1) * (geom->by + ky - 1) * (geom->bz + kz - 1);
}

CINEMA_API int cinema_sparse_halo_index(const cinema_sparse_geom* geom, int kx, int ky, int kz, int* halo_idx, void* stream) {
  if (!halo_idx) return CINEMA_ERR_BAD_ARG;
  SpP p{};
  if (int e = fill(p, geom, 8, kx, ky, kz)) return e;
  const int ncells = (2 * ((kx / 2 + p.bx - 1) / p.bx) + 1) * (2 * ((ky / 2 + p.by - 1) / p.by) + 1) * (2 * ((kz / 2 + p.bz - 1) / p.bz) + 1);
  if (ncells > 128 || p.bx * p.by * p.bz > 512) return CINEMA_ERR_UNSUPPORTED;
  launch_lanes(sparse_halo_index_kernel, sparse_halo_index_lanes_kernel, 1, dim3((unsigned)p.n_tok), dim3(256), 0, (hipStream_t)stream, SpHaloP{p, halo_idx});
  return launch_status();
}

CINEMA_API int cinema_sparse_dwconv_bwd_weight(const uint16_t* x, const uint16_t* dy, float* dw, float* dbias, float* workspace, long long workspace_bytes,
                                               const cinema_sparse_geom* geom, int c, int kx, int ky, int kz, const int* halo_idx, void* stream) {
  if (!x || !dy || !dw || !workspace) return CINEMA_ERR_BAD_ARG;
  SpP p{};
  if (int e = fill(p, geom, c, kx, ky, kz)) return e;
  if (kz != 5 || kx * ky > 32) return CINEMA_ERR_UNSUPPORTED;
  p.x = x; p.dy = dy; p.dw = dw; p.dbias = dbias; p.ws = workspace;
  const int taps = kx * ky * kz, Bv = p.bx * p.by * p.bz;
  const size_t hv = (size_t)(p.bx + kx - 1) * (p.by + ky - 1) * (p.bz + kz - 1);
  const int ncells = (2 * ((kx / 2 + p.bx - 1) / p.bx) + 1) * (2 * ((ky / 2 + p.by - 1) / p.by) + 1) * (2 * ((kz / 2 + p.bz - 1) / p.bz) + 1);
  hipStream_t st = (hipStream_t)stream;
  const bool pipe = halo_idx != nullptr && hv <= 512 && Bv <= 64;
  const size_t smem = pipe ? hv * 128 + (size_t)Bv * 128 + (hv + Bv) * 4 : hv * 128 + (size_t)Bv * 128 + (hv + ncells + Bv) * 4;
  if (smem > 160 * 1024) return CINEMA_ERR_UNSUPPORTED;
  int blocks = p.n_tok < 1024 ? p.n_tok : 1024;
  if (pipe) {  // one round of the resident workgroups (LDS and ~140 VGPRs: at most 3 per CU), a whole number of tokens each
    static int cus = 0;
    if (cus == 0) {
      int dev = 0; hipDeviceProp_t prop;
      cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    int per_cu = (int)((160 * 1024) / smem);
    per_cu = per_cu > 3 ? 3 : (per_cu < 1 ? 1 : per_cu);
    const int slots = per_cu * cus;
    if (blocks > slots) blocks = slots;
  }
  p.tok_per_block = (p.n_tok + blocks - 1) / blocks;
  blocks = (p.n_tok + p.tok_per_block - 1) / p.tok_per_block;
  if (workspace_bytes < (long long)blocks * c * (taps + 1) * 4) return CINEMA_ERR_BAD_ARG;
  if (pipe) {
    const SpPipeP q{p, halo_idx};
    const dim3 grid(blocks, (c + 63) / 64);
#define PIPE_LAUNCH(NG)                                                                                                                      \
  do {                                                                                                                                       \
    static bool attr = false;                                                                                                                \
    if (!attr) { (void)hipFuncSetAttribute((const void*)sparse_dwconv_wgrad_pipe_kernel<5, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                 (void)hipFuncSetAttribute((const void*)sparse_dwconv_wgrad_pipe_lanes_kernel<5, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
    launch_lanes(sparse_dwconv_wgrad_pipe_kernel<5, NG>, sparse_dwconv_wgrad_pipe_lanes_kernel<5, NG>, 2, grid, dim3(256), smem, st, q);     \
  } while (0)
    if (hv <= 64) PIPE_LAUNCH(2);
    else if (hv <= 192) PIPE_LAUNCH(6);
    else if (hv <= 320) PIPE_LAUNCH(10);
    else PIPE_LAUNCH(16);
#undef PIPE_LAUNCH
  } else {
    static bool attr_set[16] = {};
    (void)dyn_lds_attr_once(attr_set, (const void*)sparse_dwconv_wgrad_kernel<5>, 160 * 1024);
    launch_lanes(sparse_dwconv_wgrad_kernel<5>, sparse_dwconv_wgrad_lanes_kernel<5>, 2, dim3(blocks, (c + 63) / 64), dim3(256), smem, st, p);
  }
  const int total = c * (taps + 1);
  launch_lanes(sparse_wgrad_reduce_kernel, sparse_wgrad_reduce_lanes_kernel, 2, dim3((total + 255) / 256, 16), dim3(256), 0, st, SpRedP{(const float*)workspace, blocks, c * taps, c, dw, dbias});
  return launch_status();
}
