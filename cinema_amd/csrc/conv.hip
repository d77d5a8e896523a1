// Channels-last depthwise convolution (MaskedConvBlock.dw_conv) and non-overlapping patch gather/scatter
// (patchify / k==s convolutions) for gfx950.  HBM/L2-bound integer-indexed data movement + VALU FMAs: 16-byte
// accesses along the channel axis, weights staged once per block in LDS, sliding z-window register reuse in the
// weight-gradient kernel.
#include "common.cuh"
#include "../../include/cinema_hip.h"

namespace {

struct DwP {
  const bf16_t* x; const bf16_t* dy; const float* w; const float* bias; bf16_t* y;
  float* dw; float* dbias;
  const uint8_t* out_mask;  // optional per-voxel 0/1 multiplier on the output (masked data gradient)
  float* ws;                // weight-gradient partial slabs [gridDim.x][c * (taps + 1)] or nullptr (atomic fallback)
  int b, X, Y, Z, c, kx, ky, kz;
  int flip;  // 1: correlate with the flipped kernel (data gradient)
};

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = bf2f((bf16_t)(u.x & 0xffff)); f[1] = bf2f((bf16_t)(u.x >> 16));
  f[2] = bf2f((bf16_t)(u.y & 0xffff)); f[3] = bf2f((bf16_t)(u.y >> 16));
  f[4] = bf2f((bf16_t)(u.z & 0xffff)); f[5] = bf2f((bf16_t)(u.z >> 16));
  f[6] = bf2f((bf16_t)(u.w & 0xffff)); f[7] = bf2f((bf16_t)(u.w >> 16));
}

// y[v, c] = bias[c] + sum_t w[c][t] * x[v + t - r, c]; block = 64 voxels x 4 groups of 8 channels (one 32-channel slab)
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(DwP p) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  float* wl = reinterpret_cast<float*>(dyn_smem);  // [taps][32]
  const int taps = p.kx * p.ky * p.kz;
  const int c0 = blockIdx.y * 32;
  for (int i = threadIdx.x; i < taps * 32; i += 256) {
    const int t = i >> 5, cc = i & 31;
    const int ts = p.flip ? taps - 1 - t : t;
    wl[i] = (c0 + cc < p.c) ? p.w[(size_t)(c0 + cc) * taps + ts] : 0.f;
  }
  __syncthreads();
  const int cg = threadIdx.x & 3;
  const int ch = c0 + cg * 8;
  const long long nvox = (long long)p.b * p.X * p.Y * p.Z;
  const long long vox = (long long)blockIdx.x * 64 + (threadIdx.x >> 2);
  if (vox >= nvox || ch >= p.c) return;
  const int z = (int)(vox % p.Z), y = (int)((vox / p.Z) % p.Y), x = (int)((vox / ((long long)p.Z * p.Y)) % p.X);
  const long long bb = vox / ((long long)p.Z * p.Y * p.X);
  const int rx = p.kx >> 1, ry = p.ky >> 1, rz = p.kz >> 1;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = p.bias ? p.bias[ch + i] : 0.f;
  for (int i = 0; i < p.kx; i++) {
    const int xx = x + i - rx;
    if (xx < 0 || xx >= p.X) continue;
    for (int j = 0; j < p.ky; j++) {
      const int yy = y + j - ry;
      if (yy < 0 || yy >= p.Y) continue;
      for (int k = 0; k < p.kz; k++) {
        const int zz = z + k - rz;
        if (zz < 0 || zz >= p.Z) continue;
        const size_t src = ((((size_t)bb * p.X + xx) * p.Y + yy) * p.Z + zz) * p.c + ch;
        const uint4 u = *reinterpret_cast<const uint4*>(p.x + src);
        float f[8];
        unpack8(u, f);
        const float* wt = wl + ((i * p.ky + j) * p.kz + k) * 32 + cg * 8;
        const float4 w0 = *reinterpret_cast<const float4*>(wt), w1 = *reinterpret_cast<const float4*>(wt + 4);
        acc[0] = fmaf(w0.x, f[0], acc[0]); acc[1] = fmaf(w0.y, f[1], acc[1]); acc[2] = fmaf(w0.z, f[2], acc[2]); acc[3] = fmaf(w0.w, f[3], acc[3]);
        acc[4] = fmaf(w1.x, f[4], acc[4]); acc[5] = fmaf(w1.y, f[5], acc[5]); acc[6] = fmaf(w1.z, f[6], acc[6]); acc[7] = fmaf(w1.w, f[7], acc[7]);
      }
    }
  }
  if (p.out_mask && !p.out_mask[vox]) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = 0.f;
  }
  uint4 o;
  o.x = pack_bf2(acc[0], acc[1]); o.y = pack_bf2(acc[2], acc[3]); o.z = pack_bf2(acc[4], acc[5]); o.w = pack_bf2(acc[6], acc[7]);
  *reinterpret_cast<uint4*>(p.y + (size_t)vox * p.c + ch) = o;
}

// Register-blocked forward / data-gradient for kz == 5: a thread owns a column of ZB = 16 consecutive z outputs of one
// (b, x, y) position and 8 channels (128 fp32 accumulators).  For each of the kx*ky in-plane taps it streams the
// (ZB + 4)-long input column once (16-byte loads) and every loaded vector feeds the 5 z-taps: 40 FMAs per load + unpack
// instead of 8 in the per-voxel kernel (which was VALU/L1-bound at ~400 VALU instructions per output element).
// 2-D maps are passed as (X=1, Y=H, Z=W): the walked axis is then W.
template <int ZB, bool FULLZ>  // FULLZ: Z == ZB, every z-range test is resolved at compile time (the SAX volumes, Z = 16)
__global__ __launch_bounds__(256, 2) void dwconv_zcol_kernel(DwP p) {  // 2 waves/SIMD: the unconstrained build took 270 registers = 1 wave/SIMD
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  float* wl = reinterpret_cast<float*>(dyn_smem);  // [kx*ky][5][64] fp32 (64-channel slab)
  const int nxy = p.kx * p.ky, taps = nxy * 5;
  const int c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < taps * 64; i += 256) {
    const int t = i >> 6, cc = i & 63;
    const int ts = p.flip ? taps - 1 - t : t;
    wl[i] = (c0 + cc < p.c) ? p.w[(size_t)(c0 + cc) * taps + ts] : 0.f;
  }
  __syncthreads();
  const int cg = threadIdx.x & 7;            // 8 channel groups of 8 = 64 channels: one 128-byte line per voxel
  const int ch = c0 + cg * 8;
  const long long npos = (long long)p.b * p.X * p.Y;
  const long long pos = (long long)xcd_remap(blockIdx.x, gridDim.x) * 32 + (threadIdx.x >> 3);  // stencil halos meet in one L2
  if (pos >= npos || ch >= p.c) return;
  const int y = (int)(pos % p.Y), x = (int)((pos / p.Y) % p.X);
  const long long bb = pos / ((long long)p.Y * p.X);
  const int z0 = FULLZ ? 0 : blockIdx.z * ZB;  // wave-uniform z block
  const int rx = p.kx >> 1, ry = p.ky >> 1;
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.bias) {  // two vector loads once (a per-accumulator `bias ? load : 0` compiles to 128 serialised branch+load+wait)
    const float4 a = *reinterpret_cast<const float4*>(p.bias + ch), b4 = *reinterpret_cast<const float4*>(p.bias + ch + 4);
    bv[0] = a.x; bv[1] = a.y; bv[2] = a.z; bv[3] = a.w; bv[4] = b4.x; bv[5] = b4.y; bv[6] = b4.z; bv[7] = b4.w;
  }
  float acc[ZB][8];
#pragma unroll
  for (int z = 0; z < ZB; z++)
#pragma unroll
    for (int i = 0; i < 8; i++) acc[z][i] = bv[i];
  for (int i = 0; i < p.kx; i++) {
    const int xx = x + i - rx;
    if (xx < 0 || xx >= p.X) continue;
    for (int j = 0; j < p.ky; j++) {
      const int yy = y + j - ry;
      if (yy < 0 || yy >= p.Y) continue;
      float w5[5][8];
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const float* wt = wl + ((i * p.ky + j) * 5 + k) * 64 + cg * 8;
        const float4 a = *reinterpret_cast<const float4*>(wt), b4 = *reinterpret_cast<const float4*>(wt + 4);
        w5[k][0] = a.x; w5[k][1] = a.y; w5[k][2] = a.z; w5[k][3] = a.w; w5[k][4] = b4.x; w5[k][5] = b4.y; w5[k][6] = b4.z; w5[k][7] = b4.w;
      }
      const uint4* colp = reinterpret_cast<const uint4*>(__builtin_assume_aligned(p.x + ((((size_t)bb * p.X + xx) * p.Y + yy) * p.Z) * p.c + ch, 16));
      const int cstride = p.c >> 3;  // uint4 per voxel
#pragma unroll
      for (int zi = 0; zi < ZB + 4; zi++) {  // input z = z0 + zi - 2
        const int zin = z0 + zi - 2;
        if (FULLZ) { if (zi < 2 || zi >= ZB + 2) continue; }
        else if (zin < 0 || zin >= p.Z) continue;
        float f[8];
        { const uint4 u = colp[(size_t)zin * cstride]; unpack8(u, f); }
#pragma unroll
        for (int k = 0; k < 5; k++) {
          const int zo = zi - k;  // output z (relative) fed through z-tap k:  zin = zout + k - 2
          if (zo < 0 || zo >= ZB) continue;
#pragma unroll
          for (int c = 0; c < 8; c++) acc[zo][c] = fmaf(w5[k][c], f[c], acc[zo][c]);
        }
      }
    }
  }
  const size_t vox0 = (((size_t)bb * p.X + x) * p.Y + y) * p.Z + z0;
  uint32_t keep = 0xffffffffu;  // bit z: output voxel z0+z is visible
  if (p.out_mask) {
    keep = 0u;
#pragma unroll
    for (int z = 0; z < ZB; z++)
      if (FULLZ || z0 + z < p.Z) keep |= (p.out_mask[vox0 + z] ? 1u : 0u) << z;
  }
#pragma unroll
  for (int z = 0; z < ZB; z++) {
    if (!FULLZ && z0 + z >= p.Z) continue;
    uint4 o;
    o.x = pack_bf2(acc[z][0], acc[z][1]); o.y = pack_bf2(acc[z][2], acc[z][3]); o.z = pack_bf2(acc[z][4], acc[z][5]); o.w = pack_bf2(acc[z][6], acc[z][7]);
    if (!((keep >> z) & 1u)) o = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(p.y + (vox0 + z) * p.c + ch) = o;
  }
}

// Weight gradient with a sliding window along z (kz == KZ): thread = (tap_xy, channel group of 8); it walks whole z
// columns of its block's (b, x, y) set, re-using the KZ-wide x window in registers: 2 loads per KZ*8 FMAs.
template <int KZ, int ZC>  // ZC > 0: Z == ZC known at compile time (address arithmetic and range tests fold)
__global__ void dwconv_wgrad_walk_kernel(DwP p, int cols_per_block) {
  const int Z = ZC > 0 ? ZC : p.Z;
  const int ncg = min(8, (p.c - blockIdx.y * 64) / 8);
  const int nxy = p.kx * p.ky;
  const int role = threadIdx.x;
  if (role >= nxy * ncg) return;
  const int cg = role % ncg, txy = role / ncg;
  const int ti = txy / p.ky, tj = txy % p.ky;
  const int ch = blockIdx.y * 64 + cg * 8;
  const int rx = p.kx >> 1, ry = p.ky >> 1;
  constexpr int RZ = KZ / 2;
  float acc[KZ][8];
  float accb[8];
#pragma unroll
  for (int k = 0; k < KZ; k++)
#pragma unroll
    for (int i = 0; i < 8; i++) acc[k][i] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) accb[i] = 0.f;
  const bool center = (ti == rx && tj == ry);
  const long long ncols = (long long)p.b * p.X * p.Y;
  const long long col_begin = (long long)xcd_remap(blockIdx.x, gridDim.x) * cols_per_block;
  const long long col_end = min(ncols, col_begin + cols_per_block);
  for (long long col = col_begin; col < col_end; col++) {
    const int y = (int)(col % p.Y), x = (int)((col / p.Y) % p.X);
    const long long bb = col / ((long long)p.Y * p.X);
    const int xx = x + ti - rx, yy = y + tj - ry;
    if (xx < 0 || xx >= p.X || yy < 0 || yy >= p.Y) continue;
    const bf16_t* xcol = p.x + ((((size_t)bb * p.X + xx) * p.Y + yy) * Z) * p.c + ch;
    const bf16_t* dcol = p.dy + ((size_t)col * Z) * p.c + ch;
    // window win[k] = x[z + k - RZ]
    float win[KZ][8];
#pragma unroll
    for (int k = 0; k < KZ; k++) {
      const int zz = k - RZ;
      if (zz >= 0 && zz < Z) { const uint4 u = *reinterpret_cast<const uint4*>(xcol + (size_t)zz * p.c); unpack8(u, win[k]); }
      else {
#pragma unroll
        for (int i = 0; i < 8; i++) win[k][i] = 0.f;
      }
    }
    for (int z = 0; z < Z; z++) {  // kept rolled: unrolling by 5 or fully (to rename the window slide away) spilled and measured 1.6x slower
      float d[8];
      { const uint4 u = *reinterpret_cast<const uint4*>(dcol + (size_t)z * p.c); unpack8(u, d); }
#pragma unroll
      for (int k = 0; k < KZ; k++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[k][i] = fmaf(d[i], win[k][i], acc[k][i]);
      if (center) {
#pragma unroll
        for (int i = 0; i < 8; i++) accb[i] += d[i];
      }
      // slide
#pragma unroll
      for (int k = 0; k + 1 < KZ; k++)
#pragma unroll
        for (int i = 0; i < 8; i++) win[k][i] = win[k + 1][i];
      const int zn = z + 1 + RZ;
      if (zn < Z) { const uint4 u = *reinterpret_cast<const uint4*>(xcol + (size_t)zn * p.c); unpack8(u, win[KZ - 1]); }
      else {
#pragma unroll
        for (int i = 0; i < 8; i++) win[KZ - 1][i] = 0.f;
      }
    }
  }
  const int taps = nxy * KZ;
  if (p.ws) {  // deterministic two-pass: this block's partial sums to its slab, dwconv_wgrad_reduce_kernel adds the slabs up
    float* slab = p.ws + (size_t)blockIdx.x * p.c * (taps + 1);
#pragma unroll
    for (int k = 0; k < KZ; k++)
#pragma unroll
      for (int i = 0; i < 8; i++) slab[(size_t)(ch + i) * taps + txy * KZ + k] = acc[k][i];
    if (center) {
#pragma unroll
      for (int i = 0; i < 8; i++) slab[(size_t)p.c * taps + ch + i] = accb[i];
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < KZ; k++)
#pragma unroll
    for (int i = 0; i < 8; i++) unsafeAtomicAdd(p.dw + (size_t)(ch + i) * taps + txy * KZ + k, acc[k][i]);
  if (center && p.dbias) {
#pragma unroll
    for (int i = 0; i < 8; i++) unsafeAtomicAdd(p.dbias + ch + i, accb[i]);
  }
}

// dw[i] += sum_blocks slab[block][i]  (i < c*taps), dbias[j] += sum_blocks slab[block][c*taps + j]
__global__ __launch_bounds__(256) void dwconv_wgrad_reduce_kernel(const float* ws, int nblocks, int n_w, int n_b, float* dw, float* dbias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = n_w + n_b;
  if (i >= total) return;
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
  float s = 0.f;
  for (int b = b0; b < b1; b++) s += ws[(size_t)b * total + i];
  if (i < n_w) unsafeAtomicAdd(dw + i, s);   // 16 slices per element
  else if (dbias) unsafeAtomicAdd(dbias + (i - n_w), s);
}

// naive weight gradient for other kernel extents: thread = (tap, channel); each block reduces a voxel chunk
__global__ __launch_bounds__(256) void dwconv_wgrad_naive_kernel(DwP p, long long vox_per_block) {
  const int taps = p.kx * p.ky * p.kz;
  const long long nvox = (long long)p.b * p.X * p.Y * p.Z;
  const long long v0 = (long long)blockIdx.x * vox_per_block, v1 = min(nvox, v0 + vox_per_block);
  const int rx = p.kx >> 1, ry = p.ky >> 1, rz = p.kz >> 1;
  for (int role = threadIdx.x; role < taps * p.c; role += blockDim.x) {
    const int ch = role % p.c, t = role / p.c;
    const int k = t % p.kz, j = (t / p.kz) % p.ky, i = t / (p.kz * p.ky);
    float acc = 0.f, accb = 0.f;
    for (long long vox = v0; vox < v1; vox++) {
      const int z = (int)(vox % p.Z), y = (int)((vox / p.Z) % p.Y), x = (int)((vox / ((long long)p.Z * p.Y)) % p.X);
      const long long bb = vox / ((long long)p.Z * p.Y * p.X);
      const float d = bf2f(p.dy[(size_t)vox * p.c + ch]);
      accb += d;
      const int xx = x + i - rx, yy = y + j - ry, zz = z + k - rz;
      if (xx < 0 || xx >= p.X || yy < 0 || yy >= p.Y || zz < 0 || zz >= p.Z) continue;
      acc = fmaf(d, bf2f(p.x[((((size_t)bb * p.X + xx) * p.Y + yy) * p.Z + zz) * p.c + ch]), acc);
    }
    unsafeAtomicAdd(p.dw + (size_t)ch * taps + t, acc);
    if (p.dbias && t == (rx * p.ky + ry) * p.kz + rz) unsafeAtomicAdd(p.dbias + ch, accb);
  }
}

// ------------------------------------------------------------------------------------------------
struct PatchP {
  int b, c, gx, gy, gz, px, py, pz;
  long long sb, sc, sx, sy, sz;
  int n_rows;
  const int* token_idx;
};

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

__device__ __forceinline__ long long patch_src_offset(const PatchP& g, int row, int f, int& cc) {
  const int tok = g.token_idx ? g.token_idx[row] : row;
  const int G = g.gx * g.gy * g.gz;
  const int bb = tok / G, gi = tok % G;
  const int iz = gi % g.gz, iy = (gi / g.gz) % g.gy, ix = gi / (g.gz * g.gy);
  cc = f % g.c;
  const int pf = f / g.c;
  const int kz = pf % g.pz, ky = (pf / g.pz) % g.py, kx = pf / (g.pz * g.py);
  return (long long)bb * g.sb + (long long)(ix * g.px + kx) * g.sx + (long long)(iy * g.py + ky) * g.sy +
         (long long)(iz * g.pz + kz) * g.sz + (long long)cc * g.sc;
}

// VEC=4: c % 4 == 0 and sc == 1 (channels-last source), 4 consecutive features share one patch voxel
template <typename TS, typename TO, int VEC>
__device__ __forceinline__ void patch_gather_body(const TS* src, TO* out, int ld_out, PatchP g) {
  const int F = g.px * g.py * g.pz * g.c;
  const long long total = (long long)g.n_rows * (F / VEC);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / (F / VEC)), f = (int)(i % (F / VEC)) * VEC;
    int cc;
    const long long off = patch_src_offset(g, row, f, cc);
#pragma unroll
    for (int v = 0; v < VEC; v++) stf<TO>(out + (size_t)row * ld_out + f + v, ldf<TS>(src + off + v));
  }
}
template <typename TS, typename TO, int VEC>
struct PGatherP { const TS* src; TO* out; int ld_out; PatchP g; };
template <typename TS, typename TO, int VEC>
__global__ void patch_gather_kernel(PGatherP<TS, TO, VEC> q) { patch_gather_body<TS, TO, VEC>(q.src, q.out, q.ld_out, q.g); }
template <typename TS, typename TO, int VEC>
__global__ void patch_gather_lanes_kernel(Lanes<PGatherP<TS, TO, VEC>> L) { const PGatherP<TS, TO, VEC>& q = L.p[blockIdx.y]; patch_gather_body<TS, TO, VEC>(q.src, q.out, q.ld_out, q.g); }
template <typename TR, typename TD, int VEC>
__device__ __forceinline__ void patch_scatter_body(const TR* rows, int ld_rows, TD* dst, int accumulate, PatchP g) {
  const int F = g.px * g.py * g.pz * g.c;
  const long long total = (long long)g.n_rows * (F / VEC);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / (F / VEC)), f = (int)(i % (F / VEC)) * VEC;
    int cc;
    const long long off = patch_src_offset(g, row, f, cc);
#pragma unroll
    for (int v = 0; v < VEC; v++) {
      float val = ldf<TR>(rows + (size_t)row * ld_rows + f + v);
      if (accumulate) val += ldf<TD>(dst + off + v);
      stf<TD>(dst + off + v, val);
    }
  }
}
template <typename TR, typename TD, int VEC>
struct PScatterP { const TR* rows; int ld_rows; TD* dst; int accumulate; PatchP g; };
template <typename TR, typename TD, int VEC>
__global__ void patch_scatter_kernel(PScatterP<TR, TD, VEC> q) { patch_scatter_body<TR, TD, VEC>(q.rows, q.ld_rows, q.dst, q.accumulate, q.g); }
template <typename TR, typename TD, int VEC>
__global__ void patch_scatter_lanes_kernel(Lanes<PScatterP<TR, TD, VEC>> L) { const PScatterP<TR, TD, VEC>& q = L.p[blockIdx.y]; patch_scatter_body<TR, TD, VEC>(q.rows, q.ld_rows, q.dst, q.accumulate, q.g); }
PatchP to_dev(const cinema_patch_geom* g) {
  PatchP p;
  p.b = g->b; p.c = g->c; p.gx = g->gx; p.gy = g->gy; p.gz = g->gz; p.px = g->px; p.py = g->py; p.pz = g->pz;
  p.sb = g->sb; p.sc = g->sc; p.sx = g->sx; p.sy = g->sy; p.sz = g->sz; p.n_rows = g->n_rows; p.token_idx = g->token_idx;
  return p;
}
bool geom_ok(const cinema_patch_geom* g) {
  return g && g->b > 0 && g->c > 0 && g->gx > 0 && g->gy > 0 && g->gz > 0 && g->px > 0 && g->py > 0 && g->pz > 0 && g->n_rows > 0;
}
int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace

static int dw_check(int b, int X, int Y, int Z, int c, int kx, int ky, int kz) {
  if (b <= 0 || X <= 0 || Y <= 0 || Z <= 0 || c <= 0 || kx <= 0 || ky <= 0 || kz <= 0) return CINEMA_ERR_BAD_ARG;
  if ((c & 7) || !(kx & 1) || !(ky & 1) || !(kz & 1)) return CINEMA_ERR_UNSUPPORTED;
  return 0;
}

static int dw_launch_fwd(const DwP& p, hipStream_t st) {
  const long long nvox = (long long)p.b * p.X * p.Y * p.Z;
  dim3 grid((unsigned)((nvox + 63) / 64), (p.c + 31) / 32);
  if (p.kz == 5 && (size_t)p.kx * p.ky * 5 * 64 * sizeof(float) <= 64 * 1024) {
    const int nzb = (p.Z + 15) / 16;
    const long long npos = (long long)p.b * p.X * p.Y;
    dim3 zgrid((unsigned)((npos + 31) / 32), (p.c + 63) / 64, nzb);
    const size_t wsmem = (size_t)p.kx * p.ky * 5 * 64 * sizeof(float);
    if (p.Z == 16) CINEMA_LAUNCH((dwconv_zcol_kernel<16, true>), zgrid, dim3(256), wsmem, st, p);
    else CINEMA_LAUNCH((dwconv_zcol_kernel<16, false>), zgrid, dim3(256), wsmem, st, p);
    return launch_status();
  }
  const size_t smem = (size_t)p.kx * p.ky * p.kz * 32 * sizeof(float);
  CINEMA_LAUNCH(dwconv_fwd_kernel, grid, dim3(256), smem, st, p);
  return launch_status();
}

CINEMA_API int cinema_dwconv_fwd(const uint16_t* x, const float* w, const float* bias, uint16_t* y, int b, int X, int Y, int Z, int c, int kx,
                                 int ky, int kz, void* stream) {
  if (!x || !w || !y) return CINEMA_ERR_BAD_ARG;
  if (int e = dw_check(b, X, Y, Z, c, kx, ky, kz)) return e;
  DwP p{}; p.x = x; p.w = w; p.bias = bias; p.y = y; p.b = b; p.X = X; p.Y = Y; p.Z = Z; p.c = c; p.kx = kx; p.ky = ky; p.kz = kz; p.flip = 0;
  return dw_launch_fwd(p, (hipStream_t)stream);
}

CINEMA_API int cinema_dwconv_bwd_data(const uint16_t* dy, const float* w, uint16_t* dx, const uint8_t* out_mask, int b, int X, int Y, int Z, int c,
                                      int kx, int ky, int kz, void* stream) {
  if (!dy || !w || !dx) return CINEMA_ERR_BAD_ARG;
  if (int e = dw_check(b, X, Y, Z, c, kx, ky, kz)) return e;
  DwP p{}; p.x = dy; p.w = w; p.bias = nullptr; p.y = dx; p.b = b; p.X = X; p.Y = Y; p.Z = Z; p.c = c; p.kx = kx; p.ky = ky; p.kz = kz; p.flip = 1; p.out_mask = out_mask;
  return dw_launch_fwd(p, (hipStream_t)stream);
}

CINEMA_API int cinema_dwconv_bwd_weight(const uint16_t* x, const uint16_t* dy, float* dw, float* dbias, float* workspace, long long workspace_bytes,
                                        int b, int X, int Y, int Z, int c, int kx, int ky, int kz, void* stream) {
  if (!x || !dy || !dw) return CINEMA_ERR_BAD_ARG;
  if (int e = dw_check(b, X, Y, Z, c, kx, ky, kz)) return e;
  DwP p{}; p.x = x; p.dy = dy; p.dw = dw; p.dbias = dbias; p.b = b; p.X = X; p.Y = Y; p.Z = Z; p.c = c; p.kx = kx; p.ky = ky; p.kz = kz;
  hipStream_t st = (hipStream_t)stream;
  if (kz == 5 && kx * ky * 8 <= 1024) {
    const long long ncols = (long long)b * X * Y;
    int cpb = (int)((ncols + 1023) / 1024);
    if (cpb < 4) cpb = 4;
    dim3 grid((unsigned)((ncols + cpb - 1) / cpb), (c + 63) / 64);
    int threads = kx * ky * 8;
    threads = ((threads + 63) / 64) * 64;
    const int taps = kx * ky * kz;
    const long long need = (long long)grid.x * c * (taps + 1) * 4;
    p.ws = (workspace && workspace_bytes >= need && !(c & 63)) ? workspace : nullptr;  // every (block, channel, tap) slot is written when c % 64 == 0
    if (Z == 16) CINEMA_LAUNCH((dwconv_wgrad_walk_kernel<5, 16>), grid, dim3(threads), 0, st, p, cpb);
    else CINEMA_LAUNCH((dwconv_wgrad_walk_kernel<5, 0>), grid, dim3(threads), 0, st, p, cpb);
    if (p.ws) {
      const int total = c * (taps + 1);
      CINEMA_LAUNCH(dwconv_wgrad_reduce_kernel, dim3((total + 255) / 256, 16), dim3(256), 0, st, (const float*)p.ws, (int)grid.x, c * taps, c, dw, dbias);
    }
    return launch_status();
  }
  const long long nvox = (long long)b * X * Y * Z;
  long long vpb = (nvox + 511) / 512;
  if (vpb < 64) vpb = 64;
  CINEMA_LAUNCH(dwconv_wgrad_naive_kernel, dim3((unsigned)((nvox + vpb - 1) / vpb)), dim3(256), 0, st, p, vpb);
  return launch_status();
}

CINEMA_API int cinema_patch_gather(const void* src, int src_dtype, void* out, int out_dtype, int ld_out, const cinema_patch_geom* geom,
                                   void* stream) {
  if (!src || !out || !geom_ok(geom)) return CINEMA_ERR_BAD_ARG;
  const PatchP g = to_dev(geom);
  const long long F = (long long)g.px * g.py * g.pz * g.c;
  const bool vec = (g.c % 4 == 0) && g.sc == 1;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for((long long)g.n_rows * F / (vec ? 4 : 1), 256);
#define CINEMA_PG(TS, TO)                                                                                                  \
  do {                                                                                                                     \
    if (vec) launch_lanes(patch_gather_kernel<TS, TO, 4>, patch_gather_lanes_kernel<TS, TO, 4>, 1, dim3(grid), dim3(256), 0, st, PGatherP<TS, TO, 4>{(const TS*)src, (TO*)out, ld_out, g}); \
    else launch_lanes(patch_gather_kernel<TS, TO, 1>, patch_gather_lanes_kernel<TS, TO, 1>, 1, dim3(grid), dim3(256), 0, st, PGatherP<TS, TO, 1>{(const TS*)src, (TO*)out, ld_out, g});     \
  } while (0)
  if (src_dtype == 1 && out_dtype == 0) CINEMA_PG(float, bf16_t);
  else if (src_dtype == 1 && out_dtype == 1) CINEMA_PG(float, float);
  else if (src_dtype == 0 && out_dtype == 0) CINEMA_PG(bf16_t, bf16_t);
  else if (src_dtype == 0 && out_dtype == 1) CINEMA_PG(bf16_t, float);
  else return CINEMA_ERR_BAD_ARG;
#undef CINEMA_PG
  return launch_status();
}

CINEMA_API int cinema_patch_scatter(const void* rows, int rows_dtype, int ld_rows, void* dst, int dst_dtype, int accumulate,
                                    const cinema_patch_geom* geom, void* stream) {
  if (!rows || !dst || !geom_ok(geom)) return CINEMA_ERR_BAD_ARG;
  const PatchP g = to_dev(geom);
  const long long F = (long long)g.px * g.py * g.pz * g.c;
  const bool vec = (g.c % 4 == 0) && g.sc == 1;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for((long long)g.n_rows * F / (vec ? 4 : 1), 256);
#define CINEMA_PS(TR, TD)                                                                                                                 \
  do {                                                                                                                                    \
    if (vec) launch_lanes(patch_scatter_kernel<TR, TD, 4>, patch_scatter_lanes_kernel<TR, TD, 4>, 1, dim3(grid), dim3(256), 0, st, PScatterP<TR, TD, 4>{(const TR*)rows, ld_rows, (TD*)dst, accumulate, g}); \
    else launch_lanes(patch_scatter_kernel<TR, TD, 1>, patch_scatter_lanes_kernel<TR, TD, 1>, 1, dim3(grid), dim3(256), 0, st, PScatterP<TR, TD, 1>{(const TR*)rows, ld_rows, (TD*)dst, accumulate, g});     \
  } while (0)
  if (rows_dtype == 0 && dst_dtype == 0) CINEMA_PS(bf16_t, bf16_t);
  else if (rows_dtype == 0 && dst_dtype == 1) CINEMA_PS(bf16_t, float);
  else if (rows_dtype == 1 && dst_dtype == 1) CINEMA_PS(float, float);
  else if (rows_dtype == 1 && dst_dtype == 0) CINEMA_PS(float, bf16_t);
  else return CINEMA_ERR_BAD_ARG;
#undef CINEMA_PS
  return launch_status();
}

// ------------------------------------------------------------------------------------------------
// Dense k^n "same" convolution of the segmentation decoder (reference ConvResBlock, cinema/conv.py:276-348) as
// im2col + MFMA GEMM.  cols[v][(tap, c)] = x[v + tap - r][c] (zero outside the volume), rows padded to ld (multiple of 8)
// with zeros so that the GEMM's 16-byte alignment rules hold for any channel count (the raw-image conv has C = 1).
// The data gradient is the mirrored GATHER: dx[v][c] = sum_tap dcols[v - (tap - r)][(tap, c)] - no atomics.
// ------------------------------------------------------------------------------------------------
namespace {

struct ColP {
  const bf16_t* x; bf16_t* cols; const bf16_t* dcols; bf16_t* dx;
  int b, X, Y, Z, c, kx, ky, kz, ld;
};

__global__ __launch_bounds__(256) void im2col_kernel(ColP p) {
  const int taps = p.kx * p.ky * p.kz, F = taps * p.c;
  const long long nvox = (long long)p.b * p.X * p.Y * p.Z;
  const int rx = p.kx >> 1, ry = p.ky >> 1, rz = p.kz >> 1;
  const bool vec = (p.c & 7) == 0;
  const int per_row = vec ? p.ld / 8 : p.ld;  // work items per output row
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvox * per_row; i += (long long)gridDim.x * 256) {
    const long long v = i / per_row;
    const int j = (int)(i - v * per_row);
    const int z = (int)(v % p.Z), y = (int)((v / p.Z) % p.Y), x = (int)((v / ((long long)p.Z * p.Y)) % p.X);
    const long long bb = v / ((long long)p.Z * p.Y * p.X);
    if (vec) {
      const int f = j * 8;
      uint4 val = make_uint4(0u, 0u, 0u, 0u);
      if (f < F) {
        const int t = f / p.c, cc = f - t * p.c;
        const int k = t % p.kz, jj = (t / p.kz) % p.ky, ii = t / (p.kz * p.ky);
        const int xx = x + ii - rx, yy = y + jj - ry, zz = z + k - rz;
        if (xx >= 0 && xx < p.X && yy >= 0 && yy < p.Y && zz >= 0 && zz < p.Z)
          val = *reinterpret_cast<const uint4*>(p.x + ((((size_t)bb * p.X + xx) * p.Y + yy) * p.Z + zz) * p.c + cc);
      }
      *reinterpret_cast<uint4*>(p.cols + (size_t)v * p.ld + f) = val;
    } else {
      bf16_t val = 0;
      if (j < F) {
        const int t = j / p.c, cc = j - t * p.c;
        const int k = t % p.kz, jj = (t / p.kz) % p.ky, ii = t / (p.kz * p.ky);
        const int xx = x + ii - rx, yy = y + jj - ry, zz = z + k - rz;
        if (xx >= 0 && xx < p.X && yy >= 0 && yy < p.Y && zz >= 0 && zz < p.Z) val = p.x[((((size_t)bb * p.X + xx) * p.Y + yy) * p.Z + zz) * p.c + cc];
      }
      p.cols[(size_t)v * p.ld + j] = val;
    }
  }
}

__global__ __launch_bounds__(256) void col2im_kernel(ColP p) {
  const int taps = p.kx * p.ky * p.kz;
  const long long nvox = (long long)p.b * p.X * p.Y * p.Z;
  const int rx = p.kx >> 1, ry = p.ky >> 1, rz = p.kz >> 1;
  const bool vec = (p.c & 7) == 0;
  const int per_vox = vec ? p.c / 8 : p.c;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvox * per_vox; i += (long long)gridDim.x * 256) {
    const long long v = i / per_vox;
    const int cg = (int)(i - v * per_vox);
    const int z = (int)(v % p.Z), y = (int)((v / p.Z) % p.Y), x = (int)((v / ((long long)p.Z * p.Y)) % p.X);
    const long long bb = v / ((long long)p.Z * p.Y * p.X);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < taps; t++) {
      const int k = t % p.kz, jj = (t / p.kz) % p.ky, ii = t / (p.kz * p.ky);
      const int xx = x - (ii - rx), yy = y - (jj - ry), zz = z - (k - rz);  // the output voxel whose tap t reads this input voxel
      if (xx < 0 || xx >= p.X || yy < 0 || yy >= p.Y || zz < 0 || zz >= p.Z) continue;
      const size_t row = (((size_t)bb * p.X + xx) * p.Y + yy) * p.Z + zz;
      if (vec) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(p.dcols + row * p.ld + (size_t)t * p.c + cg * 8), f);
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] += f[e];
      } else {
        acc[0] += bf2f(p.dcols[row * p.ld + (size_t)t * p.c + cg]);
      }
    }
    if (vec) {
      uint4 o;
      o.x = pack_bf2(acc[0], acc[1]); o.y = pack_bf2(acc[2], acc[3]); o.z = pack_bf2(acc[4], acc[5]); o.w = pack_bf2(acc[6], acc[7]);
      *reinterpret_cast<uint4*>(p.dx + (size_t)v * p.c + cg * 8) = o;
    } else {
      p.dx[(size_t)v * p.c + cg] = f2bf(acc[0]);
    }
  }
}

}  // namespace

CINEMA_API int cinema_im2col(const uint16_t* x, uint16_t* cols, int ld_cols, int b, int X, int Y, int Z, int c, int kx, int ky, int kz, void* stream) {
  if (!x || !cols || b <= 0 || X <= 0 || Y <= 0 || Z <= 0 || c <= 0 || kx <= 0 || ky <= 0 || kz <= 0) return CINEMA_ERR_BAD_ARG;
  if (!(kx & 1) || !(ky & 1) || !(kz & 1) || (ld_cols & 7) || ld_cols < kx * ky * kz * c || (((uintptr_t)cols) & 15) || (!(c & 7) && (((uintptr_t)x) & 15)))
    return CINEMA_ERR_UNSUPPORTED;
  ColP p{}; p.x = x; p.cols = cols; p.b = b; p.X = X; p.Y = Y; p.Z = Z; p.c = c; p.kx = kx; p.ky = ky; p.kz = kz; p.ld = ld_cols;
  const long long items = (long long)b * X * Y * Z * ((c & 7) ? ld_cols : ld_cols / 8);
  CINEMA_LAUNCH(im2col_kernel, dim3(grid_for(items, 256)), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

CINEMA_API int cinema_col2im(const uint16_t* dcols, int ld_cols, uint16_t* dx, int b, int X, int Y, int Z, int c, int kx, int ky, int kz, void* stream) {
  if (!dcols || !dx || b <= 0 || X <= 0 || Y <= 0 || Z <= 0 || c <= 0 || kx <= 0 || ky <= 0 || kz <= 0) return CINEMA_ERR_BAD_ARG;
  if (!(kx & 1) || !(ky & 1) || !(kz & 1) || (ld_cols & 7) || ld_cols < kx * ky * kz * c || (!(c & 7) && ((((uintptr_t)dcols) & 15) || (((uintptr_t)dx) & 15))))
    return CINEMA_ERR_UNSUPPORTED;
  ColP p{}; p.dcols = dcols; p.dx = dx; p.b = b; p.X = X; p.Y = Y; p.Z = Z; p.c = c; p.kx = kx; p.ky = ky; p.kz = kz; p.ld = ld_cols;
  const long long items = (long long)b * X * Y * Z * ((c & 7) ? c : c / 8);
  CINEMA_LAUNCH(col2im_kernel, dim3(grid_for(items, 256)), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}
