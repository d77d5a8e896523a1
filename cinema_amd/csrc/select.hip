// Random-mask bookkeeping of one view as two launches (reference: cinema/mae/mae.py:30-65 get_batch_random_patch_mask, :550 boolean-mask
// indexing; the host code did this with ~50 torch launches per view - two argsorts for the rank, one for the raster-ordered selection,
// a dozen int64 index kernels - which were ~1 ms of tiny kernels per step).
//   mask_select_kernel   : noise [b][n] -> mask (True = removed, the n - n_keep LARGEST ranks, ties by index like a stable sort), and/or
//                          mask -> kept / dropped token lists in raster order (flat ids b*n + i and positions i)
//   visible_index_kernel : kept tokens -> compact rank table and the stage-1 voxel ids of the visible rows
#include "common.cuh"
#include "../../include/cinema_hip.h"

namespace {

// rank_i = #{j : noise_j < noise_i or (noise_j == noise_i and j < i)}  (argsort(argsort(noise)) with a stable sort); mask_i = rank_i >= n_keep.
// grid = (chunks of 64 elements, batch): a workgroup ranks 64 elements, each of its four waves against a quarter of every LDS piece of 2048 floats (one sample
// per workgroup was 131 us for a 2304-token row, one element per thread over the whole row 61 us: 144 workgroups of 2304 serial compares per thread)
__global__ __launch_bounds__(256) void mask_rank_kernel(const float* noise, uint8_t* mask, int n, int n_keep) {
  __shared__ __attribute__((aligned(16))) float piece[2048];
  __shared__ int part[4][64];
  const int b = blockIdx.y, tid = threadIdx.x, e = tid & 63, q = tid >> 6, i = blockIdx.x * 64 + e;
  const float* row = noise + (size_t)b * n;
  const float v = i < n ? row[i] : 0.f;
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 2048) {
    __syncthreads();
    for (int t = tid; t < 2048; t += 256) piece[t] = j0 + t < n ? row[j0 + t] : __builtin_inff();  // +inf never counts
    __syncthreads();
    const int lim = min(2048, n - j0);
    const int lo = q * 512, hi = min(lim, lo + 512);  // this wave's quarter of the piece
    for (int t = lo; t < hi; t += 4) {  // same LDS address across the wave: broadcast reads
      const float4 w = *reinterpret_cast<const float4*>(piece + t);
      const int j = j0 + t;
      rank += (w.x < v || (w.x == v && j < i)) ? 1 : 0;
      rank += (w.y < v || (w.y == v && j + 1 < i)) ? 1 : 0;
      rank += (w.z < v || (w.z == v && j + 2 < i)) ? 1 : 0;
      rank += (w.w < v || (w.w == v && j + 3 < i)) ? 1 : 0;
    }
  }
  part[q][e] = rank;
  __syncthreads();
  if (q == 0 && i < n) mask[(size_t)b * n + i] = (part[0][e] + part[1][e] + part[2][e] + part[3][e]) >= n_keep ? 1 : 0;
}

__global__ __launch_bounds__(256) void mask_select_kernel(const uint8_t* mask, int n, int* keep_pos, int* drop_pos, int* keep, int* drop) {
  __shared__ int partial[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint8_t* mrow = mask + (size_t)b * n;
  // raster-ordered compaction: thread t owns the contiguous chunk [t*per, (t+1)*per)
  const int per = (n + 255) / 256;
  const int lo = min(n, tid * per), hi = min(n, lo + per);
  int kept = 0;
  for (int i = lo; i < hi; i++) kept += mrow[i] ? 0 : 1;
  partial[tid] = kept;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {  // inclusive Hillis-Steele scan of the 256 chunk counts
    const int v = tid >= off ? partial[tid - off] : 0;
    __syncthreads();
    partial[tid] += v;
    __syncthreads();
  }
  int k = partial[tid] - kept;  // kept tokens before this chunk
  int d = lo - k;               // dropped tokens before this chunk
  const int total_keep = partial[255];
  const int base = b * n;
  for (int i = lo; i < hi; i++) {
    if (mrow[i]) {
      const size_t o = (size_t)b * (n - total_keep) + d++;
      if (drop_pos) drop_pos[o] = i;
      if (drop) drop[o] = base + i;
    } else {
      const size_t o = (size_t)b * total_keep + k++;
      if (keep_pos) keep_pos[o] = i;
      if (keep) keep[o] = base + i;
    }
  }
}

struct VisP {
  const int* keep; const int* inv1;
  int* rank; int* idx1;
  int n_rows, n_tok_all, block_vol;
  int g[3], bl[3];  // token grid and stage-1 voxels per token, padded to 3 dims with leading ones
};

// one thread per (kept token r, row offset q): idx1[r * block_vol + q] = flat id, in the (batch, *grid*block) stage-1 volume, of the voxel
// stored at row offset q of token r (inv1[q] = its raster index inside the token's block); q == 0 also writes rank[keep[r]] = r
__device__ __forceinline__ void visible_index_body(const VisP& p) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)p.n_rows * p.block_vol) return;
  const int r = (int)(idx / p.block_vol), q = (int)(idx - (long long)r * p.block_vol);
  const int tokid = p.keep[r];
  if (q == 0) p.rank[tokid] = r;
  int t = tokid % p.n_tok_all;
  const int bb = tokid / p.n_tok_all;
  const int tz = t % p.g[2]; t /= p.g[2];
  const int ty = t % p.g[1]; t /= p.g[1];
  const int tx = t;
  int u = p.inv1[q];
  const int uz = u % p.bl[2]; u /= p.bl[2];
  const int uy = u % p.bl[1]; u /= p.bl[1];
  const int ux = u;
  long long v = bb;
  v = v * (p.g[0] * p.bl[0]) + (tx * p.bl[0] + ux);
  v = v * (p.g[1] * p.bl[1]) + (ty * p.bl[1] + uy);
  v = v * (p.g[2] * p.bl[2]) + (tz * p.bl[2] + uz);
  p.idx1[idx] = (int)v;
}
__global__ __launch_bounds__(256) void visible_index_kernel(VisP p) { visible_index_body(p); }
__global__ __launch_bounds__(256) void visible_index_lanes_kernel(Lanes<VisP> L) { visible_index_body(L.p[blockIdx.y]); }
}  // namespace

CINEMA_API int cinema_mask_select(const float* noise, uint8_t* mask, int batch, int n, int n_keep, int* keep_pos, int* drop_pos, int* keep, int* drop,
                                  void* stream) {
  if (!mask || batch <= 0 || n <= 0 || n_keep < 0 || n_keep > n) return CINEMA_ERR_BAD_ARG;
  if (noise) CINEMA_LAUNCH(mask_rank_kernel, dim3((n + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream, noise, mask, n, n_keep);
  if (keep_pos || drop_pos || keep || drop)
    CINEMA_LAUNCH(mask_select_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)mask, n, keep_pos, drop_pos, keep, drop);
  return launch_status();
}

CINEMA_API int cinema_visible_index(const int* keep, int n_rows, int n_dims, const int* grid_host, const int* block_host, const int* inv1, int* rank, int* idx1,
                                    void* stream) {
  if (!keep || !inv1 || !rank || !idx1 || n_rows <= 0 || n_dims < 1 || n_dims > 3 || !grid_host || !block_host) return CINEMA_ERR_BAD_ARG;
  VisP p;
  p.keep = keep; p.inv1 = inv1; p.rank = rank; p.idx1 = idx1; p.n_rows = n_rows;
  for (int d = 0; d < 3; d++) { p.g[d] = 1; p.bl[d] = 1; }
  for (int d = 0; d < n_dims; d++) { p.g[3 - n_dims + d] = grid_host[d]; p.bl[3 - n_dims + d] = block_host[d]; }
  p.n_tok_all = p.g[0] * p.g[1] * p.g[2];
  p.block_vol = p.bl[0] * p.bl[1] * p.bl[2];
  const long long total = (long long)n_rows * p.block_vol;
  launch_lanes(visible_index_kernel, visible_index_lanes_kernel, 1, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}
