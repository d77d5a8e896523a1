// Row gather/scatter/add, dtype casts, transposes and elementwise GELU for gfx950 (HBM-bound data movement).
#include "common.cuh"
#include "../../include/cinema_hip.h"

namespace {

__device__ __forceinline__ float4 ld4(const void* base, int dtype, size_t off) {
  if (dtype == 0) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(base) + off);
    return make_float4(bf2f((bf16_t)(u.x & 0xffff)), bf2f((bf16_t)(u.x >> 16)), bf2f((bf16_t)(u.y & 0xffff)), bf2f((bf16_t)(u.y >> 16)));
  }
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
}
__device__ __forceinline__ void st4(void* base, int dtype, size_t off, float4 v) {
  if (dtype == 0) {
    uint2 u; u.x = pack_bf2(v.x, v.y); u.y = pack_bf2(v.z, v.w);
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(base) + off) = u;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off) = v;
  }
}
__device__ __forceinline__ float ld1(const void* base, int dtype, size_t off) {
  return dtype == 0 ? bf2f(reinterpret_cast<const bf16_t*>(base)[off]) : reinterpret_cast<const float*>(base)[off];
}
__device__ __forceinline__ void st1(void* base, int dtype, size_t off, float v) {
  if (dtype == 0) reinterpret_cast<bf16_t*>(base)[off] = f2bf(v); else reinterpret_cast<float*>(base)[off] = v;
}

struct RowP {
  void* dst; int dst_dtype, ld_dst; const int* dst_idx;
  const void* src; int src_dtype, ld_src; const int* src_idx;
  const void* add; int add_dtype, ld_add; const int* add_idx;
  int n_rows, c, accumulate;
};

template <int VEC>
__device__ __forceinline__ void row_copy_body(const RowP& p) {
  const int per_row = p.c / VEC;
  const long long total = (long long)p.n_rows * per_row;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / per_row), col = (int)(i % per_row) * VEC;
    const int dr = p.dst_idx ? p.dst_idx[row] : row;
    const int sr = p.src_idx ? p.src_idx[row] : row;
    const int ar = p.add_idx ? p.add_idx[row] : row;
    if (VEC == 4) {
      float4 v = p.src ? ld4(p.src, p.src_dtype, (size_t)sr * p.ld_src + col) : make_float4(0, 0, 0, 0);
      if (p.add) { const float4 a = ld4(p.add, p.add_dtype, (size_t)ar * p.ld_add + col); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
      if (p.accumulate) { const float4 a = ld4(p.dst, p.dst_dtype, (size_t)dr * p.ld_dst + col); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
      st4(p.dst, p.dst_dtype, (size_t)dr * p.ld_dst + col, v);
    } else {
      float v = p.src ? ld1(p.src, p.src_dtype, (size_t)sr * p.ld_src + col) : 0.f;
      if (p.add) v += ld1(p.add, p.add_dtype, (size_t)ar * p.ld_add + col);
      if (p.accumulate) v += ld1(p.dst, p.dst_dtype, (size_t)dr * p.ld_dst + col);
      st1(p.dst, p.dst_dtype, (size_t)dr * p.ld_dst + col, v);
    }
  }
}
template <int VEC>
__global__ void row_copy_kernel(RowP p) { row_copy_body<VEC>(p); }
template <int VEC>
__global__ void row_copy_lanes_kernel(Lanes<RowP> L) { row_copy_body<VEC>(L.p[blockIdx.y]); }
// several independent row copies in one grid (blockIdx.y = segment): the token assembly / split ops of a step are 4-9 small copies each
struct RowMultiP { RowP seg[12]; };
__global__ __launch_bounds__(256) void row_copy_multi_kernel(RowMultiP m) {
  const RowP& p = m.seg[blockIdx.y];
  const bool vec = !(p.c & 3) && !(p.ld_dst & 3) && (!p.src || !(p.ld_src & 3)) && (!p.add || !(p.ld_add & 3)) && !(((uintptr_t)p.dst) & 15) &&
                   (!p.src || !(((uintptr_t)p.src) & 15)) && (!p.add || !(((uintptr_t)p.add) & 15));
  const int vw = vec ? 4 : 1;
  const int per_row = p.c / vw;
  const long long total = (long long)p.n_rows * per_row;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int row = (int)(i / per_row), col = (int)(i % per_row) * vw;
    const int dr = p.dst_idx ? p.dst_idx[row] : row;
    const int sr = p.src_idx ? p.src_idx[row] : row;
    const int ar = p.add_idx ? p.add_idx[row] : row;
    if (vec) {
      float4 v = p.src ? ld4(p.src, p.src_dtype, (size_t)sr * p.ld_src + col) : make_float4(0, 0, 0, 0);
      if (p.add) { const float4 a = ld4(p.add, p.add_dtype, (size_t)ar * p.ld_add + col); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
      if (p.accumulate) { const float4 a = ld4(p.dst, p.dst_dtype, (size_t)dr * p.ld_dst + col); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
      st4(p.dst, p.dst_dtype, (size_t)dr * p.ld_dst + col, v);
    } else {
      float v = p.src ? ld1(p.src, p.src_dtype, (size_t)sr * p.ld_src + col) : 0.f;
      if (p.add) v += ld1(p.add, p.add_dtype, (size_t)ar * p.ld_add + col);
      if (p.accumulate) v += ld1(p.dst, p.dst_dtype, (size_t)dr * p.ld_dst + col);
      st1(p.dst, p.dst_dtype, (size_t)dr * p.ld_dst + col, v);
    }
  }
}

__device__ __forceinline__ void cast_body(const void* src, int sd, void* dst, int dd, long long n) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    st4(dst, dd, (size_t)i * 4, ld4(src, sd, (size_t)i * 4));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) st1(dst, dd, (size_t)(n4 * 4 + threadIdx.x), ld1(src, sd, (size_t)(n4 * 4 + threadIdx.x)));
}
struct CastP { const void* src; int sd; void* dst; int dd; long long n; };
__global__ void cast_kernel(CastP q) { cast_body(q.src, q.sd, q.dst, q.dd, q.n); }
__global__ void cast_lanes_kernel(Lanes<CastP> L) { const CastP& q = L.p[blockIdx.y]; cast_body(q.src, q.sd, q.dst, q.dd, q.n); }
// dst[c][r] = src[r][c]; 64x64 tile through LDS (+1 pad)
__global__ __launch_bounds__(256) void transpose_cast_kernel(const void* src, int sd, int rows, int cols, bf16_t* dst) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? ld1(src, sd, (size_t)(r0 + r) * cols + c0 + c) : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < rows && c0 + c < cols) dst[(size_t)(c0 + c) * rows + r0 + r] = f2bf(tile[r][c]);
  }
}

__global__ void gelu_fwd_kernel(const bf16_t* x, bf16_t* y, long long n) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = ld4(x, 0, (size_t)i * 4);
    v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w);
    st4(y, 0, (size_t)i * 4, v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[n4 * 4 + threadIdx.x] = f2bf(gelu_f(bf2f(x[n4 * 4 + threadIdx.x])));
}
__global__ void gelu_bwd_kernel(const bf16_t* x, const bf16_t* dy, bf16_t* dx, long long n) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = ld4(x, 0, (size_t)i * 4);
    float4 d = ld4(dy, 0, (size_t)i * 4);
    d.x *= gelu_grad_f(v.x); d.y *= gelu_grad_f(v.y); d.z *= gelu_grad_f(v.z); d.w *= gelu_grad_f(v.w);
    st4(dx, 0, (size_t)i * 4, d);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t j = n4 * 4 + threadIdx.x;
    dx[j] = f2bf(bf2f(dy[j]) * gelu_grad_f(bf2f(x[j])));
  }
}

int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}
bool a16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Segment mean over consecutive row blocks (token pooling of the classification heads, reference cinema/convvit.py:523-547):
// out[s][c] = scale * sum_{r < seg_rows} x[s * seg_rows + r][c]; thread = one column of one segment, grid.y = segment.
__global__ __launch_bounds__(256) void segment_mean_fwd_kernel(const float* x, int ldx, int seg_rows, int c, float scale, float* out) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= c) return;
  const float* src = x + (size_t)blockIdx.y * seg_rows * ldx + col;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = 0;
  for (; r + 3 < seg_rows; r += 4) { s0 += src[(size_t)r * ldx]; s1 += src[(size_t)(r + 1) * ldx]; s2 += src[(size_t)(r + 2) * ldx]; s3 += src[(size_t)(r + 3) * ldx]; }
  for (; r < seg_rows; r++) s0 += src[(size_t)r * ldx];
  out[(size_t)blockIdx.y * c + col] = ((s0 + s1) + (s2 + s3)) * scale;
}
// dx[s * seg_rows + r][c] (+)= scale * dy[s][c]
__global__ __launch_bounds__(256) void segment_mean_bwd_kernel(const float* dy, int seg_rows, int c, float scale, float* dx, int lddx, int accumulate) {
  const long long total = (long long)gridDim.y * seg_rows * c;
  (void)total;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= c) return;
  const float g = dy[(size_t)blockIdx.y * c + col] * scale;
  float* dst = dx + (size_t)blockIdx.y * seg_rows * lddx + col;
  for (int r = 0; r < seg_rows; r++) {
    if (accumulate) dst[(size_t)r * lddx] += g; else dst[(size_t)r * lddx] = g;
  }
}
// y[r][j] = a[r][j] * (b_rows ? b[r][j] : b[j]): timm LayerScale (x * gamma, cinema/vit.py:561,576 - forward, and dx = dy * gamma), and with a full second operand the
// element products behind d gamma = column sums of dy * x and the GELU derivative of an unfused Mlp.  Contiguous rows of c elements; fp32 / bf16 operands and result.
struct MulRowsP { const void* a; const void* b; void* y; long long n; int c; int a_bf16, b_bf16, b_rows, y_bf16; };
__device__ __forceinline__ float ld_any(const void* p, long long i, int is_bf16) {
  return is_bf16 ? bf2f(reinterpret_cast<const bf16_t*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}
__global__ __launch_bounds__(256) void mul_rows_kernel(MulRowsP p) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.n; i += (long long)gridDim.x * 256) {
    const float v = ld_any(p.a, i, p.a_bf16) * ld_any(p.b, p.b_rows ? i : i % p.c, p.b_bf16);
    if (p.y_bf16) reinterpret_cast<bf16_t*>(p.y)[i] = f2bf(v);
    else reinterpret_cast<float*>(p.y)[i] = v;
  }
}

__global__ __launch_bounds__(256) void scale_f32_kernel(const float* x, float alpha, float* y, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = x[i] * alpha;
}

// Conv weight (out, c, *k) fp32 <-> GEMM operand rows [out][ld] in the patch feature order (*k, c):
//   direction 0: rows_bf16[o][jj * c + i] = w[(o * c + i) * kvol + jm]   (tail of the row up to ld zero-filled)
//   direction 1: w_grad[(o * c + i) * kvol + jm] += rows_f32[o][jj * c + i]
// with jm = jmap ? jmap[jj] : jj (voxel permutation of the visible-voxel stem).  One launch replaces permute + contiguous + cast
// (forward, every step for every k == s conv) and permute + contiguous + accumulate (end of the backward pass).
__device__ __forceinline__ void patch_weight_relayout_body(float* w, void* rows, int rows_bf16, int outer, int c, int kvol, int ld, const int* jmap,
                                                                   int direction) {
  const long long total = (long long)outer * ld;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int o = (int)(idx / ld), f = (int)(idx - (long long)o * ld);
    if (f >= kvol * c) {
      if (direction == 0) { if (rows_bf16) reinterpret_cast<bf16_t*>(rows)[idx] = 0; else reinterpret_cast<float*>(rows)[idx] = 0.f; }
      continue;
    }
    const int jj = f / c, i = f - jj * c;
    const int jm = jmap ? jmap[jj] : jj;
    const size_t wi = ((size_t)o * c + i) * kvol + jm;
    if (direction == 0) {
      if (rows_bf16) reinterpret_cast<bf16_t*>(rows)[idx] = f2bf(w[wi]); else reinterpret_cast<float*>(rows)[idx] = w[wi];
    } else {
      w[wi] += reinterpret_cast<const float*>(rows)[idx];
    }
  }
}
struct RelayoutP { float* w; void* rows; int rows_bf16; int outer; int c; int kvol; int ld; const int* jmap; int direction; };
__global__ __launch_bounds__(256) void patch_weight_relayout_kernel(RelayoutP q) { patch_weight_relayout_body(q.w, q.rows, q.rows_bf16, q.outer, q.c, q.kvol, q.ld, q.jmap, q.direction); }
__global__ __launch_bounds__(256) void patch_weight_relayout_lanes_kernel(Lanes<RelayoutP> L) { const RelayoutP& q = L.p[blockIdx.y]; patch_weight_relayout_body(q.w, q.rows, q.rows_bf16, q.outer, q.c, q.kvol, q.ld, q.jmap, q.direction); }
__device__ __forceinline__ void fill_u32_body(uint32_t* dst, uint32_t word, long long n_words) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (long long)gridDim.x * 256) dst[i] = word;
}
struct FillP { uint32_t* dst; uint32_t word; long long n_words; };
__global__ __launch_bounds__(256) void fill_u32_kernel(FillP q) { fill_u32_body(q.dst, q.word, q.n_words); }
__global__ __launch_bounds__(256) void fill_u32_lanes_kernel(Lanes<FillP> L) { const FillP& q = L.p[blockIdx.y]; fill_u32_body(q.dst, q.word, q.n_words); }
__global__ __launch_bounds__(256) void mul_scalar_kernel(const float* x, const float* s, float* y, long long n) {
  const float f = s[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = x[i] * f;
}

}  // namespace

CINEMA_API int cinema_hip_info(int* out) {
  if (!out) return CINEMA_ERR_BAD_ARG;
  for (int i = 0; i < 8; i++) out[i] = 0;
  out[0] = 1;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
    out[1] = prop.multiProcessorCount; out[2] = (int)prop.sharedMemPerBlock; out[3] = prop.warpSize;
  }
  (void)hipGetLastError();
  return 0;
}

CINEMA_API int cinema_row_copy(void* dst, int dst_dtype, int ld_dst, const int* dst_idx, const void* src, int src_dtype, int ld_src,
                               const int* src_idx, const void* add, int add_dtype, int ld_add, const int* add_idx, int n_rows, int c,
                               int accumulate, void* stream) {
  if (!dst || n_rows <= 0 || c <= 0 || (!src && !add)) return CINEMA_ERR_BAD_ARG;
  RowP p{dst, dst_dtype, ld_dst, dst_idx, src, src_dtype, ld_src, src_idx, add, add_dtype, ld_add, add_idx, n_rows, c, accumulate};
  const bool vec = !(c & 3) && !(ld_dst & 3) && (!src || !(ld_src & 3)) && (!add || !(ld_add & 3)) && a16(dst) && (!src || a16(src)) && (!add || a16(add));
  if (vec) launch_lanes(row_copy_kernel<4>, row_copy_lanes_kernel<4>, 1, dim3(grid_for((long long)n_rows * c / 4, 256)), dim3(256), 0, (hipStream_t)stream, p);
  else launch_lanes(row_copy_kernel<1>, row_copy_lanes_kernel<1>, 1, dim3(grid_for((long long)n_rows * c, 256)), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

CINEMA_API int cinema_row_copy_multi(const cinema_row_copy_args* segs, int count, void* stream) {
  if (!segs || count <= 0) return CINEMA_ERR_BAD_ARG;
  for (int i0 = 0; i0 < count; i0 += 12) {
    RowMultiP m;
    const int n = count - i0 < 12 ? count - i0 : 12;
    long long most = 1;
    for (int i = 0; i < n; i++) {
      const cinema_row_copy_args& a = segs[i0 + i];
      if (!a.dst || a.n_rows <= 0 || a.c <= 0 || (!a.src && !a.add)) return CINEMA_ERR_BAD_ARG;
      m.seg[i] = RowP{a.dst, a.dst_dtype, a.ld_dst, a.dst_idx, a.src, a.src_dtype, a.ld_src, a.src_idx, a.add, a.add_dtype, a.ld_add, a.add_idx, a.n_rows, a.c,
                      a.accumulate};
      const long long work = (long long)a.n_rows * a.c / 4 + 1;
      if (work > most) most = work;
    }
    CINEMA_LAUNCH(row_copy_multi_kernel, dim3(grid_for(most, 256), n), dim3(256), 0, (hipStream_t)stream, m);
  }
  return launch_status();
}

CINEMA_API int cinema_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long long n, void* stream) {
  if (!src || !dst || n <= 0) return CINEMA_ERR_BAD_ARG;
  if (!a16(src) || !a16(dst)) return CINEMA_ERR_UNSUPPORTED;
  launch_lanes(cast_kernel, cast_lanes_kernel, 1, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, CastP{src, src_dtype, dst, dst_dtype, n});
  return launch_status();
}

CINEMA_API int cinema_transpose_cast(const void* src, int src_dtype, int rows, int cols, uint16_t* dst, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(transpose_cast_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, src, src_dtype, rows, cols, dst);
  return launch_status();
}

CINEMA_API int cinema_gelu_fwd(const uint16_t* x, uint16_t* y, long long n, void* stream) {
  if (!x || !y || n <= 0) return CINEMA_ERR_BAD_ARG;
  if (!a16(x) || !a16(y)) return CINEMA_ERR_UNSUPPORTED;
  CINEMA_LAUNCH(gelu_fwd_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  return launch_status();
}
CINEMA_API int cinema_gelu_bwd(const uint16_t* x, const uint16_t* dy, uint16_t* dx, long long n, void* stream) {
  if (!x || !dy || !dx || n <= 0) return CINEMA_ERR_BAD_ARG;
  if (!a16(x) || !a16(dy) || !a16(dx)) return CINEMA_ERR_UNSUPPORTED;
  CINEMA_LAUNCH(gelu_bwd_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n);
  return launch_status();
}

CINEMA_API int cinema_segment_mean_fwd(const float* x, int ldx, int n_seg, int seg_rows, int c, float scale, float* out, void* stream) {
  if (!x || !out || n_seg <= 0 || seg_rows <= 0 || c <= 0) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(segment_mean_fwd_kernel, dim3((c + 255) / 256, n_seg), dim3(256), 0, (hipStream_t)stream, x, ldx, seg_rows, c, scale, out);
  return launch_status();
}

CINEMA_API int cinema_segment_mean_bwd(const float* dy, int n_seg, int seg_rows, int c, float scale, float* dx, int lddx, int accumulate, void* stream) {
  if (!dy || !dx || n_seg <= 0 || seg_rows <= 0 || c <= 0) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(segment_mean_bwd_kernel, dim3((c + 255) / 256, n_seg), dim3(256), 0, (hipStream_t)stream, dy, seg_rows, c, scale, dx, lddx, accumulate);
  return launch_status();
}

CINEMA_API int cinema_mul_rows(const void* a, int a_bf16, const void* b, int b_bf16, int b_rows, void* y, int y_bf16, long long rows, int c, void* stream) {
  if (!a || !b || !y || rows <= 0 || c <= 0) return CINEMA_ERR_BAD_ARG;
  const long long n = rows * c;
  CINEMA_LAUNCH(mul_rows_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, MulRowsP{a, b, y, n, c, a_bf16, b_bf16, b_rows, y_bf16});
  return launch_status();
}

CINEMA_API int cinema_scale_f32(const float* x, float alpha, float* y, long long n, void* stream) {
  if (!x || !y || n <= 0) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(scale_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, alpha, y, n);
  return launch_status();
}

CINEMA_API int cinema_patch_weight_relayout(float* w, void* rows, int rows_is_bf16, int outer, int c, int kvol, int ld, const int* jmap, int direction,
                                            void* stream) {
  if (!w || !rows || outer <= 0 || c <= 0 || kvol <= 0 || ld < c * kvol || (direction != 0 && direction != 1)) return CINEMA_ERR_BAD_ARG;
  if (direction == 1 && rows_is_bf16) return CINEMA_ERR_UNSUPPORTED;
  launch_lanes(patch_weight_relayout_kernel, patch_weight_relayout_lanes_kernel, 1, dim3(grid_for((long long)outer * ld, 256)), dim3(256), 0, (hipStream_t)stream,
               RelayoutP{w, rows, rows_is_bf16, outer, c, kvol, ld, jmap, direction});
  return launch_status();
}

// dst[0 .. n_words) = word (32-bit pattern: 0 for any zero tensor, the bits of a float for fp32 fills).  A kernel rather than hipMemsetAsync so
// that it is an ordinary launch of the replayable call list (cinema_amd/replay.py) like everything else on the path.
CINEMA_API int cinema_fill_u32(void* dst, unsigned int word, long long n_words, void* stream) {
  if (!dst || n_words < 0 || (((uintptr_t)dst) & 3)) return CINEMA_ERR_BAD_ARG;
  if (n_words == 0) return 0;
  launch_lanes(fill_u32_kernel, fill_u32_lanes_kernel, 1, dim3(grid_for(n_words, 256)), dim3(256), 0, (hipStream_t)stream, FillP{(uint32_t*)dst, (uint32_t)word, n_words});
  return launch_status();
}

// Rotary embedding as the reference calls it (cinema/vit.py:496-499 -> cinema/rotary.py:30-60): q and k arrive as (batch, heads, tokens,
// head_dim) and RotaryEmbedding indexes its table with dim 1, so the angle depends on the HEAD index h (table row h), the same for every
// token: x1' = x1 cos - x2 sin, x2' = x2 cos + x1 sin over the two halves of the rotated part (rotate_half, rotary.py:12-24).  In place on bf16
// rows [rows, ld] holding n_slots consecutive head slots of hd columns from column 0 (slot s uses table row s % heads: the fused q|k
// projection is 2*heads slots).  inverse = 1 applies the transposed rotation (the gradient of the forward one).  half = rotated pairs per head.
template <int VEC>
__global__ __launch_bounds__(256) void rope_heads_kernel(bf16_t* x, int ld, long long rows, int n_slots, int heads, int hd, int half, const float* cos_t,
                                                         const float* sin_t, int inverse) {
  const int chunks = half / VEC;                       // per head slot
  const long long total = rows * n_slots * chunks;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % chunks);
    const long long t = i / chunks;
    const int slot = (int)(t % n_slots);
    const long long r = t / n_slots;
    const int h = slot % heads;
    bf16_t* p1 = x + r * ld + (long long)slot * hd + ch * VEC;
    bf16_t* p2 = p1 + half;
    const float* cs = cos_t + (long long)h * half + ch * VEC;
    const float* sn = sin_t + (long long)h * half + ch * VEC;
    if constexpr (VEC == 8) {
      bf16x8 a = *reinterpret_cast<const bf16x8*>(p1), b = *reinterpret_cast<const bf16x8*>(p2);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float c = cs[j], s_ = inverse ? -sn[j] : sn[j];
        const float x1 = bf2f(a.v[j]), x2 = bf2f(b.v[j]);
        a.v[j] = f2bf(x1 * c - x2 * s_);
        b.v[j] = f2bf(x2 * c + x1 * s_);
      }
      *reinterpret_cast<bf16x8*>(p1) = a;
      *reinterpret_cast<bf16x8*>(p2) = b;
    } else {
      const float c = cs[0], s_ = inverse ? -sn[0] : sn[0];
      const float x1 = bf2f(p1[0]), x2 = bf2f(p2[0]);
      p1[0] = f2bf(x1 * c - x2 * s_);
      p2[0] = f2bf(x2 * c + x1 * s_);
    }
  }
}

CINEMA_API int cinema_rope_heads(uint16_t* x, int ld, long long rows, int n_slots, int heads, int head_dim, int rotary_dim, const float* cos_table,
                                 const float* sin_table, int inverse, void* stream) {
  if (!x || !cos_table || !sin_table || rows <= 0 || n_slots <= 0 || heads <= 0 || head_dim <= 0 || rotary_dim <= 0 || (rotary_dim & 1) ||
      rotary_dim > head_dim || ld < n_slots * head_dim)
    return CINEMA_ERR_BAD_ARG;
  const int half = rotary_dim / 2;
  const bool vec = (half % 8 == 0) && (head_dim % 8 == 0) && (ld % 8 == 0) && ((((uintptr_t)x) & 15) == 0);
  const long long total = rows * n_slots * (vec ? half / 8 : half);
  if (vec)
    CINEMA_LAUNCH(rope_heads_kernel<8>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, ld, rows, n_slots, heads, head_dim, half,
                       cos_table, sin_table, inverse);
  else
    CINEMA_LAUNCH(rope_heads_kernel<1>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, ld, rows, n_slots, heads, head_dim, half,
                       cos_table, sin_table, inverse);
  return launch_status();
}

// ---- stochastic regularisers of the fine-tuning recipes (nn.Dropout in ConvResBlock, cinema/conv.py:329,343; timm DropPath in Block,
// cinema/vit.py:561-577, 606-609; both 0.1 in cinema/segmentation/acdc/config.yaml:64-65).  Counter-based RNG (Philox4x32-10, the generator
// family torch's CUDA dropout uses): the keep decision of element i of call site `salt` in step `state[0]` is a pure function of
// (seed, step, salt, i), so the backward pass regenerates the mask instead of storing it, and a recorded step (cinema_amd/replay.py) gets new
// masks on every replay because the step counter lives in device memory (cinema_rng_advance is one more launch of the list).
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x, hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += W0; key.y += W1;
  }
  return ctr;
}
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }  // [0, 1), 24 bits

// state[0] = step counter (advanced once per optimisation step), state[1] = seed
__global__ void rng_advance_kernel(unsigned long long* state) {
  if (threadIdx.x == 0 && blockIdx.x == 0) state[0] += 1;
}

// y = x * keep_mask / (1 - p) on bf16, 8 elements per thread (two Philox draws); forward and backward are the same map
__global__ __launch_bounds__(256) void dropout_bf16_kernel(const bf16_t* x, bf16_t* y, long long n, float p, const unsigned long long* state, uint32_t salt) {
  const unsigned long long step = state[0], seed = state[1];
  const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  const float inv_keep = 1.0f / (1.0f - p);
  const long long n8 = (n + 7) >> 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const uint4 r0 = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)(i >> 32), salt, (uint32_t)step), key);
    const uint4 r1 = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)(i >> 32) | 0x80000000u, salt, (uint32_t)step), key);
    const uint32_t r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    if (i * 8 + 8 <= n) {
      bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) v.v[j] = u01(r[j]) >= p ? f2bf(bf2f(v.v[j]) * inv_keep) : (bf16_t)0;
      *reinterpret_cast<bf16x8*>(y + i * 8) = v;
    } else {
      for (int j = 0; i * 8 + j < n; ++j) y[i * 8 + j] = u01(r[j]) >= p ? f2bf(bf2f(x[i * 8 + j]) * inv_keep) : (bf16_t)0;
    }
  }
}

// DropPath factors: scale[b] = (u >= p) / (1 - p) per sample (timm DropPath, scale_by_keep=True)
__global__ void droppath_scale_kernel(float* scale, int batch, float p, const unsigned long long* state, uint32_t salt) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const unsigned long long step = state[0], seed = state[1];
  const uint4 r = philox4x32_10(make_uint4((uint32_t)b, 0x40000000u, salt, (uint32_t)step), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  scale[b] = u01(r.x) >= p ? 1.0f / (1.0f - p) : 0.f;
}

// out[r, :] = (res ? res[r, :] : 0) + scale[r / rows_per_sample] * h[r, :]   (fp32 rows; the residual add behind a DropPath; backward: res = nullptr)
__global__ __launch_bounds__(256) void scale_rows_add_kernel(const float* h, const float* res, const float* scale, float* out, long long rows, int c,
                                                             int rows_per_sample) {
  const int c4 = c >> 2;
  const long long total = rows * c4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c4;
    const float s = scale[r / rows_per_sample];
    float4 v = reinterpret_cast<const float4*>(h)[i];
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    if (res) { const float4 q = reinterpret_cast<const float4*>(res)[i]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

// out16[r, :] = bf16(scale[r / rows_per_sample] * h[r, :]): the gradient of the DropPath branch, whose only readers are the branch's weight- and data-gradient GEMMs
__global__ __launch_bounds__(256) void scale_rows_bf16_kernel(const float* h, const float* scale, bf16_t* out, long long rows, int c, int rows_per_sample) {
  const int c4 = c >> 2;
  const long long total = rows * c4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c4;
    const float s = scale[r / rows_per_sample];
    const float4 v = reinterpret_cast<const float4*>(h)[i];
    uint2 pk;
    pk.x = pack_bf2(v.x * s, v.y * s); pk.y = pack_bf2(v.z * s, v.w * s);
    reinterpret_cast<uint2*>(out)[i] = pk;
  }
}

CINEMA_API int cinema_rng_advance(unsigned long long* state, void* stream) {
  if (!state) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(rng_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state);
  return launch_status();
}

CINEMA_API int cinema_dropout_bf16(const uint16_t* x, uint16_t* y, long long n, float p, const unsigned long long* state, unsigned int salt, void* stream) {
  if (!x || !y || !state || n <= 0 || !(p >= 0.f) || !(p < 1.f)) return CINEMA_ERR_BAD_ARG;
  if ((((uintptr_t)x) | ((uintptr_t)y)) & 15) return CINEMA_ERR_UNSUPPORTED;
  CINEMA_LAUNCH(dropout_bf16_kernel, dim3(grid_for((n + 7) / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, p, state, salt);
  return launch_status();
}

CINEMA_API int cinema_droppath_scale(float* scale, int batch, float p, const unsigned long long* state, unsigned int salt, void* stream) {
  if (!scale || !state || batch <= 0 || !(p >= 0.f) || !(p < 1.f)) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(droppath_scale_kernel, dim3((batch + 63) / 64), dim3(64), 0, (hipStream_t)stream, scale, batch, p, state, salt);
  return launch_status();
}

CINEMA_API int cinema_scale_rows_add(const float* h, const float* residual, const float* scale, float* out, long long rows, int c, int rows_per_sample,
                                     void* stream) {
  if (!h || !scale || !out || rows <= 0 || c <= 0 || rows_per_sample <= 0) return CINEMA_ERR_BAD_ARG;
  if ((c & 3) || ((((uintptr_t)h) | ((uintptr_t)out) | ((uintptr_t)residual)) & 15)) return CINEMA_ERR_UNSUPPORTED;
  CINEMA_LAUNCH(scale_rows_add_kernel, dim3(grid_for(rows * (c >> 2), 256)), dim3(256), 0, (hipStream_t)stream, h, residual, scale, out, rows, c,
                     rows_per_sample);
  return launch_status();
}

CINEMA_API int cinema_scale_rows_bf16(const float* h, const float* scale, uint16_t* out, long long rows, int c, int rows_per_sample, void* stream) {
  if (!h || !scale || !out || rows <= 0 || c <= 0 || rows_per_sample <= 0) return CINEMA_ERR_BAD_ARG;
  if ((c & 3) || (((uintptr_t)h) & 15) || (((uintptr_t)out) & 7)) return CINEMA_ERR_UNSUPPORTED;
  CINEMA_LAUNCH(scale_rows_bf16_kernel, dim3(grid_for(rows * (c >> 2), 256)), dim3(256), 0, (hipStream_t)stream, h, scale, out, rows, c, rows_per_sample);
  return launch_status();
}

// ---- per-tensor e4m3 quantisation (fp8 forward GEMMs, BASELINE config 5) -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void amax_bf16_kernel(const bf16_t* x, long long n8, unsigned int* amax_bits) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[i];
#pragma unroll
    for (int j = 0; j < 8; j++) m = fmaxf(m, fabsf(bf2f(v.v[j])));
  }
  __shared__ float part[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {  // ONE atomic per block (thousands of same-address atomics serialise: the first form took 100 us for 17 MB)
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    if (m > 0.f) atomicMax(amax_bits, __float_as_uint(m));  // non-negative floats order like their bit patterns
  }
}

__global__ __launch_bounds__(256) void quantize_fp8_kernel(const bf16_t* x, long long n8, uint8_t* y, const unsigned int* amax_bits, float* scale_out) {
  const float amax = __uint_as_float(amax_bits[0]);
  const float scale = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / scale;
  if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = scale;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[i];
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.v[0]) * inv, bf2f(v.v[1]) * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.v[2]) * inv, bf2f(v.v[3]) * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.v[4]) * inv, bf2f(v.v[5]) * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.v[6]) * inv, bf2f(v.v[7]) * inv, hi, true);
    reinterpret_cast<int2*>(y)[i] = make_int2(lo, hi);
  }
}

// the same over many segments of one flat bf16 buffer at once (the weight shadows of a whole model: one scale per segment), blockIdx.y = segment
__global__ __launch_bounds__(256) void amax_bf16_seg_kernel(const bf16_t* x, const long long* seg, unsigned int* amax_bits) {
  const long long b8 = seg[2 * blockIdx.y] >> 3, e8 = seg[2 * blockIdx.y + 1] >> 3;
  float m = 0.f;
  for (long long i = b8 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < e8; i += (long long)gridDim.x * blockDim.x) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[i];
#pragma unroll
    for (int j = 0; j < 8; j++) m = fmaxf(m, fabsf(bf2f(v.v[j])));
  }
  __shared__ float part[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    if (m > 0.f) atomicMax(amax_bits + blockIdx.y, __float_as_uint(m));
  }
}

__global__ __launch_bounds__(256) void quantize_fp8_seg_kernel(const bf16_t* x, const long long* seg, uint8_t* y, const unsigned int* amax_bits, float* scales) {
  const long long b8 = seg[2 * blockIdx.y] >> 3, e8 = seg[2 * blockIdx.y + 1] >> 3;
  const float amax = __uint_as_float(amax_bits[blockIdx.y]);
  const float scale = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / scale;
  if (blockIdx.x == 0 && threadIdx.x == 0) scales[blockIdx.y] = scale;
  for (long long i = b8 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < e8; i += (long long)gridDim.x * blockDim.x) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[i];
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.v[0]) * inv, bf2f(v.v[1]) * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.v[2]) * inv, bf2f(v.v[3]) * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.v[4]) * inv, bf2f(v.v[5]) * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.v[6]) * inv, bf2f(v.v[7]) * inv, hi, true);
    reinterpret_cast<int2*>(y)[i] = make_int2(lo, hi);
  }
}

// per-ROW e4m3 quantisation of a dense bf16 matrix [rows][c] (c % 8 == 0): one wave per row, two passes over the row from L1/L2 in one launch
__global__ __launch_bounds__(256) void quantize_fp8_rows_kernel(const bf16_t* x, int rows, int c, uint8_t* y, float* row_scale) {
  const int lane = threadIdx.x & 63, c8 = c >> 3;
  for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(x + row * c);
    float m = 0.f;
    for (int i = lane; i < c8; i += 64) {
      const bf16x8 v = src[i];
#pragma unroll
      for (int j = 0; j < 8; j++) m = fmaxf(m, fabsf(bf2f(v.v[j])));
    }
    m = wave_max(m);
    const float scale = m > 0.f ? m * (1.0f / 448.0f) : 1.0f, inv = 1.0f / scale;
    if (lane == 0) row_scale[row] = scale;
    for (int i = lane; i < c8; i += 64) {
      const bf16x8 v = src[i];
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.v[0]) * inv, bf2f(v.v[1]) * inv, lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.v[2]) * inv, bf2f(v.v[3]) * inv, lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.v[4]) * inv, bf2f(v.v[5]) * inv, hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v.v[6]) * inv, bf2f(v.v[7]) * inv, hi, true);
      reinterpret_cast<int2*>(y + row * c)[i] = make_int2(lo, hi);
    }
  }
}

CINEMA_API int cinema_quantize_fp8_rows(const uint16_t* x, int rows, int c, uint8_t* y, float* row_scale, void* stream) {
  if (!x || !y || !row_scale || rows <= 0 || c <= 0) return CINEMA_ERR_BAD_ARG;
  if ((c & 7) || (((uintptr_t)x) & 15) || (((uintptr_t)y) & 7)) return CINEMA_ERR_UNSUPPORTED;
  int g = (rows + 3) / 4;
  if (g > 8192) g = 8192;
  CINEMA_LAUNCH(quantize_fp8_rows_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, rows, c, y, row_scale);
  return launch_status();
}

CINEMA_API int cinema_quantize_fp8_segments(const uint16_t* x, const long long* seg_bounds, int n_seg, uint8_t* y, float* scales, unsigned int* amax_ws,
                                            void* stream) {
  if (!x || !seg_bounds || !y || !scales || !amax_ws || n_seg <= 0) return CINEMA_ERR_BAD_ARG;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)y) & 7)) return CINEMA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  CINEMA_LAUNCH(fill_u32_kernel, dim3(grid_for(n_seg, 256)), dim3(256), 0, st, FillP{(uint32_t*)amax_ws, 0u, n_seg});
  CINEMA_LAUNCH(amax_bf16_seg_kernel, dim3(32, n_seg), dim3(256), 0, st, x, seg_bounds, amax_ws);
  CINEMA_LAUNCH(quantize_fp8_seg_kernel, dim3(32, n_seg), dim3(256), 0, st, x, seg_bounds, y, (const unsigned int*)amax_ws, scales);
  return launch_status();
}

// TRANSPOSED e4m3 shadows of the 2-D weights of a flat bf16 buffer (fp8 data-gradient GEMMs: dX = dY W needs W with the OUTPUT features contiguous):
// segment s = matrix [rows][cols] at element offset desc[3s], written as [cols][rows] bytes at the same offset of yt with the segment's scale from
// cinema_quantize_fp8_segments (same amax).  blockIdx.y = segment, the blocks of a segment walk its 64 x 64 tiles; desc[3s + 1] = rows, + 2 = cols.
__global__ __launch_bounds__(256) void quantize_fp8_seg_t_kernel(const bf16_t* x, const long long* desc, const float* scales, uint8_t* yt) {
  __shared__ float tile[64][65];
  const long long off = desc[3 * blockIdx.y];
  const int rows = (int)desc[3 * blockIdx.y + 1], cols = (int)desc[3 * blockIdx.y + 2];
  const float inv = 1.0f / scales[blockIdx.y];
  const int tr = (rows + 63) / 64, tc = (cols + 63) / 64;
  for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
    const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {  // 64 rows x 8 chunks of 8 columns
      const int r = i >> 3, c8 = (i & 7) * 8;
      if (r0 + r < rows && c0 + c8 < cols) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + off + (long long)(r0 + r) * cols + c0 + c8);
#pragma unroll
        for (int j = 0; j < 8; j++) tile[r][c8 + j] = bf2f(v.v[j]) * inv;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {  // 64 output rows (= source columns) x 8 chunks of 8 source rows
      const int c = i >> 3, r8 = (i & 7) * 8;
      if (c0 + c < cols && r0 + r8 < rows) {
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(tile[r8][c], tile[r8 + 1][c], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(tile[r8 + 2][c], tile[r8 + 3][c], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(tile[r8 + 4][c], tile[r8 + 5][c], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(tile[r8 + 6][c], tile[r8 + 7][c], hi, true);
        *reinterpret_cast<int2*>(yt + off + (long long)(c0 + c) * rows + r0 + r8) = make_int2(lo, hi);
      }
    }
  }
}

CINEMA_API int cinema_quantize_fp8_segments_t(const uint16_t* x, const long long* seg_desc, int n_seg, const float* scales, uint8_t* yt, void* stream) {
  if (!x || !seg_desc || !scales || !yt || n_seg <= 0) return CINEMA_ERR_BAD_ARG;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)yt) & 7)) return CINEMA_ERR_UNSUPPORTED;
  CINEMA_LAUNCH(quantize_fp8_seg_t_kernel, dim3(48, n_seg), dim3(256), 0, (hipStream_t)stream, x, seg_desc, scales, yt);
  return launch_status();
}

// ---- per-tensor DELAYED scaling (cinema_q8_out): maxima -> scales, one wave per site; and the stand-alone producer ------------------------------------
namespace {
__global__ __launch_bounds__(256) void fp8_sites_update_kernel(unsigned int* amax, float* scale, float* inv, int n_sites, float margin) {
  __shared__ float part[4];
  const int s = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;  // one workgroup per site
  unsigned int* slots = amax + (size_t)s * Q8_SLOTS;
  float a = 0.f;
  for (int i = threadIdx.x; i < Q8_SLOTS; i += 256) { a = fmaxf(a, __uint_as_float(slots[i])); slots[i] = 0u; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 64));
  if (lane == 0) part[wave] = a;
  __syncthreads();
  a = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
  if (threadIdx.x == 0 && a > 0.f && a < 3.0e38f) {  // nothing recorded / a non-finite step: keep the previous scale
    const float sc = a * margin * (1.0f / 448.0f);
    scale[s] = sc;
    inv[s] = 1.0f / sc;
  }
}
__global__ __launch_bounds__(256) void quantize_fp8_site_kernel(const bf16_t* x, long long n8, uint8_t* y, const float* inv_p, unsigned int* amax) {
  const float inv = y ? *inv_p : 1.f;
  float mx = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const uint4 u = reinterpret_cast<const uint4*>(x)[i];
    float f[8] = {bf_lo16(u.x), bf_hi16(u.x), bf_lo16(u.y), bf_hi16(u.y), bf_lo16(u.z), bf_hi16(u.z), bf_lo16(u.w), bf_hi16(u.w)};
#pragma unroll
    for (int e = 0; e < 8; e++) mx = fmaxf(mx, fabsf(f[e]));
    if (y) {
      uint2 pk;
      pk.x = (uint32_t)q8_pack4(f[0], f[1], f[2], f[3], inv); pk.y = (uint32_t)q8_pack4(f[4], f[5], f[6], f[7], inv);
      reinterpret_cast<uint2*>(y)[i] = pk;
    }
  }
  q8_amax_commit(amax, mx, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
}
// the same over a [rows][c] matrix with its COLUMN SUMS on the side (colsum[c] += sum over rows): the gradient tensors whose 8-bit copy is the dY operand of a
// weight-gradient GEMM also give that layer's bias gradient - one pass over the bf16 rows instead of two.  Block = 32 column lanes x 8 bf16 x 8 row lanes.
__global__ __launch_bounds__(256) void quantize_fp8_site_cols_kernel(const bf16_t* x, int rows, int c, int ldx, uint8_t* y, const float* inv_p, unsigned int* amax,
                                                                    float* colsum, int rows_per_block) {
  __shared__ float part[8][32][9];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = blockIdx.x * 256 + tx * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  const float inv = y ? *inv_p : 1.f;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float mx = 0.f;
  if (col < c) {
    for (int r = r0 + ty; r < r1; r += 8) {
      const uint4 u = *reinterpret_cast<const uint4*>(x + (size_t)r * ldx + col);
      const float f[8] = {bf_lo16(u.x), bf_hi16(u.x), bf_lo16(u.y), bf_hi16(u.y), bf_lo16(u.z), bf_hi16(u.z), bf_lo16(u.w), bf_hi16(u.w)};
#pragma unroll
      for (int e = 0; e < 8; e++) { acc[e] += f[e]; mx = fmaxf(mx, fabsf(f[e])); }
      if (y) {
        uint2 pk;
        pk.x = (uint32_t)q8_pack4(f[0], f[1], f[2], f[3], inv); pk.y = (uint32_t)q8_pack4(f[4], f[5], f[6], f[7], inv);
        *reinterpret_cast<uint2*>(y + (size_t)r * c + col) = pk;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) part[ty][tx][i] = acc[i];
  __syncthreads();
  const int cc = threadIdx.x;
  if (blockIdx.x * 256 + cc < c) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) s += part[j][cc >> 3][cc & 7];
    unsafeAtomicAdd(colsum + blockIdx.x * 256 + cc, s);
  }
  q8_amax_commit(amax, mx, ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * 4 + (int)(threadIdx.x >> 6));
}
__global__ __launch_bounds__(256) void dequantize_fp8_kernel(const uint8_t* x, long long n8, const float* scale_p, bf16_t* y) {
  const float sc = *scale_p;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const uint2 u = reinterpret_cast<const uint2*>(x)[i];
    const int lo = (int)u.x, hi = (int)u.y;  // (the byte selector of v_cvt_f32_fp8 is an immediate)
    const float f[8] = {__builtin_amdgcn_cvt_f32_fp8(lo, 0) * sc, __builtin_amdgcn_cvt_f32_fp8(lo, 1) * sc, __builtin_amdgcn_cvt_f32_fp8(lo, 2) * sc, __builtin_amdgcn_cvt_f32_fp8(lo, 3) * sc,
                        __builtin_amdgcn_cvt_f32_fp8(hi, 0) * sc, __builtin_amdgcn_cvt_f32_fp8(hi, 1) * sc, __builtin_amdgcn_cvt_f32_fp8(hi, 2) * sc, __builtin_amdgcn_cvt_f32_fp8(hi, 3) * sc};
    uint4 o;
    o.x = pack_bf2(f[0], f[1]); o.y = pack_bf2(f[2], f[3]); o.z = pack_bf2(f[4], f[5]); o.w = pack_bf2(f[6], f[7]);
    reinterpret_cast<uint4*>(y)[i] = o;
  }
}
}  // namespace
CINEMA_API int cinema_dequantize_fp8(const uint8_t* x8, long long n, const float* scale, uint16_t* y, void* stream) {
  if (!x8 || !scale || !y || n <= 0 || (n & 7) || (((uintptr_t)x8) & 7) || (((uintptr_t)y) & 15)) return CINEMA_ERR_BAD_ARG;
  long long g = (n / 8 + 255) / 256;
  if (g > 2048) g = 2048;
  CINEMA_LAUNCH(dequantize_fp8_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x8, n / 8, scale, (bf16_t*)y);
  return launch_status();
}
CINEMA_API int cinema_quantize_fp8_site_colsum(const uint16_t* x, int rows, int c, int ldx, const cinema_q8_out* q8, float* colsum, void* stream) {
  if (!x || rows <= 0 || c <= 0 || (c & 7) || (ldx & 7) || !q8 || !q8->amax_slots || (q8->data && !q8->inv_scale) || !colsum || (((uintptr_t)x) & 15) ||
      (((uintptr_t)q8->data) & 7))
    return CINEMA_ERR_BAD_ARG;
  const int col_blocks = (c + 255) / 256;
  int row_chunks = (1024 + col_blocks - 1) / col_blocks;
  if (row_chunks > (rows + 63) / 64) row_chunks = (rows + 63) / 64;
  const int rpb = (((rows + row_chunks - 1) / row_chunks) + 7) / 8 * 8;
  dim3 grid(col_blocks, (rows + rpb - 1) / rpb);
  CINEMA_LAUNCH(quantize_fp8_site_cols_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, rows, c, ldx, q8->data, q8->inv_scale, q8->amax_slots, colsum, rpb);
  return launch_status();
}
CINEMA_API int cinema_fp8_sites_update(unsigned int* amax_slots, float* scale, float* inv_scale, int n_sites, float margin, void* stream) {
  if (!amax_slots || !scale || !inv_scale || n_sites <= 0 || !(margin >= 1.0f)) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(fp8_sites_update_kernel, dim3(n_sites), dim3(256), 0, (hipStream_t)stream, amax_slots, scale, inv_scale, n_sites, margin);
  return launch_status();
}
CINEMA_API int cinema_quantize_fp8_site(const uint16_t* x, long long n, const cinema_q8_out* q8, void* stream) {
  if (!x || n <= 0 || (n & 7) || !q8 || !q8->amax_slots || (q8->data && !q8->inv_scale) || (((uintptr_t)x) & 15) || (((uintptr_t)q8->data) & 7)) return CINEMA_ERR_BAD_ARG;
  long long g = (n / 8 + 255) / 256;
  if (g > 2048) g = 2048;
  CINEMA_LAUNCH(quantize_fp8_site_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, n / 8, q8->data, q8->inv_scale, q8->amax_slots);
  return launch_status();
}

CINEMA_API int cinema_quantize_fp8(const uint16_t* x, long long n, uint8_t* y, float* scale_out, unsigned int* amax_ws, void* stream) {
  if (!x || !y || !scale_out || !amax_ws || n <= 0) return CINEMA_ERR_BAD_ARG;
  if ((n & 7) || (((uintptr_t)x) & 15) || (((uintptr_t)y) & 7)) return CINEMA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  int g = grid_for(n / 8, 256);
  if (g > 2048) g = 2048;
  int ga = g > 512 ? 512 : g;
  CINEMA_LAUNCH(fill_u32_kernel, dim3(1), dim3(256), 0, st, FillP{(uint32_t*)amax_ws, 0u, 1});
  CINEMA_LAUNCH(amax_bf16_kernel, dim3(ga), dim3(256), 0, st, x, n / 8, amax_ws);
  CINEMA_LAUNCH(quantize_fp8_kernel, dim3(g), dim3(256), 0, st, x, n / 8, y, (const unsigned int*)amax_ws, scale_out);
  return launch_status();
}

// rows[ci][tap * c_out + co] = bf16(w[co][ci][tap]): the B operand of the implicit-GEMM data gradient of a dense convolution (gemm.hip cinema_conv_gemm_bf16)
__global__ __launch_bounds__(256) void conv_weight_dgrad_kernel(const float* w, bf16_t* rows, int c_out, int c_in, int kvol, int ld) {
  const long long total = (long long)c_in * ld;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i / ld), f = (int)(i % ld);
    float v = 0.f;
    if (f < kvol * c_out) {
      const int tap = f / c_out, co = f - tap * c_out;
      v = w[((long long)co * c_in + ci) * kvol + tap];
    }
    rows[i] = f2bf(v);
  }
}

CINEMA_API int cinema_conv_weight_dgrad(const float* w, uint16_t* rows, int c_out, int c_in, int kvol, int ld, void* stream) {
  if (!w || !rows || c_out <= 0 || c_in <= 0 || kvol <= 0 || ld < kvol * c_out) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(conv_weight_dgrad_kernel, dim3(grid_for((long long)c_in * ld, 256)), dim3(256), 0, (hipStream_t)stream, w, rows, c_out, c_in, kvol, ld);
  return launch_status();
}

// ---- transposed-conv weights (k == s up-sampling, tape.op_conv_transpose): w fp32 (c_in, c_out, kvol) <-> GEMM rows [(kv, co)][ci]
// direction 0: rows bf16 <- w (and bias_t[kv * c_out + co] = bias[co]); direction 1: w += rows fp32 (the gradient way back)
__global__ __launch_bounds__(256) void convt_weight_relayout_kernel(float* w, void* rows, int c_in, int c_out, int kvol, int direction, const float* bias,
                                                                     float* bias_t) {
  const long long total = (long long)c_in * c_out * kvol;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % c_in);
    const int r = (int)(i / c_in);  // kv * c_out + co
    const int kv = r / c_out, co = r - kv * c_out;
    const long long wi = ((long long)ci * c_out + co) * kvol + kv;
    if (direction == 0) reinterpret_cast<bf16_t*>(rows)[i] = f2bf(w[wi]);
    else w[wi] += reinterpret_cast<const float*>(rows)[i];
  }
  if (direction == 0 && bias && blockIdx.x == 0)
    for (int j = threadIdx.x; j < kvol * c_out; j += blockDim.x) bias_t[j] = bias[j % c_out];
}
CINEMA_API int cinema_convt_weight_relayout(float* w, void* rows, int c_in, int c_out, int kvol, int direction, const float* bias, float* bias_t, void* stream) {
  if (!w || !rows || c_in <= 0 || c_out <= 0 || kvol <= 0 || (direction != 0 && direction != 1) || (bias && !bias_t)) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(convt_weight_relayout_kernel, dim3(grid_for((long long)c_in * c_out * kvol, 256)), dim3(256), 0, (hipStream_t)stream, w, rows, c_in, c_out, kvol,
                direction, bias, bias_t);
  return launch_status();
}

// ---- z-blocked implicit convolution (gemm.hip cinema_conv_gemm_bf16, conv_zb > 1): block-banded weights and the fold of their gradient
__global__ __launch_bounds__(256) void conv_weight_zblock_kernel(const bf16_t* w, int n, int c, int ld_w, int zb, int transpose, bf16_t* out, const float* bias,
                                                                 float* bias_zb) {
  const int ld_o = 9 * (zb + 2) * c;
  const long long total = (long long)zb * n * ld_o;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / ld_o), f = (int)(i % ld_o);
    const int zo = row / n, co = row - zo * n;
    const int tapz = f / c, ci = f - tapz * c;
    const int txy = tapz / (zb + 2), dzi = tapz - txy * (zb + 2);
    const int tz = transpose ? zo + 2 - dzi : dzi - zo;
    out[i] = (tz >= 0 && tz <= 2) ? w[(long long)co * ld_w + (txy * 3 + tz) * c + ci] : (bf16_t)0;
  }
  if (bias && blockIdx.x == 0)
    for (int j = threadIdx.x; j < zb * n; j += blockDim.x) bias_zb[j] = bias[j % n];
}
__global__ __launch_bounds__(256) void conv_wgrad_zfold_kernel(const float* r, int n, int c, int zb, float* dst, int ld_dst, const float* rowsum_zb, float* db) {
  const int ld_r = 9 * (zb + 2) * c;
  const long long total = (long long)n * 27 * c;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(i / (27 * c)), f = (int)(i % (27 * c));
    const int tap = f / c, ci = f - tap * c;
    const int txy = tap / 3, tz = tap - txy * 3;
    float s = 0.f;
    for (int zo = 0; zo < zb; zo++) s += r[(long long)(zo * n + co) * ld_r + (txy * (zb + 2) + zo + tz) * c + ci];
    dst[(long long)co * ld_dst + f] += s;
  }
  if (db && blockIdx.x == 0)
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      float s = 0.f;
      for (int zo = 0; zo < zb; zo++) s += rowsum_zb[zo * n + j];
      db[j] += s;
    }
}

CINEMA_API int cinema_conv_weight_zblock(const uint16_t* w, int n, int c, int ld_w, int zb, int transpose, uint16_t* w_zb, const float* bias, float* bias_zb,
                                         void* stream) {
  if (!w || !w_zb || n <= 0 || c <= 0 || zb < 2 || zb > 6 || ld_w < 27 * c || (bias && !bias_zb)) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(conv_weight_zblock_kernel, dim3(grid_for((long long)zb * n * 9 * (zb + 2) * c, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, n, c, ld_w,
                zb, transpose, (bf16_t*)w_zb, bias, bias_zb);
  return launch_status();
}
CINEMA_API int cinema_conv_wgrad_zfold(const float* r, int n, int c, int zb, float* dst, int ld_dst, const float* rowsum_zb, float* db, void* stream) {
  if (!r || !dst || n <= 0 || c <= 0 || zb < 2 || zb > 6 || ld_dst < 27 * c || (db && !rowsum_zb)) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(conv_wgrad_zfold_kernel, dim3(grid_for((long long)n * 27 * c, 256)), dim3(256), 0, (hipStream_t)stream, r, n, c, zb, dst, ld_dst, rowsum_zb, db);
  return launch_status();
}

// ---- thin linear layers (the 4-class segmentation head over millions of voxels: N <= 8 outputs, K <= 64 inputs): pure streaming, one thread per row.
// The MFMA GEMM needs N % 8 == 0 and the generic kernel ran these at 0.6 TF (4.5 ms + a 3.3 ms column sum per step of config 4).
constexpr int THIN_MAXN = 8, THIN_MAXK = 64;
struct ThinP { const bf16_t* x; const float* w; const float* bias; float* y; const float* dy; bf16_t* dx; float* dw; float* db; long long rows; int n, k; };

__global__ __launch_bounds__(256) void thin_linear_fwd_kernel(ThinP p) {
  __shared__ float ws[THIN_MAXN * THIN_MAXK + THIN_MAXN];
  for (int i = threadIdx.x; i < p.n * p.k; i += 256) ws[i] = p.w[i];
  for (int i = threadIdx.x; i < p.n; i += 256) ws[p.n * p.k + i] = p.bias ? p.bias[i] : 0.f;
  __syncthreads();
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < p.rows; r += (long long)gridDim.x * 256) {
    float acc[THIN_MAXN];
#pragma unroll
    for (int j = 0; j < THIN_MAXN; j++) acc[j] = j < p.n ? ws[p.n * p.k + j] : 0.f;
    for (int k0 = 0; k0 < p.k; k0 += 8) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(p.x + r * p.k + k0);
#pragma unroll
      for (int j = 0; j < THIN_MAXN; j++)
        if (j < p.n) {
#pragma unroll
          for (int t = 0; t < 8; t++) acc[j] = fmaf(bf2f(v.v[t]), ws[j * p.k + k0 + t], acc[j]);
        }
    }
    for (int j = 0; j < p.n; j++) p.y[r * p.n + j] = acc[j];
  }
}

// dx[r][k] = sum_n dy[r][n] W[n][k] (bf16) ; dW[n][k] += sum_r dy[r][n] x[r][k] ; db[n] += sum_r dy[r][n]   (dy fp32 [rows][n])
__global__ __launch_bounds__(256) void thin_linear_bwd_kernel(ThinP p) {
  __shared__ float ws[THIN_MAXN * THIN_MAXK];
  __shared__ float red[THIN_MAXN * THIN_MAXK + THIN_MAXN];
  for (int i = threadIdx.x; i < p.n * p.k; i += 256) ws[i] = p.w[i];
  for (int i = threadIdx.x; i < p.n * p.k + p.n; i += 256) red[i] = 0.f;
  __syncthreads();
  // weight-gradient partials: lane l of a wave owns column k = l (k <= 64) for every n: a wave walks rows together, x[r][lane] broadcast-free
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gw[THIN_MAXN], gb[THIN_MAXN];
#pragma unroll
  for (int j = 0; j < THIN_MAXN; j++) { gw[j] = 0.f; gb[j] = 0.f; }
  const long long waves = (long long)gridDim.x * 4;
  for (long long r0 = ((long long)blockIdx.x * 4 + wave) * 64; r0 < p.rows; r0 += waves * 64) {
    // phase 1: each lane handles row r0 + lane for the data gradient (and keeps its dy row for the bias gradient)
    const long long r = r0 + lane;
    float d[THIN_MAXN];
#pragma unroll
    for (int j = 0; j < THIN_MAXN; j++) d[j] = (j < p.n && r < p.rows) ? p.dy[r * p.n + j] : 0.f;
#pragma unroll
    for (int j = 0; j < THIN_MAXN; j++) gb[j] += d[j];
    if (p.dx && r < p.rows) {
      for (int k0 = 0; k0 < p.k; k0 += 8) {
        bf16x8 o;
#pragma unroll
        for (int t = 0; t < 8; t++) {
          float a = 0.f;
#pragma unroll
          for (int j = 0; j < THIN_MAXN; j++)
            if (j < p.n) a = fmaf(d[j], ws[j * p.k + k0 + t], a);
          o.v[t] = f2bf(a);
        }
        *reinterpret_cast<bf16x8*>(p.dx + r * p.k + k0) = o;
      }
    }
    // phase 2: weight gradient - lane = input column k, the 64 rows of this group one after the other (dy of row i comes from lane i by shuffle)
    if (p.dw) {
      const int rows_here = (int)min((long long)64, p.rows - r0);
      for (int i = 0; i < rows_here; i++) {
        const float xv = lane < p.k ? bf2f(p.x[(r0 + i) * p.k + lane]) : 0.f;
#pragma unroll
        for (int j = 0; j < THIN_MAXN; j++)
          if (j < p.n) gw[j] = fmaf(__shfl(d[j], i, 64), xv, gw[j]);
      }
    }
  }
  if (p.dw && lane < p.k)
    for (int j = 0; j < p.n; j++) atomicAdd(&red[j * p.k + lane], gw[j]);
  for (int j = 0; j < p.n; j++) {
    const float s = wave_sum(gb[j]);
    if (lane == 0) atomicAdd(&red[p.n * p.k + j], s);
  }
  __syncthreads();
  if (p.dw)
    for (int i = threadIdx.x; i < p.n * p.k; i += 256) unsafeAtomicAdd(p.dw + i, red[i]);
  if (p.db)
    for (int i = threadIdx.x; i < p.n; i += 256) unsafeAtomicAdd(p.db + i, red[p.n * p.k + i]);
}

CINEMA_API int cinema_thin_linear_fwd(const uint16_t* x, const float* w, const float* bias, float* y, long long rows, int n, int k, void* stream) {
  if (!x || !w || !y || rows <= 0 || n <= 0 || k <= 0) return CINEMA_ERR_BAD_ARG;
  if (n > THIN_MAXN || k > THIN_MAXK || (k & 7) || (((uintptr_t)x) & 15)) return CINEMA_ERR_UNSUPPORTED;
  ThinP p{x, w, bias, y, nullptr, nullptr, nullptr, nullptr, rows, n, k};
  CINEMA_LAUNCH(thin_linear_fwd_kernel, dim3(grid_for(rows, 256)), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

CINEMA_API int cinema_thin_linear_bwd(const uint16_t* x, const float* w, const float* dy, uint16_t* dx, float* dw, float* db, long long rows, int n, int k,
                                      void* stream) {
  if (!x || !w || !dy || rows <= 0 || n <= 0 || k <= 0) return CINEMA_ERR_BAD_ARG;
  if (n > THIN_MAXN || k > THIN_MAXK || (k & 7) || (((uintptr_t)x) & 15) || (dx && (((uintptr_t)dx) & 15))) return CINEMA_ERR_UNSUPPORTED;
  ThinP p{x, w, nullptr, nullptr, dy, dx, dw, db, rows, n, k};
  int g = (int)((rows + 255) / 256);
  if (g > 1024) g = 1024;
  CINEMA_LAUNCH(thin_linear_bwd_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

// ---- fan-out linear layers (k <= 8 inputs, n in {4, 8, 16, 32, 64} outputs): the 1x1 shortcut convolution of the raw-image ConvResBlock (1 -> 32 channels over
// every voxel).  The MFMA GEMM needs K % 8 == 0; the generic kernel ran the weight gradient (m = 32, n = 1, k = millions of rows) at 0.1 TF plus a column-sum launch.
constexpr int FAN_MAXN = 64, FAN_MAXK = 8;
__global__ __launch_bounds__(256) void fanout_linear_fwd_kernel(ThinP p) {
  __shared__ float ws[FAN_MAXN * FAN_MAXK + FAN_MAXN];
  for (int i = threadIdx.x; i < p.n * p.k; i += 256) ws[i] = p.w[i];
  for (int i = threadIdx.x; i < p.n; i += 256) ws[p.n * p.k + i] = p.bias ? p.bias[i] : 0.f;
  __syncthreads();
  const int g4 = p.n >> 2;
  const long long total = p.rows * g4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / g4;
    const int q = (int)(i - r * g4) * 4;
    float a[4] = {ws[p.n * p.k + q], ws[p.n * p.k + q + 1], ws[p.n * p.k + q + 2], ws[p.n * p.k + q + 3]};
    for (int kk = 0; kk < p.k; kk++) {
      const float xv = bf2f(p.x[r * p.k + kk]);
#pragma unroll
      for (int j = 0; j < 4; j++) a[j] = fmaf(xv, ws[(q + j) * p.k + kk], a[j]);
    }
    *reinterpret_cast<float4*>(p.y + r * p.n + q) = make_float4(a[0], a[1], a[2], a[3]);
  }
}
// thread = (row lane, group of 4 output columns); the n / 4 threads of one row are adjacent lanes (n / 4 is a power of two <= 16)
__global__ __launch_bounds__(256) void fanout_linear_bwd_kernel(ThinP p) {
  __shared__ float ws[FAN_MAXN * FAN_MAXK];
  __shared__ float red[FAN_MAXN * FAN_MAXK + FAN_MAXN];
  for (int i = threadIdx.x; i < p.n * p.k; i += 256) ws[i] = p.w[i];
  for (int i = threadIdx.x; i < p.n * p.k + p.n; i += 256) red[i] = 0.f;
  __syncthreads();
  const int g4 = p.n >> 2, rl = threadIdx.x / g4, q = (threadIdx.x - rl * g4) * 4, lanes = 256 / g4;
  float gw[4][FAN_MAXK], gb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int kk = 0; kk < FAN_MAXK; kk++) gw[j][kk] = 0.f;
  for (long long r0 = (long long)blockIdx.x * lanes; r0 < p.rows; r0 += (long long)gridDim.x * lanes) {  // block-uniform trip count (shuffles below)
    const long long r = r0 + rl;
    const bool live = r < p.rows;
    const float4 d4 = live ? *reinterpret_cast<const float4*>(p.dy + r * p.n + q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
    for (int j = 0; j < 4; j++) gb[j] += d[j];
#pragma unroll
    for (int kk = 0; kk < FAN_MAXK; kk++) {
      if (kk < p.k) {
        const float xv = live ? bf2f(p.x[r * p.k + kk]) : 0.f;
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          gw[j][kk] = fmaf(d[j], xv, gw[j][kk]);
          a = fmaf(d[j], ws[(q + j) * p.k + kk], a);
        }
        if (p.dx) {
          for (int o = g4 >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
          if (q == 0 && live) p.dx[r * p.k + kk] = f2bf(a);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    atomicAdd(&red[p.n * p.k + q + j], gb[j]);
#pragma unroll
    for (int kk = 0; kk < FAN_MAXK; kk++)
      if (kk < p.k) atomicAdd(&red[(q + j) * p.k + kk], gw[j][kk]);
  }
  __syncthreads();
  if (p.dw)
    for (int i = threadIdx.x; i < p.n * p.k; i += 256) unsafeAtomicAdd(p.dw + i, red[i]);
  if (p.db)
    for (int i = threadIdx.x; i < p.n; i += 256) unsafeAtomicAdd(p.db + i, red[p.n * p.k + i]);
}
static bool fanout_ok(int n, int k) { return k >= 1 && k <= FAN_MAXK && (n == 4 || n == 8 || n == 16 || n == 32 || n == 64); }
CINEMA_API int cinema_fanout_linear_fwd(const uint16_t* x, const float* w, const float* bias, float* y, long long rows, int n, int k, void* stream) {
  if (!x || !w || !y || rows <= 0) return CINEMA_ERR_BAD_ARG;
  if (!fanout_ok(n, k) || (((uintptr_t)y) & 15)) return CINEMA_ERR_UNSUPPORTED;
  ThinP p{x, w, bias, y, nullptr, nullptr, nullptr, nullptr, rows, n, k};
  CINEMA_LAUNCH(fanout_linear_fwd_kernel, dim3(grid_for(rows * (n >> 2), 256)), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}
CINEMA_API int cinema_fanout_linear_bwd(const uint16_t* x, const float* w, const float* dy, uint16_t* dx, float* dw, float* db, long long rows, int n, int k,
                                        void* stream) {
  if (!x || !w || !dy || rows <= 0) return CINEMA_ERR_BAD_ARG;
  if (!fanout_ok(n, k) || (((uintptr_t)dy) & 15)) return CINEMA_ERR_UNSUPPORTED;
  ThinP p{x, w, nullptr, nullptr, dy, dx, dw, db, rows, n, k};
  const int lanes = 256 / (n >> 2);
  long long g = (rows + lanes - 1) / lanes;
  if (g > 2048) g = 2048;
  CINEMA_LAUNCH(fanout_linear_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

// ---- "same" convolution of a ONE-channel volume (the first conv of the raw-image ConvResBlock, cinema/conv.py:320-345 with in_chans = 1): 27 taps x n outputs per
// voxel are too thin for the MFMA path (K = 27), and im2col wrote 32 bf16 per voxel twice per step.  Direct stencil kernels on the fp32 master weights
// [n][taps], kept in LDS as [tap][n].  Kernel extents 1 or 3 per axis; n in {4, 8, 16, 32, 64}.
struct Sten1P {
  const bf16_t* x; const float* w; const float* bias; float* y; const float* dy; bf16_t* dx; float* dw; float* db;
  int b, X, Y, Z, kx, ky, kz, n;
};
__device__ __forceinline__ void sten1_coords(const Sten1P& p, long long r, int& x, int& y, int& z) {
  z = (int)(r % p.Z);
  const long long t = r / p.Z;
  y = (int)(t % p.Y);
  x = (int)((t / p.Y) % p.X);
}
// forward: thread = voxel, all N outputs in registers; per tap one input load, N / 4 broadcast LDS reads of the weights and N FMAs
template <int N>
__global__ __launch_bounds__(256) void sten1_fwd_kernel(Sten1P p) {
  __shared__ __attribute__((aligned(16))) float ws[28 * N];
  const int taps = p.kx * p.ky * p.kz;
  for (int i = threadIdx.x; i < N * taps; i += 256) ws[(i % taps) * N + i / taps] = p.w[i];
  for (int i = threadIdx.x; i < N; i += 256) ws[taps * N + i] = p.bias ? p.bias[i] : 0.f;
  __syncthreads();
  const long long rows = (long long)p.b * p.X * p.Y * p.Z;
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
    int x, y, z;
    sten1_coords(p, r, x, y, z);
    float a[N];
#pragma unroll
    for (int j = 0; j < N; j++) a[j] = ws[taps * N + j];
    int tap = 0;
    for (int tx = 0; tx < p.kx; tx++)
      for (int ty = 0; ty < p.ky; ty++)
        for (int tz = 0; tz < p.kz; tz++, tap++) {
          const int dx = tx - (p.kx >> 1), dy = ty - (p.ky >> 1), dz = tz - (p.kz >> 1);
          if ((unsigned)(x + dx) >= (unsigned)p.X || (unsigned)(y + dy) >= (unsigned)p.Y || (unsigned)(z + dz) >= (unsigned)p.Z) continue;
          const float xv = bf2f(p.x[r + ((long long)dx * p.Y + dy) * p.Z + dz]);
#pragma unroll
          for (int j = 0; j < N; j += 4) {
            const float4 w4 = *reinterpret_cast<const float4*>(ws + tap * N + j);
            a[j] = fmaf(xv, w4.x, a[j]); a[j + 1] = fmaf(xv, w4.y, a[j + 1]); a[j + 2] = fmaf(xv, w4.z, a[j + 2]); a[j + 3] = fmaf(xv, w4.w, a[j + 3]);
          }
        }
#pragma unroll
    for (int j = 0; j < N; j += 4) *reinterpret_cast<float4*>(p.y + r * N + j) = make_float4(a[j], a[j + 1], a[j + 2], a[j + 3]);
  }
}
// weight / bias gradient: thread = (voxel lane 0..7, tap slot 0..31): slot t < taps accumulates dy[r][:] * x[nbr_t(r)] over the lane's voxels in N registers,
// slot `taps` the bias gradient (x = 1); the 32 slots of a voxel are adjacent lanes, so the dy row is one broadcast load per 16 bytes.  Block reduction over
// the 8 voxel lanes through LDS, then one atomic per entry and block.
template <int N>
__global__ __launch_bounds__(256) void sten1_wgrad_kernel(Sten1P p) {
  __shared__ float red[28 * N];
  const int taps = p.kx * p.ky * p.kz;
  for (int i = threadIdx.x; i < 28 * N; i += 256) red[i] = 0.f;
  __syncthreads();
  const int t = threadIdx.x & 31, vl = threadIdx.x >> 5;
  const int tz = t % p.kz, ty = (t / p.kz) % p.ky, tx = t / (p.kz * p.ky);
  const int dx = tx - (p.kx >> 1), dy = ty - (p.ky >> 1), dz = tz - (p.kz >> 1);
  const long long off = ((long long)dx * p.Y + dy) * p.Z + dz;
  float g[N];
#pragma unroll
  for (int j = 0; j < N; j++) g[j] = 0.f;
  const long long rows = (long long)p.b * p.X * p.Y * p.Z;
  if (t <= taps) {
    for (long long r = (long long)blockIdx.x * 8 + vl; r < rows; r += (long long)gridDim.x * 8) {
      float xv = 1.f;  // the bias slot
      if (t < taps) {
        int x, y, z;
        sten1_coords(p, r, x, y, z);
        if ((unsigned)(x + dx) >= (unsigned)p.X || (unsigned)(y + dy) >= (unsigned)p.Y || (unsigned)(z + dz) >= (unsigned)p.Z) continue;
        xv = bf2f(p.x[r + off]);
      }
#pragma unroll
      for (int j = 0; j < N; j += 4) {
        const float4 d = *reinterpret_cast<const float4*>(p.dy + r * N + j);
        g[j] = fmaf(d.x, xv, g[j]); g[j + 1] = fmaf(d.y, xv, g[j + 1]); g[j + 2] = fmaf(d.z, xv, g[j + 2]); g[j + 3] = fmaf(d.w, xv, g[j + 3]);
      }
    }
#pragma unroll
    for (int j = 0; j < N; j++) atomicAdd(&red[t * N + j], g[j]);
  }
  __syncthreads();
  if (p.dw)
    for (int i = threadIdx.x; i < N * taps; i += 256) unsafeAtomicAdd(p.dw + i, red[(i % taps) * N + i / taps]);
  if (p.db)
    for (int i = threadIdx.x; i < N; i += 256) unsafeAtomicAdd(p.db + i, red[taps * N + i]);
}
// data gradient: thread = voxel; dx[r] = sum_tap sum_co dy[r - off(tap)][co] * w[co][tap]
template <int N>
__global__ __launch_bounds__(256) void sten1_dgrad_kernel(Sten1P p) {
  __shared__ __attribute__((aligned(16))) float ws[27 * N];
  const int taps = p.kx * p.ky * p.kz;
  for (int i = threadIdx.x; i < N * taps; i += 256) ws[(i % taps) * N + i / taps] = p.w[i];
  __syncthreads();
  const long long rows = (long long)p.b * p.X * p.Y * p.Z;
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
    int x, y, z;
    sten1_coords(p, r, x, y, z);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int tap = 0;
    for (int tx = 0; tx < p.kx; tx++)
      for (int ty = 0; ty < p.ky; ty++)
        for (int tz = 0; tz < p.kz; tz++, tap++) {
          const int dx = tx - (p.kx >> 1), dy = ty - (p.ky >> 1), dz = tz - (p.kz >> 1);
          if ((unsigned)(x - dx) >= (unsigned)p.X || (unsigned)(y - dy) >= (unsigned)p.Y || (unsigned)(z - dz) >= (unsigned)p.Z) continue;
          const float* drow = p.dy + (r - (((long long)dx * p.Y + dy) * p.Z + dz)) * N;
#pragma unroll
          for (int j = 0; j < N; j += 4) {
            const float4 d = *reinterpret_cast<const float4*>(drow + j);
            const float4 w4 = *reinterpret_cast<const float4*>(ws + tap * N + j);
            a0 = fmaf(d.x, w4.x, a0); a1 = fmaf(d.y, w4.y, a1); a2 = fmaf(d.z, w4.z, a2); a3 = fmaf(d.w, w4.w, a3);
          }
        }
    p.dx[r] = f2bf((a0 + a1) + (a2 + a3));
  }
}
static bool sten1_ok(const Sten1P& p) {
  auto k13 = [](int k) { return k == 1 || k == 3; };
  return p.b > 0 && p.X > 0 && p.Y > 0 && p.Z > 0 && k13(p.kx) && k13(p.ky) && k13(p.kz) && fanout_ok(p.n, 1);
}
#define STEN1_DISPATCH(KERNEL, GRID)                                                                        \
  switch (n) {                                                                                              \
    case 4: CINEMA_LAUNCH(KERNEL<4>, GRID, dim3(256), 0, (hipStream_t)stream, p); break;                    \
    case 8: CINEMA_LAUNCH(KERNEL<8>, GRID, dim3(256), 0, (hipStream_t)stream, p); break;                    \
    case 16: CINEMA_LAUNCH(KERNEL<16>, GRID, dim3(256), 0, (hipStream_t)stream, p); break;                  \
    case 32: CINEMA_LAUNCH(KERNEL<32>, GRID, dim3(256), 0, (hipStream_t)stream, p); break;                  \
    default: CINEMA_LAUNCH(KERNEL<64>, GRID, dim3(256), 0, (hipStream_t)stream, p); break;                  \
  }
CINEMA_API int cinema_conv1ch_fwd(const uint16_t* x, const float* w, const float* bias, float* y, int b, int X, int Y, int Z, int kx, int ky, int kz, int n, void* stream) {
  if (!x || !w || !y) return CINEMA_ERR_BAD_ARG;
  Sten1P p{x, w, bias, y, nullptr, nullptr, nullptr, nullptr, b, X, Y, Z, kx, ky, kz, n};
  if (!sten1_ok(p) || (((uintptr_t)y) & 15)) return CINEMA_ERR_UNSUPPORTED;
  const dim3 grid(grid_for((long long)b * X * Y * Z, 256));
  STEN1_DISPATCH(sten1_fwd_kernel, grid)
  return launch_status();
}
CINEMA_API int cinema_conv1ch_bwd(const uint16_t* x, const float* w, const float* dy, uint16_t* dx, float* dw, float* db, int b, int X, int Y, int Z, int kx, int ky,
                                  int kz, int n, void* stream) {
  if (!x || !w || !dy) return CINEMA_ERR_BAD_ARG;
  Sten1P p{x, w, nullptr, nullptr, dy, dx, dw, db, b, X, Y, Z, kx, ky, kz, n};
  if (!sten1_ok(p) || (((uintptr_t)dy) & 15)) return CINEMA_ERR_UNSUPPORTED;
  const long long rows = (long long)b * X * Y * Z;
  if (dw || db) {
    long long g = (rows + 7) / 8;
    if (g > 2048) g = 2048;  // every block ends with one global atomic per gradient entry
    const dim3 grid((unsigned)g);
    STEN1_DISPATCH(sten1_wgrad_kernel, grid)
  }
  if (dx) {
    const dim3 grid(grid_for(rows, 256));
    STEN1_DISPATCH(sten1_dgrad_kernel, grid)
  }
  return launch_status();
}
#undef STEN1_DISPATCH

// y[i] = x[i] * s[0] with the scalar read from device memory (chain rule through scalar losses without a host round trip)
CINEMA_API int cinema_mul_scalar_f32(const float* x, const float* s, float* y, long long n, void* stream) {
  if (!x || !s || !y || n <= 0) return CINEMA_ERR_BAD_ARG;
  CINEMA_LAUNCH(mul_scalar_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, s, y, n);
  return launch_status();
}
