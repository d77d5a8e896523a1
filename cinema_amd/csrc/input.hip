// GPU input pipeline of the pre-training step (reference cinema/mae/pretrain.py:157-200: monai RandZoomd -> ScaleIntensityd -> SpatialPadd(method="end")
// on the CPU workers).  fp32 single-channel images / volumes, HBM-bound elementwise kernels:
//   zoom_resample : monai Zoom(keep_size=True, padding_mode="constant") = torch interpolate(scale_factor=zoom, recompute_scale_factor=True,
//                   align_corners=False; trilinear for the SAX volume, bicubic (a = -0.75) for the LAX images) followed by a centred pad / crop back
//                   to the input size; also reduces min / max of the result (ScaleIntensity needs them).
//   scale_pad     : (x - min) / (max - min) (all zeros when max == min, monai rescale_array with minv = 0) written into the zero-padded batch slot.
#include "common.cuh"
#include "../../include/cinema_hip.h"

namespace {

// order-preserving float <-> uint map for atomicMin / atomicMax on floats of any sign
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

struct ZoomP {
  const float* src; float* dst; uint32_t* minmax_ord;
  int X, Y, Z;        // size (Z = 1 for 2-D)
  int ox, oy, oz;     // zoomed size floor(size * zoom)
  int mode;           // 0: (tri)linear, 1: bicubic over (x, y) (Z must be 1)
};

__device__ __forceinline__ float cubic1(float x) { const float A = -0.75f; return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }           // |x| <= 1
__device__ __forceinline__ float cubic2(float x) { const float A = -0.75f; return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }      // 1 < |x| < 2

// axis helper: destination index d (in the kept-size frame) -> index i in the zoomed frame, or -1 when d lies in the constant padding
__device__ __forceinline__ int to_zoomed(int d, int s, int o) { const int i = o < s ? d - (s - o) / 2 : d + (o - s) / 2; return (i >= 0 && i < o) ? i : -1; }

__global__ __launch_bounds__(256) void zoom_resample_kernel(ZoomP p) {
  __shared__ uint32_t smin, smax;
  if (threadIdx.x == 0) { smin = 0xffffffffu; smax = 0u; }
  __syncthreads();
  const long long n = (long long)p.X * p.Y * p.Z;
  float lo = INFINITY, hi = -INFINITY;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long long)gridDim.x * 256) {
    const int z = (int)(v % p.Z), y = (int)((v / p.Z) % p.Y), x = (int)(v / ((long long)p.Z * p.Y));
    const int ix = to_zoomed(x, p.X, p.ox), iy = to_zoomed(y, p.Y, p.oy), iz = to_zoomed(z, p.Z, p.oz);
    float val = 0.f;
    if (ix >= 0 && iy >= 0 && iz >= 0) {
      const float fx = (ix + 0.5f) * ((float)p.X / p.ox) - 0.5f, fy = (iy + 0.5f) * ((float)p.Y / p.oy) - 0.5f, fz = (iz + 0.5f) * ((float)p.Z / p.oz) - 0.5f;
      if (p.mode == 0) {
        const float cx = fmaxf(fx, 0.f), cy = fmaxf(fy, 0.f), cz = fmaxf(fz, 0.f);
        const int x0 = min((int)cx, p.X - 1), y0 = min((int)cy, p.Y - 1), z0 = min((int)cz, p.Z - 1);
        const int x1 = min(x0 + 1, p.X - 1), y1 = min(y0 + 1, p.Y - 1), z1 = min(z0 + 1, p.Z - 1);
        const float tx = cx - x0, ty = cy - y0, tz = cz - z0;
        auto at = [&](int a, int b, int c) { return p.src[((long long)a * p.Y + b) * p.Z + c]; };
        const float c00 = at(x0, y0, z0) * (1.f - tz) + at(x0, y0, z1) * tz, c01 = at(x0, y1, z0) * (1.f - tz) + at(x0, y1, z1) * tz;
        const float c10 = at(x1, y0, z0) * (1.f - tz) + at(x1, y0, z1) * tz, c11 = at(x1, y1, z0) * (1.f - tz) + at(x1, y1, z1) * tz;
        val = (c00 * (1.f - ty) + c01 * ty) * (1.f - tx) + (c10 * (1.f - ty) + c11 * ty) * tx;
      } else {
        const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
        const float tx = fx - x0, ty = fy - y0;
        const float wx[4] = {cubic2(tx + 1.f), cubic1(tx), cubic1(1.f - tx), cubic2(2.f - tx)};
        const float wy[4] = {cubic2(ty + 1.f), cubic1(ty), cubic1(1.f - ty), cubic2(2.f - ty)};
        for (int a = 0; a < 4; a++) {
          const int xa = min(max(x0 - 1 + a, 0), p.X - 1);
          float row = 0.f;
          for (int b = 0; b < 4; b++) row += wy[b] * p.src[(long long)xa * p.Y + min(max(y0 - 1 + b, 0), p.Y - 1)];
          val += wx[a] * row;
        }
      }
    }
    p.dst[v] = val;
    lo = fminf(lo, val); hi = fmaxf(hi, val);
  }
  lo = -wave_max(-lo); hi = wave_max(hi);
  if ((threadIdx.x & 63) == 0) { atomicMin(&smin, f2ord(lo)); atomicMax(&smax, f2ord(hi)); }
  __syncthreads();
  if (threadIdx.x == 0) { atomicMin(&p.minmax_ord[0], smin); atomicMax(&p.minmax_ord[1], smax); }
}

__global__ void minmax_init_kernel(uint32_t* mm) { mm[0] = 0xffffffffu; mm[1] = 0u; }

__global__ __launch_bounds__(256) void scale_pad_kernel(const float* src, int X, int Y, int Z, const uint32_t* minmax_ord, float* dst, int PX, int PY, int PZ) {
  const float mn = ord2f(minmax_ord[0]), mx = ord2f(minmax_ord[1]);
  const float inv = mx > mn ? 1.f / (mx - mn) : 0.f;
  const long long n = (long long)PX * PY * PZ;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long long)gridDim.x * 256) {
    const int z = (int)(v % PZ), y = (int)((v / PZ) % PY), x = (int)(v / ((long long)PZ * PY));
    float out = 0.f;
    if (x < X && y < Y && z < Z) out = mx > mn ? (src[((long long)x * Y + y) * Z + z] - mn) * inv : 0.f;
    dst[v] = out;
  }
}

}  // namespace

CINEMA_API int cinema_zoom_resample(const float* src, int X, int Y, int Z, float zoom_x, float zoom_y, float zoom_z, int mode, float* dst, unsigned int* minmax,
                                    void* stream) {
  if (!src || !dst || !minmax || X <= 0 || Y <= 0 || Z <= 0 || !(zoom_x > 0.f) || !(zoom_y > 0.f) || !(zoom_z > 0.f)) return CINEMA_ERR_BAD_ARG;
  if (mode != 0 && mode != 1) return CINEMA_ERR_BAD_ARG;
  if (mode == 1 && Z != 1) return CINEMA_ERR_UNSUPPORTED;
  ZoomP p{src, dst, minmax, X, Y, Z, (int)floorf(X * zoom_x), (int)floorf(Y * zoom_y), Z == 1 ? 1 : (int)floorf(Z * zoom_z), mode};
  if (p.ox < 1 || p.oy < 1 || p.oz < 1) return CINEMA_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  CINEMA_LAUNCH(minmax_init_kernel, dim3(1), dim3(1), 0, st, minmax);
  long long g = ((long long)X * Y * Z + 255) / 256;
  if (g > 2048) g = 2048;
  CINEMA_LAUNCH(zoom_resample_kernel, dim3((unsigned)g), dim3(256), 0, st, p);
  return launch_status();
}

CINEMA_API int cinema_scale_intensity_pad(const float* src, int X, int Y, int Z, const unsigned int* minmax, float* dst, int PX, int PY, int PZ, void* stream) {
  if (!src || !dst || !minmax || X <= 0 || Y <= 0 || Z <= 0 || PX <= 0 || PY <= 0 || PZ <= 0) return CINEMA_ERR_BAD_ARG;
  long long g = ((long long)PX * PY * PZ + 255) / 256;
  if (g > 2048) g = 2048;
  CINEMA_LAUNCH(scale_pad_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src, X, Y, Z, minmax, dst, PX, PY, PZ);
  return launch_status();
}
