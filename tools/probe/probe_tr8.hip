// Hardware-semantics probe for gfx950 (dev tooling): ds_read_b64_tr_b8 lane / byte mapping, and the e4m3 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, unit
// scales) fed by transpose reads of REDUCTION-STRIDED [k][m] byte tiles - the operand layout of an fp8 weight-gradient GEMM.
// Build: hipcc --offload-arch=gfx950 -O2 probe_tr8.hip -o probe_tr8 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(2))) int i2;
typedef __attribute__((ext_vector_type(8))) int i8;
typedef __attribute__((ext_vector_type(16))) float f16v;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

static float e4m3(unsigned char b) {  // OCP e4m3fn
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -v : v;
}

// (1) lane t reads 8 bytes at byte address addr[t]; lds[i] = i & 255 over a [32 rows][16 bytes] image = value (row << 4 | col) for rows < 16
__global__ void k_map(const int* addr, unsigned char* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned char)i;
  __syncthreads();
  i2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((i2 __attribute__((address_space(3)))*)(lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; j++) { out[threadIdx.x * 8 + j] = (v[0] >> (8 * j)) & 255; out[threadIdx.x * 8 + 4 + j] = (v[1] >> (8 * j)) & 255; }
}

// (2) C[i][j] = sum_k A[k][i] * B[k][j], A and B row-major [64 k][32] e4m3 bytes in LDS; fragments by four transpose reads per lane:
// 16-lane group q = l >> 4, t = l & 15: read h covers k rows 32 (q >> 1) + 8 h + (t >> 1), byte columns 16 (q & 1) + 8 (t & 1) .. + 7; the lane is assumed to
// receive column 16 (q & 1) + t = l & 31 at k = 32 (l >> 5) + 8 h + 0..7
__global__ void k_mfma(const unsigned char* A, const unsigned char* B, float* C) {
  __shared__ __attribute__((aligned(16))) unsigned char la[64 * 32], lb[64 * 32];
  const int l = threadIdx.x;
  for (int i = l; i < 64 * 32; i += 64) { la[i] = A[i]; lb[i] = B[i]; }
  __syncthreads();
  const int q = l >> 4, t = l & 15;
  i8 a, b;
  for (int h = 0; h < 4; h++) {
    const int row = 32 * (q >> 1) + 8 * h + (t >> 1), col = 16 * (q & 1) + 8 * (t & 1);
    const i2 va = __builtin_amdgcn_ds_read_tr8_b64_v2i32((i2 __attribute__((address_space(3)))*)(la + row * 32 + col));
    const i2 vb = __builtin_amdgcn_ds_read_tr8_b64_v2i32((i2 __attribute__((address_space(3)))*)(lb + row * 32 + col));
    a[2 * h] = va[0]; a[2 * h + 1] = va[1]; b[2 * h] = vb[0]; b[2 * h + 1] = vb[1];
  }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

int main() {
  int* d_addr; unsigned char* d_out;
  CK(hipMalloc(&d_addr, 64 * 4)); CK(hipMalloc(&d_out, 64 * 8));
  int addr[64];
  for (int l = 0; l < 64; l++) { const int q = l >> 4, t = l & 15; addr[l] = q * 256 + (t >> 1) * 16 + (t & 1) * 8; }  // group q: rows 16 q .. 16 q + 7 of a [.][16] image
  CK(hipMemcpy(d_addr, addr, sizeof(addr), hipMemcpyHostToDevice));
  k_map<<<1, 64>>>(d_addr, d_out);
  unsigned char out[512];
  CK(hipMemcpy(out, d_out, 512, hipMemcpyDeviceToHost));
  printf("ds_read_b64_tr_b8: lane t of a 16-lane group supplies row t>>1, bytes 8 (t&1)..; received (row, col) per byte:\n");
  int ok = 1;
  for (int l = 0; l < 64; l++) {
    const int q = l >> 4, t = l & 15;
    if (l < 18 || (l & 15) == 0) printf("  lane %2d:", l);
    for (int j = 0; j < 8; j++) {
      const int v = out[l * 8 + j], row = (v >> 4) & 15, col = v & 15;
      if (l < 18 || (l & 15) == 0) printf(" (%d,%d)", row, col);
      if (row != j || col != t || (q & 0) != 0) ok = 0;
    }
    if (l < 18 || (l & 15) == 0) printf("\n");
  }
  printf("assumed mapping (lane t <- column t, rows 0..7 of its group's 8 x 16 block): %s\n", ok ? "CONFIRMED" : "WRONG");

  std::vector<unsigned char> A(64 * 32), B(64 * 32);
  srand(7);
  for (auto& x : A) { x = rand() & 255; if ((x & 0x7f) == 0x7f) x = 0x3c; if (((x >> 3) & 15) > 9) x &= 0xcf; }
  for (auto& x : B) { x = rand() & 255; if ((x & 0x7f) == 0x7f) x = 0x3c; if (((x >> 3) & 15) > 9) x &= 0xcf; }
  unsigned char *dA, *dB; float* dC;
  CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dC, 32 * 32 * 4));
  CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
  k_mfma<<<1, 64>>>(dA, dB, dC);
  std::vector<float> C(1024);
  CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0;
  for (int i = 0; i < 32; i++)
    for (int j = 0; j < 32; j++) {
      double ref = 0;
      for (int k = 0; k < 64; k++) ref += (double)e4m3(A[k * 32 + i]) * e4m3(B[k * 32 + j]);
      worst = fmax(worst, fabs(ref - C[i * 32 + j])); scale = fmax(scale, fabs(ref));
    }
  printf("e4m3 MFMA 32x32x64 with transpose-read fragments of [k][m] tiles: max |err| %.3g of max |ref| %.3g -> %s\n", worst, scale, worst <= 1e-4 * scale ? "MATCH" : "MISMATCH");
  return 0;
}
