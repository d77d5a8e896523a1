// Hardware-semantics probe for gfx950 (dev tooling, not part of the product path).
// Dumps: (1) ds_read_b64_tr_b16 lane/element mapping, (2) MFMA bf16 operand layouts
// checked against a host reference with asymmetric operands, (3) device properties.
// Build: hipcc --offload-arch=gfx950 -O2 probe.hip -o probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(8))) short s8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4v;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }

// ---- (1) transpose read: lane t reads 8 bytes at byte address addr[t]; LDS holds lds[i] = i (u16)
__global__ void k_tr(const int* addr_elems, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + addr_elems[threadIdx.x]));
  for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = v[j];
}

// ---- (2a) MFMA 32x32x16 bf16. Assumed: A lane l holds A[i=l&31][k=8*(l>>5)+e]; B lane l holds B[k=8*(l>>5)+e][j=l&31]
//          C reg r of lane l = C[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]
__global__ void k_mfma32(const unsigned short* A, const unsigned short* B, float* C) {
  int l = threadIdx.x;
  s8 a, b;
  for (int e = 0; e < 8; e++) {
    a[e] = A[(l & 31) * 16 + 8 * (l >> 5) + e];
    b[e] = B[(8 * (l >> 5) + e) * 32 + (l & 31)];
  }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    C[row * 32 + (l & 31)] = c[r];
  }
}
// ---- (2b) MFMA 16x16x32 bf16. Assumed: A lane l holds A[i=l&15][k=8*(l>>4)+e]; B[k=8*(l>>4)+e][j=l&15]; C reg r: row=4*(l>>4)+r, col=l&15
__global__ void k_mfma16(const unsigned short* A, const unsigned short* B, float* C) {
  int l = threadIdx.x;
  s8 a, b;
  for (int e = 0; e < 8; e++) {
    a[e] = A[(l & 15) * 32 + 8 * (l >> 4) + e];
    b[e] = B[(8 * (l >> 4) + e) * 16 + (l & 15)];
  }
  f4v c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

// ---- (3) MFMA B operand fed by tr-read from a row-major [k][n] LDS tile (the wgrad / attention-V pattern).
// B tile: Bm[k=0..15][n=0..31] row-major in LDS (row stride 32 elems). For 32x32x16: lane l (j=l&31, g=l>>5) needs k=8g..8g+7.
// 16-lane group q=l>>4 covers cols 16*(q&1)..+15 and k-rows 8*(q>>1)+{0..7} with two tr reads (rows +0..3, +4..7).
// Per-lane address (t=l&15): row = 8*(q>>1) + 4*h + (t>>2), col = 16*(q&1) + 4*(t&3).
__global__ void k_mfma32_trB(const unsigned short* A, const unsigned short* B, float* C) {
  __shared__ __attribute__((aligned(16))) short lds[16 * 32];
  int l = threadIdx.x;
  for (int i = l; i < 16 * 32; i += 64) lds[i] = B[i];
  __syncthreads();
  s8 a, b;
  for (int e = 0; e < 8; e++) a[e] = A[(l & 31) * 16 + 8 * (l >> 5) + e];
  int q = l >> 4, t = l & 15;
  for (int h = 0; h < 2; h++) {
    int row = 8 * (q >> 1) + 4 * h + (t >> 2), col = 16 * (q & 1) + 4 * (t & 3);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + row * 32 + col));
    for (int j = 0; j < 4; j++) b[4 * h + j] = v[j];
  }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    C[row * 32 + (l & 31)] = c[r];
  }
}

// ---- (4) global_load_lds 16B: lane-linear destination check
__global__ void k_glds(const int* src, int* out) {
  __shared__ __attribute__((aligned(16))) int lds[64 * 4];
  int l = threadIdx.x;
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + l * 4),
                                   (void __attribute__((address_space(3)))*)lds, 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int j = 0; j < 4; j++) out[l * 4 + j] = lds[l * 4 + j];
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s arch=%s CUs=%d clock=%d MHz mem=%.1f GB l2=%d KB smemPerBlock=%zu regsPerBlock=%d warp=%d\n",
         p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000, p.totalGlobalMem / 1e9, p.l2CacheSize / 1024,
         p.sharedMemPerBlock, p.regsPerBlock, p.warpSize);

  // (1) tr read with linear addresses: lane t at element 4*t
  {
    std::vector<int> addr(64); for (int t = 0; t < 64; t++) addr[t] = 4 * t;
    int* d_addr; short* d_out; CK(hipMalloc(&d_addr, 256)); CK(hipMalloc(&d_out, 512));
    CK(hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice));
    k_tr<<<1, 64>>>(d_addr, d_out); CK(hipDeviceSynchronize());
    short out[256]; CK(hipMemcpy(out, d_out, 512, hipMemcpyDeviceToHost));
    printf("TR16_B64 linear (lane t addr = elem 4t): lane: e0 e1 e2 e3\n");
    for (int t = 0; t < 64; t++) printf("  lane %2d: %4d %4d %4d %4d\n", t, out[4 * t], out[4 * t + 1], out[4 * t + 2], out[4 * t + 3]);
    // hypothesis: within 16-lane group, result[t][j] = fetched[4*j + (t>>2)][t&3]
    int ok = 1;
    for (int l = 0; l < 64; l++) { int g = l >> 4, t = l & 15; for (int j = 0; j < 4; j++) { int src_lane = 16 * g + 4 * j + (t >> 2); int exp = 4 * src_lane + (t & 3); if (out[4 * l + j] != exp) ok = 0; } }
    printf("TR16 hypothesis H1 (res[t][j]=fetch[4j+(t>>2)][t&3]): %s\n", ok ? "PASS" : "FAIL");
    ok = 1;
    for (int l = 0; l < 64; l++) { int g = l >> 4, t = l & 15; for (int j = 0; j < 4; j++) { int src_lane = 16 * g + 4 * (t & 3) + j; int exp = 4 * src_lane + (t >> 2); if (out[4 * l + j] != exp) ok = 0; } }
    printf("TR16 hypothesis H2 (res[t][j]=fetch[4(t&3)+j][t>>2]): %s\n", ok ? "PASS" : "FAIL");
    // scattered addresses: lane t at element 64*t (to show that per-lane addresses are honoured)
    for (int t = 0; t < 64; t++) addr[t] = 64 * t + 8 * (t & 3);
    CK(hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice));
    k_tr<<<1, 64>>>(d_addr, d_out); CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, d_out, 512, hipMemcpyDeviceToHost));
    ok = 1;
    for (int l = 0; l < 64; l++) { int g = l >> 4, t = l & 15; for (int j = 0; j < 4; j++) { int src_lane = 16 * g + 4 * j + (t >> 2); int exp = addr[src_lane] + (t & 3); if (out[4 * l + j] != exp) ok = 0; } }
    printf("TR16 H1 with scattered per-lane addresses: %s\n", ok ? "PASS" : "FAIL");
    if (!ok) for (int t = 0; t < 64; t++) printf("  lane %2d (addr %4d): %4d %4d %4d %4d\n", t, addr[t], out[4 * t], out[4 * t + 1], out[4 * t + 2], out[4 * t + 3]);
  }
  // (2) MFMA layouts
  {
    std::vector<unsigned short> A(32 * 32), B(32 * 32);
    std::vector<float> Af(32 * 32), Bf(32 * 32);
    srand(1);
    for (int i = 0; i < 1024; i++) { float a = (float)((rand() % 17) - 8) / 4.f, b = (float)((rand() % 13) - 6) / 2.f; A[i] = f2bf(a); B[i] = f2bf(b); Af[i] = bf2f(A[i]); Bf[i] = bf2f(B[i]); }
    unsigned short *dA, *dB; float* dC; CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dC, 4096));
    CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
    float C[1024];
    // 32x32x16: A[32][16], B[16][32]
    k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C, dC, 4096, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { float r = 0; for (int k = 0; k < 16; k++) r += Af[i * 16 + k] * Bf[k * 32 + j]; err = fmax(err, fabs(r - C[i * 32 + j])); }
    printf("MFMA 32x32x16 bf16 layout check: max err %.3g -> %s\n", err, err < 1e-3 ? "PASS" : "FAIL");
    k_mfma32_trB<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C, dC, 4096, hipMemcpyDeviceToHost));
    err = 0; for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { float r = 0; for (int k = 0; k < 16; k++) r += Af[i * 16 + k] * Bf[k * 32 + j]; err = fmax(err, fabs(r - C[i * 32 + j])); }
    printf("MFMA 32x32x16 with tr-read B from row-major [k][n] LDS: max err %.3g -> %s\n", err, err < 1e-3 ? "PASS" : "FAIL");
    // 16x16x32: A[16][32], B[32][16]
    k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C, dC, 1024, hipMemcpyDeviceToHost));
    err = 0; for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { float r = 0; for (int k = 0; k < 32; k++) r += Af[i * 32 + k] * Bf[k * 16 + j]; err = fmax(err, fabs(r - C[i * 16 + j])); }
    printf("MFMA 16x16x32 bf16 layout check: max err %.3g -> %s\n", err, err < 1e-3 ? "PASS" : "FAIL");
  }
  // (4) glds
  {
    std::vector<int> src(256); for (int i = 0; i < 256; i++) src[i] = 1000 + i;
    int *dS, *dO; CK(hipMalloc(&dS, 1024)); CK(hipMalloc(&dO, 1024)); CK(hipMemcpy(dS, src.data(), 1024, hipMemcpyHostToDevice));
    k_glds<<<1, 64>>>(dS, dO); CK(hipDeviceSynchronize());
    int out[256]; CK(hipMemcpy(out, dO, 1024, hipMemcpyDeviceToHost));
    int ok = 1; for (int i = 0; i < 256; i++) if (out[i] != src[i]) ok = 0;
    printf("global_load_lds 16B lane-linear: %s\n", ok ? "PASS" : "FAIL");
  }
  return 0;
}
