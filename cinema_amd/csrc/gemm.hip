// bf16 GEMM with fused epilogues for gfx950: MFMA 32x32x16 tiles, register-staged double-buffered LDS,
// XOR-swizzled K-major tiles, ds_read_b64_tr_b16 transpose reads for reduction-strided operands
// (dgrad / wgrad), fp32-atomic split-K for the weight gradients.  See include/cinema_hip.h for the contract.
#include "gemm_shared.cuh"

namespace {


// 128x128x32 kernel, 3 stages of 16 KiB as a ring (2 as a double buffer: CINEMA_K32_STAGES=2), three workgroups per CU (see TileIO32).  gridDim.z = split-K slices as in the BK = 64 kernel;
// no split tail, no bias-gradient row sums; K % 32 == 0.
template <bool A_KMAJ, bool B_KMAJ, int EPI, int NST>
__device__ __forceinline__ void gemm_mfma_k32_body(const GemmP& p, char* smem) {
  using AIO = TileIO32<A_KMAJ>;
  using BIO = TileIO32<B_KMAJ>;
  constexpr int STAGE = AIO::BYTES + BIO::BYTES;  // 16 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int tiles_n = (p.n + BN - 1) / BN, tiles_m = (p.m + BM - 1) / BM;
  const int nkt = p.k / BK32;
  const int n_main = (int)gridDim.x;
  const int logical = xcd_remap(blockIdx.z * n_main + blockIdx.x, n_main * gridDim.z);
  const int zsplit = logical / n_main, tile = logical - zsplit * n_main;
  const int kt_begin = zsplit * p.ktiles_per_split * 2, kt_end = min(nkt, kt_begin + p.ktiles_per_split * 2);  // ktiles_per_split counts 64-wide tiles
  int tm, tn;
  tile_of(tile, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  if (kt_begin >= kt_end) return;

  float16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const typename AIO::Src2 asrc = AIO::src2(p.a, p.lda, m0, p.m, lane, wave_u);
  const typename BIO::Src2 bsrc = BIO::src2(p.b, p.ldb, n0, p.n, lane, wave_u);
  const size_t astep = AIO::k_step(p.lda), bstep = BIO::k_step(p.ldb);
  const uint32_t smem_addr = __builtin_amdgcn_readfirstlane(lds_address(smem));
  auto load_tile = [&](int stage, int kt) {  // 4 DMA pieces per wave: A blocks wave, wave+4; B blocks wave, wave+4
    const uint32_t sa = smem_addr + stage * STAGE + wave_u * 1024, sb = sa + AIO::BYTES;
    const size_t ka = (size_t)kt * astep, kb = (size_t)kt * bstep;
    glds16x4(sa, sa + 4096, sb, sb + 4096, asrc.p[0] + ka, asrc.p[1] + ka, bsrc.p[0] + kb, bsrc.p[1] + kb);
  };
  if constexpr (NST > 2) {
    // ring of NST stages (round 4): the LDS-DMA of k-tile kt + NST - 1 is issued at the top of k-tile kt - into the stage read in kt - 1, everyone has passed the
    // barrier that ended it - and waited for with a COUNTED vmcnt at the end of k-tile kt + NST - 2: two k-tiles of flight instead of less than one, nobody drains
    // its queue inside the loop.  With the drain at the end of every k-tile a CU has ~32 KB in flight on average, and 32 KB per ~1800 clocks of loaded latency is the
    // 18 B/clk per CU this kernel family stages (fill-rate probe).  Alone 10-20 % faster on the K = 512 shapes (decoder q / proj 31.0 -> 28.1 us, fc1
    // data gradient 101 -> 84 us, profiles/r04_ak_k32_ring.txt); in the step -0.2 ms on config 2, +-0.1 ms on configs 4 / 5.  Results identical to the 2-stage form.
    load_tile(0, kt_begin);
#pragma unroll
    for (int s = 1; s < NST - 1; s++)
      if (kt_begin + s < kt_end) load_tile(s, kt_begin + s);
    if (min(NST - 2, kt_end - 1 - kt_begin) >= 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0, nxt = NST - 1;
    for (int kt = kt_begin; kt < kt_end; kt++) {
      const char* sa = smem + cur * STAGE;
      const char* sb = sa + AIO::BYTES;
      if (kt + NST - 1 < kt_end) load_tile(nxt, kt + NST - 1);
#pragma unroll
      for (int ks = 0; ks < BK32 / 16; ks++) {
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
      // k-tile kt + 1 must have landed; kt + 2 (if issued) stays in flight
      if (kt + 2 < kt_end) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur = cur + 1 == NST ? 0 : cur + 1;
      nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
  } else {
  load_tile(0, kt_begin);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; kt++) {
    const int cur = (kt - kt_begin) & 1;
    const char* sa = smem + cur * STAGE;
    const char* sb = sa + AIO::BYTES;
#pragma unroll
    for (int ks = 0; ks < BK32 / 16; ks++) {
      short8v fa[2], fb[2];
      fa[0] = AIO::frag(sa, wm, ks, lane);
      fa[1] = AIO::frag(sa, wm + 32, ks, lane);
      fb[0] = BIO::frag(sb, wn, ks, lane);
      fb[1] = BIO::frag(sb, wn + 32, ks, lane);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      if (ks == 0 && kt + 1 < kt_end) load_tile(cur ^ 1, kt + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  }
  float* stg = reinterpret_cast<float*>(smem + wave * 8192);  // 8 KiB per wave: one 32x64 half at a time
  half_epilogue<EPI>(p, acc[0], m0 + wm, n0 + wn, lane, zsplit, stg);
  half_epilogue<EPI>(p, acc[1], m0 + wm + 32, n0 + wn, lane, zsplit, stg);
}
template <bool A_KMAJ, bool B_KMAJ, int EPI, int NST = 2>
__global__ __launch_bounds__(256, 3) void gemm_mfma_k32_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) char smem[NST * (TileIO32<A_KMAJ>::BYTES + TileIO32<B_KMAJ>::BYTES)];
  gemm_mfma_k32_body<A_KMAJ, B_KMAJ, EPI, NST>(p, smem);
}
// lanes form (common.cuh): blockIdx.y = lane, one parameter block per lane (identical shapes, different pointers)
template <bool A_KMAJ, bool B_KMAJ, int EPI, int NST = 2>
__global__ __launch_bounds__(256, 3) void gemm_mfma_k32_lanes_kernel(Lanes<GemmP> L) {
  __shared__ __attribute__((aligned(16))) char smem[NST * (TileIO32<A_KMAJ>::BYTES + TileIO32<B_KMAJ>::BYTES)];
  gemm_mfma_k32_body<A_KMAJ, B_KMAJ, EPI, NST>(L.p[blockIdx.y], smem);
}

// One 128x128 output tile (or k-slice of one) of problem p: everything after the work-item decoding of the kernels below.
// FP8: the operands are e4m3 bytes, k-major; the host describes them as bf16 matrices of HALF the reduction length (p.k, p.lda, p.ldb count byte
// pairs), so the tile loader moves the same 128-byte rows (= 128 fp8 values of k); only the fragment reads and the MFMA differ:
// v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (E8M0 = 127), twice the multiply-accumulates per cycle of the bf16 instruction.
typedef int int8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int8v frag_fp8(const char* lds, int base, int ks, int lane) {
  const int row = base + (lane & 31);
  const int c0 = ks * 4 + (lane >> 5) * 2;  // the lane's 32 consecutive k values = two 16-byte chunks of the 128-byte row
  const uint4 lo = *reinterpret_cast<const uint4*>(lds + swz_off<128>(row, c0)), hi = *reinterpret_cast<const uint4*>(lds + swz_off<128>(row, c0 + 1));
  int8v out;
  out[0] = lo.x; out[1] = lo.y; out[2] = lo.z; out[3] = lo.w; out[4] = hi.x; out[5] = hi.y; out[6] = hi.z; out[7] = hi.w;
  return out;
}

// Split tail finished INSIDE the launch, reduce-scatter form: every k-slice of a tail tile (1) publishes its fp32 accumulators fragment-ordered with
// write-through (sc1) 16-byte stores (each lane one store per accumulator quad: coalesced, no LDS staging), drains, and bumps the tile's publish counter;
// (2) waits until all n slices of the tile have published (they are the last workgroups of the grid and n_tiles x n <= the workgroup slots, so they are
// co-resident or waiting for slots that free themselves: no slice ever waits for a workgroup that cannot start); (3) sums ITS share of the tile - the
// 32x32 accumulator blocks u = slice, slice + n, ... of the 16, one block per wave at a time - over all slices in slice order (deterministic: the rounding
// does not depend on arrival order) with sc1 loads and runs the fused epilogue on it.  The serial part per workgroup is 64 KiB written + 64 KiB read
// whatever n is (the last-arriver form this replaces had ONE workgroup read up to 15 x 64 KiB; the fix-up launch costs a kernel boundary + 5 us).
// The slice that finishes last (second counter) zeroes both counters for the next launch.
template <int W>
__device__ __forceinline__ void block_epilogue_rows(const GemmP& p, const float16v& acc, int mw, int nw, int lane, float* stg) {
  constexpr int LPR = 32 / W, RPP = 64 / LPR, PASSES = 32 / RPP;  // W = 8: 4 lanes per row, 16 rows per pass; W = 4: 8 lanes per row, 8 rows per pass
  const int ml = lane & 31, hi = lane >> 5;
  const int rl = lane / LPR, cl = (lane % LPR) * W;
  const int n = nw + cl;
  float bv[W];
  float q8max = 0.f;
  const float q8inv = p.out8 ? *p.out8_inv : 0.f;
  EpiPre<W> pre[PASSES];
  epi_load_bias<W>(p, n, true, bv);
#pragma unroll
  for (int pss = 0; pss < PASSES; pss++) {
    const int m = mw + pss * RPP + rl;
    if (m < p.m && n < p.n) epi_load<W>(p, m, n, pre[pss]);
  }
#pragma unroll
  for (int q = 0; q < 4; q++)
    *reinterpret_cast<float4*>(stg + ml * 32 + (((2 * q + hi) ^ (ml & 7)) << 2)) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
#pragma unroll
  for (int pss = 0; pss < PASSES; pss++) {
    const int r = pss * RPP + rl;
    const int m = mw + r;
    float v[W];
#pragma unroll
    for (int c = 0; c < W / 4; c++) {
      const float4 t = *reinterpret_cast<const float4*>(stg + r * 32 + ((((cl >> 2) + c) ^ (r & 7)) << 2));
      v[4 * c] = t.x; v[4 * c + 1] = t.y; v[4 * c + 2] = t.z; v[4 * c + 3] = t.w;
    }
    if (m < p.m && n < p.n) epi_apply<W>(p, m, n, v, bv, pre[pss], q8max, q8inv);
  }
  if (p.out8_amax) q8_amax_commit(p.out8_amax, q8max, (int)blockIdx.x + (int)(threadIdx.x >> 6));
}
typedef unsigned int tail_u32x4 __attribute__((ext_vector_type(4)));
template <int EPI, int NI>
__device__ __forceinline__ void tail_finish_in_launch(const GemmP& p, const float16v (&acc)[NI][2], int tile, int slice, int m0, int n0, float* tail_dst, char* smem) {
  // NI = 2: the BK = 64 kernel (a wave holds 2x2 blocks of its 64x64 quadrant); the partial tile is 16 blocks x 4 quads x 64 lanes x 16 B
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t_local = tile - p.tail_begin, n_pieces = p.tail_split;
  unsigned* cnt = p.tail_cnt + 2 * t_local;
  {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(tail_dst, 0, BM * BN * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int idx = (wave_u * 16 + (i * 2 + j) * 4 + q) * 64 + lane;
          tail_u32x4 v = {__float_as_uint(acc[i][j][4 * q]), __float_as_uint(acc[i][j][4 * q + 1]), __float_as_uint(acc[i][j][4 * q + 2]),
                          __float_as_uint(acc[i][j][4 * q + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, idx * 16, 0, 16 /* sc1: write-through */);
        }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)n_pieces) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 26)) { p.tail_cnt[TAIL_ERROR_WORD] = 1u; break; }  // never in a healthy run (the host checks the word in tests / soak runs)
    }
  }
  __syncthreads();
  const float* tile_base = p.tail_ws + (size_t)t_local * n_pieces * (BM * BN);
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tile_base), 0, n_pieces * BM * BN * 4, 0x00020000);
  float* stg = reinterpret_cast<float*>(smem + wave_u * 4096);  // 4 KiB per wave (operand stages are free after the loop's last barrier)
  const GemmP q = epi_fold<EPI>(p);
  for (int u = slice + wave_u * n_pieces; u < 16; u += 4 * n_pieces) {
    float16v r;
#pragma unroll
    for (int e = 0; e < 16; e++) r[e] = 0.f;
    for (int s2 = 0; s2 < n_pieces; s2++) {
      tail_u32x4 t[4];
#pragma unroll
      for (int qq = 0; qq < 4; qq++) t[qq] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((u * 4 + qq) * 64 + lane) * 16, s2 * (BM * BN * 4), 16 /* sc1 */);
#pragma unroll
      for (int qq = 0; qq < 4; qq++) {
        r[4 * qq] += __uint_as_float(t[qq][0]); r[4 * qq + 1] += __uint_as_float(t[qq][1]);
        r[4 * qq + 2] += __uint_as_float(t[qq][2]); r[4 * qq + 3] += __uint_as_float(t[qq][3]);
      }
    }
    const int w_src = u >> 2, ij = u & 3;
    const int mw = m0 + (w_src >> 1) * 64 + (ij >> 1) * 32, nw = n0 + (w_src & 1) * 64 + (ij & 1) * 32;
    if (EPI == EPI_F32) block_epilogue_rows<4>(q, r, mw, nw, lane, stg);
    else block_epilogue_rows<8>(q, r, mw, nw, lane, stg);
  }
  __syncthreads();
  if (tid == 0) {
    const unsigned done = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (unsigned)(n_pieces - 1)) {  // every slice has passed its wait and its reads: leave both counters zero
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

enum { MODE_PLAIN = 0, MODE_FP8 = 1, MODE_CONV = 2, MODE_CONVW = 3 };
template <bool A_KMAJ, bool B_KMAJ, int EPI, int MODE = MODE_PLAIN>
__device__ __forceinline__ void gemm_tile(const GemmP& p, int tile, int zsplit, int kt_begin, int kt_end, float* tail_dst, char* smem) {
  constexpr bool FP8 = MODE == MODE_FP8, CONV = MODE == MODE_CONV, CONVW = MODE == MODE_CONVW;
  using AIO = TileIO<A_KMAJ>;
  using BIO = TileIO<B_KMAJ>;
  constexpr int STAGE = AIO::BYTES + BIO::BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int tiles_n = (p.n + BN - 1) / BN, tiles_m = (p.m + BM - 1) / BM;
  int tm, tn;
  tile_of(tile, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  if (kt_begin >= kt_end) return;

  float16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  float rs[2] = {0.f, 0.f};
  const bool do_rowsum = p.a_rowsum != nullptr && wn == 0 && n0 == 0;  // one wave column of the first n-tile covers every A row once
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bf16_t* zero_page = reinterpret_cast<const bf16_t*>(g_zero_page);
  const typename AIO::Src4 asrc = AIO::src4(p.a, p.lda, m0, p.m, lane, wave_u);
  const typename BIO::Src4 bsrc = BIO::src4(p.b, p.ldb, n0, p.n, lane, wave_u);
  const size_t astep = AIO::k_step(p.lda), bstep = BIO::k_step(p.ldb);
  const uint32_t smem_addr = __builtin_amdgcn_readfirstlane(lds_address(smem));
  // implicit convolution: the lane's four tile rows (one per DMA piece) as voxel coordinates, and the 16-byte chunk column it fills in every k-tile
  int cvx[4], cvy[4], cvz[4];
  long long cvrow[4];
  int conv_chunk = 0;
  int4 conv_ent = make_int4(0, 0, 0, 0);
  if constexpr (CONV) {
#pragma unroll
    for (int pss = 0; pss < 4; pss++) {
      const int row = (pss * 4 + wave_u) * 8 + (lane >> 3);
      int rg = m0 + row;
      rg = rg < p.m ? rg : p.m - 1;
      const int zg = p.cZ / p.cZB;  // rows along z (groups of cZB voxels)
      cvrow[pss] = (long long)rg * p.cZB;  // voxel row of the group's first voxel
      cvz[pss] = (rg % zg) * p.cZB;
      const int t = rg / zg;
      cvy[pss] = t % p.cY;
      cvx[pss] = (t / p.cY) % p.cX;
    }
    const int row0 = wave_u * 8 + (lane >> 3);
    conv_chunk = (lane & 7) ^ ((row0 >> 1) & 7);  // the same for the lane's four pieces (their rows differ by multiples of 32)
    conv_ent = p.conv_taps[kt_begin * 8 + conv_chunk];
  }
  // weight gradient of the implicit convolution: the lane's 16-byte column chunk (8 channels of one tap) is the same for its four DMA pieces and
  // for every k-tile; the rows (voxels) change with the k-tile and come with their coordinates from conv_coords
  int wdx = 0, wdy = 0, wdz = 0, wcol_ok = 0, wkr0 = 0;
  long long wdelta = 0;
  int wcoord[4] = {0, 0, 0, 0};
  if constexpr (CONVW) {
    wkr0 = wave_u * 4 + (lane >> 4);                      // k row of piece 0 inside the tile; piece pss adds 16 * pss
    const int chunk = (lane & 15) ^ ((wkr0 & 3) << 2);
    const int col = n0 + chunk * 8;
    const int ci = col % p.cC;
    const int4 e = p.conv_taps[min(col, p.n - 8) >> 3];
    wcol_ok = (col < p.n) && e.w;
    wdx = (e.y & 3) - 1; wdy = ((e.y >> 2) & 3) - 1; wdz = ((e.y >> 4) & 7) - 1;  // dz field: 3 bits (-1 .. cZB of a z-blocked table)
    wdelta = (long long)e.x * p.cC + ci;
#pragma unroll
    for (int pss = 0; pss < 4; pss++) {
      const int r = kt_begin * BK + wkr0 + 16 * pss;
      wcoord[pss] = r < p.k ? p.conv_coords[r] : -1;
    }
  }
  auto load_tile = [&](int stage, int kt) {  // operands of k-tile kt -> stage
    if constexpr (CONVW) {
      if ((kt + 1) * BK <= p.k) AIO::glds_at(smem_addr + stage * STAGE, asrc, (size_t)kt * astep, wave_u);
      else AIO::glds(smem + stage * STAGE, p.a, p.lda, m0, p.m, kt * BK, p.k, lane, wave_u, zero_page);
      const bf16_t* src[4];
#pragma unroll
      for (int pss = 0; pss < 4; pss++) {
        const int cd = wcoord[pss];
        const int x = cd & 1023, y = (cd >> 10) & 1023, z = (cd >> 20) & 1023;
        const bool ok = wcol_ok && cd >= 0 && (unsigned)(x + wdx) < (unsigned)p.cX && (unsigned)(y + wdy) < (unsigned)p.cY && (unsigned)(z + wdz) < (unsigned)p.cZ;
        const long long r = (long long)kt * BK + wkr0 + 16 * pss;
        src[pss] = ok ? p.b + (r * p.cZB * p.cC + wdelta) : zero_page;
      }
      const uint32_t b0 = smem_addr + stage * STAGE + AIO::BYTES + wave_u * 1024;
      glds16x4(b0, b0 + 4096, b0 + 8192, b0 + 12288, src[0], src[1], src[2], src[3]);
      if (kt + 1 < kt_end) {  // next tile's voxel coordinates: land under this tile's MFMAs
#pragma unroll
        for (int pss = 0; pss < 4; pss++) {
          const int r = (kt + 1) * BK + wkr0 + 16 * pss;
          wcoord[pss] = r < p.k ? p.conv_coords[r] : -1;
        }
      }
      return;
    }
    if constexpr (CONV) {
      const int4 e = conv_ent;
      const int dx = (e.y & 3) - 1, dy = ((e.y >> 2) & 3) - 1, dz = ((e.y >> 4) & 7) - 1;
      const bf16_t* src[4];
#pragma unroll
      for (int pss = 0; pss < 4; pss++) {
        const bool ok = e.w && (unsigned)(cvx[pss] + dx) < (unsigned)p.cX && (unsigned)(cvy[pss] + dy) < (unsigned)p.cY && (unsigned)(cvz[pss] + dz) < (unsigned)p.cZ;
        src[pss] = ok ? p.a + ((cvrow[pss] + e.x) * p.cC + e.z) : zero_page;
      }
      const uint32_t a0 = smem_addr + stage * STAGE + wave_u * 1024;
      glds16x4(a0, a0 + 4096, a0 + 8192, a0 + 12288, src[0], src[1], src[2], src[3]);
      if ((kt + 1) * BK <= p.k) BIO::glds_at(smem_addr + stage * STAGE + AIO::BYTES, bsrc, (size_t)kt * bstep, wave_u);
      else BIO::glds(smem + stage * STAGE + AIO::BYTES, p.b, p.ldb, n0, p.n, kt * BK, p.k, lane, wave_u, zero_page);
      if (kt + 1 < kt_end) conv_ent = p.conv_taps[(kt + 1) * 8 + conv_chunk];  // next tile's table entry: lands under this tile's MFMAs
      return;
    }
    if ((kt + 1) * BK <= p.k) {
      AIO::glds_at(smem_addr + stage * STAGE, asrc, (size_t)kt * astep, wave_u);
      BIO::glds_at(smem_addr + stage * STAGE + AIO::BYTES, bsrc, (size_t)kt * bstep, wave_u);
    } else {  // ragged last k-tile: per-element range tests, zero page
      AIO::glds(smem + stage * STAGE, p.a, p.lda, m0, p.m, kt * BK, p.k, lane, wave_u, zero_page);
      BIO::glds(smem + stage * STAGE + AIO::BYTES, p.b, p.ldb, n0, p.n, kt * BK, p.k, lane, wave_u, zero_page);
    }
  };
  load_tile(0, kt_begin);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = kt_begin; kt < kt_end; kt++) {
    const int cur = (kt - kt_begin) & 1;
    const char* sa = smem + cur * STAGE;
    const char* sb = sa + AIO::BYTES;
    const bool more = kt + 1 < kt_end;
    // The next k-tile's DMA is issued at the TOP of this k-tile (the other stage was last read before the barrier that ended the previous iteration): with two
    // stages and a drain at the end of every k-tile the flight time of a piece is what bounds this loop (fill-rate probe), and the ~150 clocks gained over
    // issuing it behind the first MFMA group (rounds 1-4) are worth 0.19 ms per config-2 step (four-round A/B, profiles/r04_aw_dma_first.txt); the issuing wave's
    // stall is covered by the other workgroup on the CU.
    if (more) load_tile(cur ^ 1, kt + 1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FP8) {
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        int8v fa[2], fb[2];
        fa[0] = frag_fp8(sa, wm, ks, lane);
        fa[1] = frag_fp8(sa, wm + 32, ks, lane);
        fb[0] = frag_fp8(sb, wn, ks, lane);
        fb[1] = frag_fp8(sb, wn + 32, ks, lane);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[j], fa[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    } else {
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
      if (!A_KMAJ && do_rowsum) { rs[0] += frag_sum8(fa[0]); rs[1] += frag_sum8(fa[1]); }  // VALU work in the shadow of the MFMAs
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile's DMA must have landed before anyone reads it
    __syncthreads();
  }

  if (!A_KMAJ && do_rowsum) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const float t = rs[i] + __shfl_xor(rs[i], 32, 64);
      const int m = m0 + wm + i * 32 + lane;
      if (lane < 32 && m < p.m) unsafeAtomicAdd(p.a_rowsum + m, t);
    }
  }
  // epilogue: both operand stages (64 KiB) are free after the last barrier -> 16 KiB of staging per wave
  float* stg = reinterpret_cast<float*>(smem + wave * 16384);
  if (EPI == EPI_GENERAL && p.accumulate && !p.ws) {  // atomic fallback (split-K without a workspace): register epilogue
    const bool add_bias = zsplit == 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int m = m0 + wm + i * 32 + (lane & 31);
      if (m >= p.m) continue;
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int n = n0 + wn + j * 32 + 8 * q + 4 * (lane >> 5);
          if (n < p.n) epilogue4(p, m, n, acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3], add_bias);
        }
    }
    return;
  }
  if constexpr (FP8) {  // dequantisation: alpha x scale_a x scale_b (per-tensor scales from device memory)
    GemmP q = p;
    q.alpha = p.alpha * p.scale_b[0] * (p.scale_a_rows ? 1.0f : p.scale_a[0]);
    tile_epilogue<EPI>(q, acc, m0 + wm, n0 + wn, lane, zsplit, stg);
    return;
  }
  // fp32 partial of a split-tail k-slice: plain [128][128] rows (address = base + m * 128 + n with the tile origin folded into base)
  if (EPI != EPI_GENERAL && tail_dst && p.tail_cnt) {  // (the general class keeps the fix-up launch: its epilogue has no registers to spare)
    tail_finish_in_launch<EPI>(p, acc, tile, zsplit, m0, n0, tail_dst, smem);
    return;
  }
  if (tail_dst) tile_epilogue<EPI>(p, acc, m0 + wm, n0 + wn, lane, 0, stg, tail_dst - ((long long)m0 * BN + n0), BN);
  else tile_epilogue<EPI>(p, acc, m0 + wm, n0 + wn, lane, zsplit, stg);
}

template <bool A_KMAJ, bool B_KMAJ, int EPI, int MODE = MODE_PLAIN>
__device__ __forceinline__ void gemm_mfma_body(const GemmP& p, char* smem) {
  // linear dispatch id (x fastest, then z) -> logical work item, split-major so that one XCD sees one k-range
  const int nkt = (p.k + BK - 1) / BK;
  int zsplit = 0, tile, kt_begin, kt_end;
  float* tail_dst = nullptr;
  if (p.tail_split > 0 && (int)blockIdx.x >= p.tail_begin) {
    // split tail: dispatched last (highest ids), dealt round-robin over the XCDs as they drain
    const int j = (int)blockIdx.x - p.tail_begin;
    tile = p.tail_begin + j / p.tail_split;
    const int slice = j - (tile - p.tail_begin) * p.tail_split;
    kt_begin = slice * p.tail_ktiles;
    kt_end = min(nkt, kt_begin + p.tail_ktiles);
    tail_dst = p.tail_ws + (size_t)j * (BM * BN);
    zsplit = slice;  // bias-gradient row sums / nothing else depends on it in this mode
  } else {
    const int n_main = p.tail_split > 0 ? p.tail_begin : (int)gridDim.x;
    const int logical = xcd_remap(blockIdx.z * n_main + blockIdx.x, n_main * gridDim.z);
    zsplit = logical / n_main;
    tile = logical - zsplit * n_main;
    kt_begin = zsplit * p.ktiles_per_split;
    kt_end = min(nkt, kt_begin + p.ktiles_per_split);
  }
  gemm_tile<A_KMAJ, B_KMAJ, EPI, MODE>(p, tile, zsplit, kt_begin, kt_end, tail_dst, smem);
}
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * (TileIO<true>::BYTES + TileIO<true>::BYTES)];
  gemm_mfma_body<true, true, EPI, MODE_FP8>(p, smem);
}
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_conv_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * (TileIO<true>::BYTES + TileIO<true>::BYTES)];
  gemm_mfma_body<true, true, EPI, MODE_CONV>(p, smem);
}
__global__ __launch_bounds__(256, 2) void gemm_convw_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * (TileIO<false>::BYTES + TileIO<false>::BYTES)];
  gemm_mfma_body<false, false, EPI_F32, MODE_CONVW>(p, smem);
}
template <bool A_KMAJ, bool B_KMAJ, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * (TileIO<A_KMAJ>::BYTES + TileIO<B_KMAJ>::BYTES)];
  gemm_mfma_body<A_KMAJ, B_KMAJ, EPI>(p, smem);
}
template <bool A_KMAJ, bool B_KMAJ, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_mfma_lanes_kernel(Lanes<GemmP> L) {
  __shared__ __attribute__((aligned(16))) char smem[2 * (TileIO<A_KMAJ>::BYTES + TileIO<B_KMAJ>::BYTES)];
  gemm_mfma_body<A_KMAJ, B_KMAJ, EPI>(L.p[blockIdx.y], smem);
}

// GROUPED launch: the tiles of up to 8 independent problems in one grid (no split-K).  The four weight gradients of a transformer block
// have 36-144 tiles each - far fewer than the 512 workgroup slots - so each used to be cut into 3-14 k-slices with fp32 slabs and a reduce
// launch; together they are 432 tiles, which fill the slots with whole-K tiles: no slabs, no reduce, one launch instead of eight.
struct GroupP {
  GemmP p[8];
  int tile_begin[9];  // prefix sums of the tile counts
  int count;
};
template <bool A_KMAJ, bool B_KMAJ, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_mfma_grouped_kernel(GroupP g) {
  using AIO = TileIO<A_KMAJ>;
  using BIO = TileIO<B_KMAJ>;
  constexpr int STAGE = AIO::BYTES + BIO::BYTES;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int logical = xcd_remap(blockIdx.x, gridDim.x);  // one XCD works on a contiguous tile range, i.e. mostly on one problem
  int i = 0;
#pragma unroll
  for (int j = 1; j < 8; j++) i += (j < g.count && logical >= g.tile_begin[j]) ? 1 : 0;
  const GemmP& p = g.p[i];
  gemm_tile<A_KMAJ, B_KMAJ, EPI>(p, logical - g.tile_begin[i], 0, 0, (p.k + BK - 1) / BK, nullptr, smem);
}

// Sums the k-slices of the split-tail tiles and applies the fused epilogue: thread = 8 consecutive columns of one row.
struct TailFixP { GemmP p; int tiles_m, tiles_n; };
__device__ __forceinline__ void tail_fixup_body(const GemmP& p, int tiles_m, int tiles_n) {
  const int r = blockIdx.x >> 3;
  const int idx = (blockIdx.x & 7) * 256 + threadIdx.x;
  const int row = idx >> 4, c8 = (idx & 15) * 8;
  int tm, tn;
  tile_of(p.tail_begin + r, tiles_m, tiles_n, tm, tn);
  const int m = tm * BM + row, n = tn * BN + c8;
  if (m >= p.m || n >= p.n) return;
  const float* src = p.tail_ws + (size_t)r * p.tail_split * (BM * BN) + row * BN + c8;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.tail_split; z++) {
    const float4 a = *reinterpret_cast<const float4*>(src + (size_t)z * (BM * BN)), b = *reinterpret_cast<const float4*>(src + (size_t)z * (BM * BN) + 4);
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  }
  if (p.out_f32) {
    float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
    epilogue_row<4>(p, m, n, lo, true);
    epilogue_row<4>(p, m, n + 4, hi, true);
  } else {
    float bv[8], q8max = 0.f;
    EpiPre<8> e;
    epi_load_bias<8>(p, n, true, bv);
    epi_load<8>(p, m, n, e);
    epi_apply<8>(p, m, n, v, bv, e, q8max);
    if (p.out8_amax && q8max > 0.f) {  // a few thousand threads per launch: per-lane, look before raising
      unsigned int* s8 = p.out8_amax + ((blockIdx.x * 256 + threadIdx.x) & (Q8_SLOTS - 1));
      if (__hip_atomic_load(s8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < __float_as_uint(q8max)) atomicMax(s8, __float_as_uint(q8max));
    }
  }
}
__global__ __launch_bounds__(256) void tail_fixup_kernel(TailFixP t) { tail_fixup_body(t.p, t.tiles_m, t.tiles_n); }
__global__ __launch_bounds__(256) void tail_fixup_lanes_kernel(Lanes<TailFixP> L) { const TailFixP& t = L.p[blockIdx.y]; tail_fixup_body(t.p, t.tiles_m, t.tiles_n); }

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

// dst[m][n] (+)= alpha * sum_z ws[z][m][n]   (second pass of the workspace split-K; fully coalesced, deterministic)
struct SplitkP { const float* ws; int splits, m, n; float* dst; int ldd, accumulate; float alpha; };
__device__ __forceinline__ void splitk_reduce_body(const float* ws, int splits, int m, int n, float* dst, int ldd, int accumulate, float alpha) {
  // grid.y slices the split range (tiny outputs come with hundreds of splits: a single thread summing them serially was
  // latency-bound); slices > 1 combine with fp32 atomics (only ever used with accumulate=1), one slice does a plain RMW
  const long long total4 = (long long)m * n / 4;
  const size_t slab = (size_t)m * n;
  const int per = (splits + gridDim.y - 1) / gridDim.y;
  const int z0 = blockIdx.y * per, z1 = min(splits, z0 + per);
  if (z0 >= z1) return;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const size_t e = (size_t)i * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = z0;
    for (; z + 3 < z1; z += 4) {  // 4 independent loads in flight
      const float4 a = *reinterpret_cast<const float4*>(ws + (size_t)z * slab + e), b = *reinterpret_cast<const float4*>(ws + (size_t)(z + 1) * slab + e);
      const float4 c = *reinterpret_cast<const float4*>(ws + (size_t)(z + 2) * slab + e), d4 = *reinterpret_cast<const float4*>(ws + (size_t)(z + 3) * slab + e);
      s.x += (a.x + b.x) + (c.x + d4.x); s.y += (a.y + b.y) + (c.y + d4.y); s.z += (a.z + b.z) + (c.z + d4.z); s.w += (a.w + b.w) + (c.w + d4.w);
    }
    for (; z < z1; z++) {
      const float4 v = *reinterpret_cast<const float4*>(ws + (size_t)z * slab + e);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int row = (int)(e / n), col = (int)(e % n);
    float* dp = dst + (size_t)row * ldd + col;
    if (gridDim.y > 1) {
      unsafeAtomicAdd(dp, alpha * s.x); unsafeAtomicAdd(dp + 1, alpha * s.y); unsafeAtomicAdd(dp + 2, alpha * s.z); unsafeAtomicAdd(dp + 3, alpha * s.w);
    } else {
      float4* d = reinterpret_cast<float4*>(dp);
      if (accumulate) { const float4 o = *d; s.x = o.x + alpha * s.x; s.y = o.y + alpha * s.y; s.z = o.z + alpha * s.z; s.w = o.w + alpha * s.w; }
      else { s.x *= alpha; s.y *= alpha; s.z *= alpha; s.w *= alpha; }
      *d = s;
    }
  }
}
__global__ __launch_bounds__(256) void splitk_reduce_kernel(SplitkP q) { splitk_reduce_body(q.ws, q.splits, q.m, q.n, q.dst, q.ldd, q.accumulate, q.alpha); }
__global__ __launch_bounds__(256) void splitk_reduce_lanes_kernel(Lanes<SplitkP> L) {
  const SplitkP& q = L.p[blockIdx.z];
  splitk_reduce_body(q.ws, q.splits, q.m, q.n, q.dst, q.ldd, q.accumulate, q.alpha);
}

// Streaming column sum for dense bf16 matrices (bias gradients): 16-byte loads, 8 columns per lane, 32 lanes x 8 row-lanes
// per block, grid sized to keep >= ~1k blocks in flight; HBM-bound (each element is read exactly once).
__global__ __launch_bounds__(256) void colsum_bf16_vec_kernel(const bf16_t* x, int m, int n, int ldx, float* out, int rows_per_block) {
  __shared__ float part[8][32][9];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = blockIdx.x * 256 + tx * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(m, r0 + rows_per_block);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < n) {
    int r = r0 + ty;
    for (; r + 24 < r1; r += 32) {  // 4 independent 16-byte loads in flight per lane
      uint4 u[4];
#pragma unroll
      for (int i = 0; i < 4; i++) u[i] = *reinterpret_cast<const uint4*>(x + (size_t)(r + 8 * i) * ldx + col);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        acc[0] += bf2f((bf16_t)(u[i].x & 0xffff)); acc[1] += bf2f((bf16_t)(u[i].x >> 16));
        acc[2] += bf2f((bf16_t)(u[i].y & 0xffff)); acc[3] += bf2f((bf16_t)(u[i].y >> 16));
        acc[4] += bf2f((bf16_t)(u[i].z & 0xffff)); acc[5] += bf2f((bf16_t)(u[i].z >> 16));
        acc[6] += bf2f((bf16_t)(u[i].w & 0xffff)); acc[7] += bf2f((bf16_t)(u[i].w >> 16));
      }
    }
    for (; r < r1; r += 8) {
      const uint4 u = *reinterpret_cast<const uint4*>(x + (size_t)r * ldx + col);
      acc[0] += bf2f((bf16_t)(u.x & 0xffff)); acc[1] += bf2f((bf16_t)(u.x >> 16));
      acc[2] += bf2f((bf16_t)(u.y & 0xffff)); acc[3] += bf2f((bf16_t)(u.y >> 16));
      acc[4] += bf2f((bf16_t)(u.z & 0xffff)); acc[5] += bf2f((bf16_t)(u.z >> 16));
      acc[6] += bf2f((bf16_t)(u.w & 0xffff)); acc[7] += bf2f((bf16_t)(u.w >> 16));
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) part[ty][tx][i] = acc[i];
  __syncthreads();
  // 256 threads <-> 256 columns of the block
  const int c = threadIdx.x;
  if (blockIdx.x * 256 + c < n) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) s += part[j][c >> 3][c & 7];
    unsafeAtomicAdd(out + blockIdx.x * 256 + c, s);
  }
}

}  // namespace

CINEMA_API int cinema_gemm_bf16(cinema_gemm_args* a, void* stream) {
  if (!a || !a->a || !a->b || a->m <= 0 || a->n <= 0 || a->k <= 0) return CINEMA_ERR_BAD_ARG;
  if (!a->d && (!a->out8 || a->out_f32 || a->accumulate || a->split_k > 1 || a->force_generic)) return CINEMA_ERR_BAD_ARG;  // D may be omitted only when its 8-bit copy is the output
  if (a->accumulate && !a->out_f32) return CINEMA_ERR_BAD_ARG;
  const int split = a->split_k < 1 ? 1 : a->split_k;
  if (split > 1 && !a->accumulate && !a->workspace) return CINEMA_ERR_BAD_ARG;
  GemmP p;
  p.a = (const bf16_t*)a->a; p.b = (const bf16_t*)a->b; p.d = a->d;
  p.m = a->m; p.n = a->n; p.k = a->k; p.lda = a->lda; p.ldb = a->ldb; p.ldd = a->ldd;
  p.alpha = a->alpha;
  p.bias = a->bias; p.res_f32 = a->residual_f32; p.res_bf16 = a->residual_bf16; p.ld_res = a->ld_res;
  p.gelu_in = a->gelu_in; p.ld_gelu = a->ld_gelu; p.row_mask = a->row_mask; p.aux_out = a->aux_out; p.ld_aux = a->ld_aux;
  p.act = a->act; p.gelu_deriv = a->gelu_deriv; p.out_f32 = a->out_f32; p.accumulate = a->accumulate; p.ws = nullptr; p.a_rowsum = nullptr;
  p.scale_a = nullptr; p.scale_b = nullptr; p.scale_a_rows = 0; p.conv_taps = nullptr; p.conv_coords = nullptr; p.cZB = 1;
  if (a->out8_amax) {  // 8-bit output copy / maximum: bf16 outputs only, 8-byte rows
    if (a->out_f32 || (a->out8 && (!a->out8_inv_scale || (a->ld_out8 & 7) || (((uintptr_t)a->out8) & 7)))) return CINEMA_ERR_BAD_ARG;
    p.out8 = a->out8; p.ld_out8 = a->ld_out8; p.out8_inv = a->out8_inv_scale; p.out8_amax = a->out8_amax;
  } else if (a->out8) return CINEMA_ERR_BAD_ARG;
  if (a->colsum_partials) {
    if (a->out_f32 || (((uintptr_t)a->colsum_partials) & 15) || (a->n & 7)) return CINEMA_ERR_BAD_ARG;
    p.colsum_partials = a->colsum_partials;
  }
  hipStream_t st = (hipStream_t)stream;

  auto al8 = [](int v) { return (v & 7) == 0; };
  auto ptr16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  bool fast = a->force_generic != 1 && al8(a->lda) && al8(a->ldb) && al8(a->ldd) && al8(a->n) && ptr16(a->a) && ptr16(a->b) && ptr16(a->d);
  fast = fast && (a->a_kmajor ? al8(a->k) : al8(a->m)) && (a->b_kmajor ? al8(a->k) : true);
  fast = fast && (!a->bias || ptr16(a->bias)) && (!a->residual_f32 || (al8(a->ld_res) && ptr16(a->residual_f32)));
  fast = fast && (!a->residual_bf16 || (al8(a->ld_res) && ptr16(a->residual_bf16))) && (!a->gelu_in || (al8(a->ld_gelu) && ptr16(a->gelu_in)));
  fast = fast && (!a->aux_out || (al8(a->ld_aux) && ptr16(a->aux_out)));
  fast = fast && !(a->a_kmajor == 0 && a->b_kmajor == 1);  // (M-major A, K-major B) is not used by the path
  if (!fast && (p.out8_amax || p.colsum_partials || !a->d)) return CINEMA_ERR_UNSUPPORTED;  // the 8-bit copy / strip sums are written by the staged MFMA epilogues only
  if (fast) {
    const int nkt = (a->k + BK - 1) / BK;
    const int sp = split > nkt ? nkt : split;
    p.ktiles_per_split = (nkt + sp - 1) / sp;
    const int gz = (nkt + p.ktiles_per_split - 1) / p.ktiles_per_split;
    dim3 grid(((a->m + BM - 1) / BM) * ((a->n + BN - 1) / BN), 1, gz);
    p.a_rowsum = (!a->a_kmajor) ? a->a_rowsum : nullptr;
    if (a->a_rowsum && a->a_kmajor) return CINEMA_ERR_UNSUPPORTED;
    const bool two_pass = gz > 1 && a->out_f32 && a->workspace && a->workspace_bytes >= (long long)gz * a->m * a->n * 4 && !(((uintptr_t)a->workspace) & 15) &&
                          !a->bias && !a->residual_f32 && !a->residual_bf16 && !a->gelu_in && !a->row_mask && !a->aux_out && a->act == 0;
    p.ws = two_pass ? (float*)a->workspace : nullptr;
    if (gz > 1 && !two_pass && !a->accumulate) return CINEMA_ERR_BAD_ARG;
    if (gz == 1 && a->accumulate && !a->residual_f32 && !a->residual_bf16) {  // one owner per element: plain read-modify-write, no atomics
      p.res_f32 = (const float*)a->d; p.ld_res = a->ldd; p.accumulate = 0;
    }
    a->kernel_used = (a->a_kmajor && a->b_kmajor) ? 1 : (a->a_kmajor ? 2 : 3);
    // split tail (measured: 516 tiles on 512 slots cost 1.64 rounds, the 4 left-over tiles run alone at the end): worth a second
    // launch only when the reduction has >= 12 k-tiles and the left-over tiles can be cut at least in two
    p.tail_split = 0; p.tail_begin = 0; p.tail_ktiles = 0; p.tail_ws = nullptr; p.tail_cnt = nullptr;
    bool tail = false;
    // thresholds: with the fix-up launch a tail pays from 12 k-tiles (K >= 768) and >= 4 k-tiles per slice; finished in the launch (no kernel boundary, the
    // finish spread over the slices) shorter reductions and slices pay too
    constexpr int min_nkt = 12, min_kt = 4;
    if (gz == 1 && a->force_generic == 0 && !p.accumulate && !p.colsum_partials && a->workspace && !(((uintptr_t)a->workspace) & 15) && nkt >= min_nkt) {  // (strip sums: whole tiles only)  // K >= 768 (tools/tail_ab.py: 10960x768x768 38 -> 34 us, x1024 42 -> 35 us; at K = 512 the fix-up launch costs what it saves)
      static int slots = 0;
      if (slots == 0) {
        int dev = 0; hipDeviceProp_t prop;
        slots = 2 * ((hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256);
      }
      const int tiles = (int)grid.x, rem = tiles % slots;
      if (rem > 0 && rem <= slots / 2) {
        int sp2 = slots / rem;
        if (sp2 > nkt / min_kt) sp2 = nkt / min_kt;  // >= min_kt k-tiles per slice
        if (sp2 > 16) sp2 = 16;  // = the 32x32 accumulator blocks of a tile: every slice finishes at least one of them
        const int kts = (nkt + sp2 - 1) / sp2;
        sp2 = (nkt + kts - 1) / kts;
        if (sp2 >= 2 && a->workspace_bytes >= (long long)rem * sp2 * BM * BN * 4) {
          p.tail_begin = tiles - rem; p.tail_split = sp2; p.tail_ktiles = kts; p.tail_ws = (float*)a->workspace;
          if (a->tail_counters && !(((uintptr_t)a->tail_counters) & 15) && 2 * rem <= TAIL_ERROR_WORD) p.tail_cnt = (unsigned*)a->tail_counters;
          grid.x = p.tail_begin + rem * sp2;
          tail = true;
        }
      }
    }
    {
      // epilogue class (compile-time specialisation, see EPI_*): what the argument combination needs, GENERAL for the rest
      int epi = EPI_GENERAL;
      const bool common = !p.row_mask && !p.res_bf16 && !(p.accumulate && !p.ws);
      if (p.ws) epi = EPI_F32;  // split-K slabs: only the plain partial-tile path of the epilogue runs
      else if (common && !p.out_f32 && !p.res_f32 && !p.gelu_in && p.act == 0 && !p.aux_out) epi = EPI_BF16;
      else if (common && !p.out_f32 && !p.res_f32 && !p.gelu_in && p.act == 1) epi = EPI_BF16_GELU;
      else if (common && !p.out_f32 && !p.res_f32 && p.gelu_in && p.act == 0 && !p.aux_out && !p.bias) epi = EPI_BF16_GELU_GRAD;
      else if (common && p.out_f32 && !p.gelu_in && p.act == 0 && !p.aux_out) epi = EPI_F32;
      a->kernel_used += 8 * epi;  // 1..3 = operand layout, + 8 x epilogue class
      if (epi == EPI_GENERAL) p.tail_cnt = nullptr;
      // short reductions go to the BK = 32 kernel (3-4 workgroups per CU): in the step 32.47 vs 32.79 ms with the threshold at 512, 32.52 at 768
      // (above that the BK = 64 loop and its split tail win).  Three LDS stages: a ring with counted waits
      constexpr int k32_env = 512;
      const bool k32 = epi != EPI_GENERAL && !tail && !p.a_rowsum && !(p.accumulate && !p.ws) && (a->k % 32) == 0 && a->k <= k32_env && a->force_generic == 0;
      if (k32) {
        a->kernel_used += 128;  // the BK = 32 instance of the same layout / epilogue class
#define LAUNCH_K32_N(E, N)                                                                                                          \
  do {                                                                                                                         \
    if (a->a_kmajor && a->b_kmajor) launch_lanes(gemm_mfma_k32_kernel<true, true, E, N>, gemm_mfma_k32_lanes_kernel<true, true, E, N>, 1, grid, dim3(256), 0, st, p);        \
    else if (a->a_kmajor && !a->b_kmajor) launch_lanes(gemm_mfma_k32_kernel<true, false, E, N>, gemm_mfma_k32_lanes_kernel<true, false, E, N>, 1, grid, dim3(256), 0, st, p); \
    else launch_lanes(gemm_mfma_k32_kernel<false, false, E, N>, gemm_mfma_k32_lanes_kernel<false, false, E, N>, 1, grid, dim3(256), 0, st, p);                                 \
  } while (0)
#define LAUNCH_K32(E) LAUNCH_K32_N(E, 3)
        switch (epi) {
          case EPI_BF16: LAUNCH_K32(EPI_BF16); break;
          case EPI_BF16_GELU: LAUNCH_K32(EPI_BF16_GELU); break;
          case EPI_BF16_GELU_GRAD: LAUNCH_K32(EPI_BF16_GELU_GRAD); break;
          default: LAUNCH_K32(EPI_F32); break;
        }
#undef LAUNCH_K32
#undef LAUNCH_K32_N
      } else {
#define LAUNCH_LAYOUT(E)                                                                                                    \
  do {                                                                                                                      \
    if (a->a_kmajor && a->b_kmajor) launch_lanes(gemm_mfma_kernel<true, true, E>, gemm_mfma_lanes_kernel<true, true, E>, 1, grid, dim3(256), 0, st, p);        \
    else if (a->a_kmajor && !a->b_kmajor) launch_lanes(gemm_mfma_kernel<true, false, E>, gemm_mfma_lanes_kernel<true, false, E>, 1, grid, dim3(256), 0, st, p); \
    else launch_lanes(gemm_mfma_kernel<false, false, E>, gemm_mfma_lanes_kernel<false, false, E>, 1, grid, dim3(256), 0, st, p);                                 \
  } while (0)
      switch (epi) {
        case EPI_BF16: LAUNCH_LAYOUT(EPI_BF16); break;
        case EPI_BF16_GELU: LAUNCH_LAYOUT(EPI_BF16_GELU); break;
        case EPI_BF16_GELU_GRAD: LAUNCH_LAYOUT(EPI_BF16_GELU_GRAD); break;
        case EPI_F32: LAUNCH_LAYOUT(EPI_F32); break;
        default: LAUNCH_LAYOUT(EPI_GENERAL); break;
      }
#undef LAUNCH_LAYOUT
      }
    }
    if (tail && !p.tail_cnt) {
      const int rem = ((int)grid.x - p.tail_begin) / p.tail_split;
      const TailFixP tf{p, (a->m + BM - 1) / BM, (a->n + BN - 1) / BN};
      launch_lanes(tail_fixup_kernel, tail_fixup_lanes_kernel, 1, dim3(rem * 8), dim3(256), 0, st, tf);
    }
    if (two_pass) {
      long long blocks = ((long long)a->m * a->n / 4 + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      int slices = 1;
      if (a->accumulate && blocks < 256) {  // few output elements, many splits: parallelise over the split range too
        slices = (int)(512 / blocks);
        if (slices > (gz + 7) / 8) slices = (gz + 7) / 8;
        if (slices < 1) slices = 1;
      }
      const SplitkP sk{(const float*)a->workspace, gz, a->m, a->n, (float*)a->d, a->ldd, a->accumulate, a->alpha};
      launch_lanes(splitk_reduce_kernel, splitk_reduce_lanes_kernel, 2, dim3((unsigned)blocks, slices), dim3(256), 0, st, sk);
    }
    return launch_status();
  }
  const int nkt = (a->k + 15) / 16;
  const int sp = split > nkt ? nkt : split;
  p.ktiles_per_split = (nkt + sp - 1) / sp;
  const int gz = (nkt + p.ktiles_per_split - 1) / p.ktiles_per_split;
  dim3 grid(((a->m + 63) / 64) * ((a->n + 63) / 64), 1, gz);
  if (a->a_rowsum) {  // the generic kernel does not fuse the row sums: one column-sum launch over the stored [K][M] operand
    if (a->a_kmajor) return CINEMA_ERR_UNSUPPORTED;
    int chunks = (a->k + 511) / 512;
    if (chunks > 256) chunks = 256;
    const int rpb = (a->k + chunks - 1) / chunks;
    CINEMA_LAUNCH(colsum_kernel, dim3((a->m + 63) / 64, (a->k + rpb - 1) / rpb), dim3(256), 0, st, a->a, 0, (const int*)nullptr, a->k, a->m, a->lda,
                       a->a_rowsum, rpb);
  }
  const int a_rs = a->a_kmajor ? a->lda : 1, a_cs = a->a_kmajor ? 1 : a->lda;   // element (m,k) = a[m*a_rs + k*a_cs]
  const int b_rs = a->b_kmajor ? 1 : a->ldb, b_cs = a->b_kmajor ? a->ldb : 1;   // element (k,n) = b[k*b_rs + n*b_cs]
  a->kernel_used = 0;
  CINEMA_LAUNCH(gemm_generic_kernel, grid, dim3(256), 0, st, p, a_rs, a_cs, b_rs, b_cs);
  return launch_status();
}

// fp8 (OCP e4m3) forward GEMM: D[M,N] = epilogue(alpha * scale_a * scale_b * A8 @ B8^T), A8 [M][lda] and B8 [N][ldb] bytes, k-major (the nn.Linear
// layout), K % 16 == 0, lda / ldb % 16 == 0, 16-byte aligned; epilogue terms as cinema_gemm_bf16's forward classes: bias, exact GELU with optional
// bf16 pre-activation copy (aux_out), fp32 residual; bf16 or fp32 output.  No split-K / split tail.
CINEMA_API int cinema_gemm_fp8(cinema_gemm_args* a, void* stream) {
  if (!a || !a->a || !a->b || a->m <= 0 || a->n <= 0 || a->k <= 0 || !a->scale_a || !a->scale_b) return CINEMA_ERR_BAD_ARG;
  if (!a->d && (!a->out8 || a->out_f32)) return CINEMA_ERR_BAD_ARG;  // D may be omitted only when its 8-bit copy is the output
  if (!a->a_kmajor || !a->b_kmajor || a->accumulate || a->split_k > 1 || a->row_mask || a->residual_bf16 || a->a_rowsum) return CINEMA_ERR_UNSUPPORTED;
  if (a->gelu_in && ((a->ld_gelu & 7) || (((uintptr_t)a->gelu_in) & 15))) return CINEMA_ERR_UNSUPPORTED;
  auto al = [](long long v, int q) { return (v % q) == 0; };
  if (!al(a->k, 16) || !al(a->lda, 16) || !al(a->ldb, 16) || !al(a->ldd, 8) || !al(a->n, 8) || !al((uintptr_t)a->a, 16) || !al((uintptr_t)a->b, 16) ||
      !al((uintptr_t)a->d, 16) || (a->bias && !al((uintptr_t)a->bias, 16)) || (a->residual_f32 && (!al(a->ld_res, 8) || !al((uintptr_t)a->residual_f32, 16))) ||
      (a->aux_out && (!al(a->ld_aux, 8) || !al((uintptr_t)a->aux_out, 16))))
    return CINEMA_ERR_UNSUPPORTED;
  GemmP p;
  p.a = (const bf16_t*)a->a; p.b = (const bf16_t*)a->b; p.d = a->d;
  p.m = a->m; p.n = a->n; p.k = a->k / 2; p.lda = a->lda / 2; p.ldb = a->ldb / 2; p.ldd = a->ldd;  // byte pairs: see gemm_tile<.., FP8>
  p.alpha = a->alpha;
  p.bias = a->bias; p.res_f32 = a->residual_f32; p.res_bf16 = nullptr; p.ld_res = a->ld_res;
  p.gelu_in = a->gelu_in; p.ld_gelu = a->ld_gelu; p.row_mask = nullptr; p.aux_out = a->aux_out; p.ld_aux = a->ld_aux;
  p.act = a->act; p.gelu_deriv = a->gelu_deriv; p.out_f32 = a->out_f32; p.accumulate = 0; p.ws = nullptr; p.a_rowsum = nullptr;
  p.tail_begin = 0; p.tail_split = 0; p.tail_ktiles = 0; p.tail_ws = nullptr; p.tail_cnt = nullptr;
  p.scale_a = a->scale_a; p.scale_b = a->scale_b; p.scale_a_rows = a->scale_a_rows ? 1 : 0; p.conv_taps = nullptr; p.conv_coords = nullptr; p.cZB = 1;
  if (a->out8_amax) {
    if (a->out_f32 || (a->out8 && (!a->out8_inv_scale || (a->ld_out8 & 7) || (((uintptr_t)a->out8) & 7)))) return CINEMA_ERR_BAD_ARG;
    p.out8 = a->out8; p.ld_out8 = a->ld_out8; p.out8_inv = a->out8_inv_scale; p.out8_amax = a->out8_amax;
  } else if (a->out8) return CINEMA_ERR_BAD_ARG;
  if (a->colsum_partials) {
    if (a->out_f32 || (((uintptr_t)a->colsum_partials) & 15)) return CINEMA_ERR_BAD_ARG;
    p.colsum_partials = a->colsum_partials;
  }
  const int nkt = (p.k + BK - 1) / BK;
  p.ktiles_per_split = nkt;
  dim3 grid(((a->m + BM - 1) / BM) * ((a->n + BN - 1) / BN), 1, 1);
  hipStream_t st = (hipStream_t)stream;
  if (p.gelu_in) {
    if (p.out_f32 || p.res_f32 || p.act || p.aux_out || p.bias) return CINEMA_ERR_UNSUPPORTED;
    CINEMA_LAUNCH(gemm_fp8_kernel<EPI_BF16_GELU_GRAD>, grid, dim3(256), 0, st, p); a->kernel_used = 256 + EPI_BF16_GELU_GRAD;
  }
  else if (!p.out_f32 && !p.res_f32 && p.act == 0 && !p.aux_out) { CINEMA_LAUNCH(gemm_fp8_kernel<EPI_BF16>, grid, dim3(256), 0, st, p); a->kernel_used = 256 + EPI_BF16; }
  else if (!p.out_f32 && !p.res_f32 && p.act == 1) { CINEMA_LAUNCH(gemm_fp8_kernel<EPI_BF16_GELU>, grid, dim3(256), 0, st, p); a->kernel_used = 256 + EPI_BF16_GELU; }
  else if (p.out_f32 && p.act == 0 && !p.aux_out) { CINEMA_LAUNCH(gemm_fp8_kernel<EPI_F32>, grid, dim3(256), 0, st, p); a->kernel_used = 256 + EPI_F32; }
  else return CINEMA_ERR_UNSUPPORTED;
  return launch_status();
}

// Implicit-GEMM "same" convolution on channels-last bf16 volumes (reference: the dense 3^n convs of ConvResBlock, cinema/conv.py:320-345, and their data
// gradient): D[r][n] = epilogue(sum_j sum_c x[nbr_j(r)][c] * B[n][j*C + c]) - the im2col matrix is never materialised, the A tiles of the MFMA kernel are
// gathered straight from the volume by the LDS-DMA (one 16-byte piece = 8 channels of one tap; out-of-volume neighbours read a zero page).
//   args->a = x [batch*X*Y*Z][C] (C % 8 == 0), args->m = batch*X*Y*Z, args->b = weights [N][ldb] bf16 with features (tap, channel) zero-padded to
//   ldb = k (k % 8 == 0), conv_taps = device int4 table with k / 8 entries {row delta, (dx+1) | (dy+1) << 2 | (dz+1) << 4, first channel, valid}.
//   Epilogue: bias, fp32 residual; bf16 or fp32 D.
//   conv_zb = ZB > 1 ("z-blocked"): a row is a group of ZB consecutive z voxels (m = batch*X*Y*Z/ZB), its n = ZB*c_out outputs are the c_out channels of those
//   voxels (the same memory as the plain [voxel][c_out] rows), the taps run over (3, 3, ZB + 2) input offsets (dz field = offset from the group's first
//   voxel + 1) and the weights are the block-banded [ZB*c_out][9*(ZB+2)*C] matrix of cinema_conv_weight_zblock.  For c_out = 32 this fills the 128-wide tile
//   with 50 % useful MACs instead of 25 % and halves the gathered bytes per output.
CINEMA_API int cinema_conv_gemm_bf16(cinema_gemm_args* a, void* stream) {
  if (!a || !a->a || !a->b || !a->d || a->m <= 0 || a->n <= 0 || a->k <= 0 || !a->conv_taps || a->conv_x <= 0 || a->conv_y <= 0 || a->conv_z <= 0 || a->conv_c <= 0)
    return CINEMA_ERR_BAD_ARG;
  auto al = [](long long v, int q) { return (v % q) == 0; };
  const int zb = a->conv_zb > 1 ? a->conv_zb : 1;
  if (zb > 6 || a->conv_z % zb || a->m % ((long long)a->conv_x * a->conv_y * (a->conv_z / zb)) != 0) return CINEMA_ERR_BAD_ARG;
  if (!al(a->k, 8) || !al(a->ldb, 8) || !al(a->conv_c, 8) || !al(a->n, 8) || !al(a->ldd, 8) || !al((uintptr_t)a->a, 16) || !al((uintptr_t)a->b, 16) ||
      !al((uintptr_t)a->d, 16) || !al((uintptr_t)a->conv_taps, 16) || (a->bias && !al((uintptr_t)a->bias, 16)) ||
      (a->residual_f32 && (!al(a->ld_res, 8) || !al((uintptr_t)a->residual_f32, 16))))
    return CINEMA_ERR_UNSUPPORTED;
  if (a->accumulate || a->split_k > 1 || a->gelu_in || a->row_mask || a->residual_bf16 || a->a_rowsum || a->aux_out || a->act) return CINEMA_ERR_UNSUPPORTED;
  GemmP p;
  p.a = (const bf16_t*)a->a; p.b = (const bf16_t*)a->b; p.d = a->d;
  p.m = a->m; p.n = a->n; p.k = a->k; p.lda = 0; p.ldb = a->ldb; p.ldd = a->ldd;
  p.alpha = a->alpha;
  p.bias = a->bias; p.res_f32 = a->residual_f32; p.res_bf16 = nullptr; p.ld_res = a->ld_res;
  p.gelu_in = nullptr; p.ld_gelu = 0; p.row_mask = nullptr; p.aux_out = nullptr; p.ld_aux = 0;
  p.act = 0; p.gelu_deriv = 0; p.out_f32 = a->out_f32; p.accumulate = 0; p.ws = nullptr; p.a_rowsum = nullptr;
  p.tail_begin = 0; p.tail_split = 0; p.tail_ktiles = 0; p.tail_ws = nullptr; p.tail_cnt = nullptr;
  p.scale_a = nullptr; p.scale_b = nullptr; p.scale_a_rows = 0;
  p.conv_taps = (const int4*)a->conv_taps; p.cX = a->conv_x; p.cY = a->conv_y; p.cZ = a->conv_z; p.cC = a->conv_c; p.conv_coords = nullptr;
  p.cZB = zb;
  p.ktiles_per_split = (p.k + BK - 1) / BK;
  dim3 grid(((a->m + BM - 1) / BM) * ((a->n + BN - 1) / BN), 1, 1);
  hipStream_t st = (hipStream_t)stream;
  if (p.out_f32) { CINEMA_LAUNCH(gemm_conv_kernel<EPI_F32>, grid, dim3(256), 0, st, p); a->kernel_used = 512 + EPI_F32; }
  else if (!p.res_f32) { CINEMA_LAUNCH(gemm_conv_kernel<EPI_BF16>, grid, dim3(256), 0, st, p); a->kernel_used = 512 + EPI_BF16; }
  else return CINEMA_ERR_UNSUPPORTED;
  return launch_status();
}

// Weight gradient of the implicit convolution: dW[co][(tap, ci)] (+)= sum_r dy[r][co] * x[nbr_tap(r)][ci] (+ bias gradient = column sums of dy through
// a_rowsum): args->a = dy [rows][lda] bf16 (rows = k, m = c_out), args->b = x volume [rows][C], n = ld of the weight rows (taps * C padded to 8),
// conv_taps as in cinema_conv_gemm_bf16 (forward offsets), conv_coords[r] = x | y << 10 | z << 20.  fp32 D [c_out][ldd], split-K through the workspace.
CINEMA_API int cinema_conv_wgrad_bf16(cinema_gemm_args* a, void* stream) {
  if (!a || !a->a || !a->b || !a->d || a->m <= 0 || a->n <= 0 || a->k <= 0 || !a->conv_taps || !a->conv_coords || a->conv_c <= 0) return CINEMA_ERR_BAD_ARG;
  auto al = [](long long v, int q) { return (v % q) == 0; };
  if (!a->out_f32 || !al(a->m, 8) || !al(a->n, 8) || !al(a->lda, 8) || !al(a->ldd, 8) || !al(a->conv_c, 8) || !al((uintptr_t)a->a, 16) || !al((uintptr_t)a->b, 16) ||
      !al((uintptr_t)a->d, 16) || !al((uintptr_t)a->conv_taps, 16) || a->conv_x > 1023 || a->conv_y > 1023 || a->conv_z > 1023)
    return CINEMA_ERR_UNSUPPORTED;
  if (a->bias || a->residual_f32 || a->residual_bf16 || a->gelu_in || a->row_mask || a->aux_out || a->act) return CINEMA_ERR_UNSUPPORTED;
  const int split = a->split_k < 1 ? 1 : a->split_k;
  GemmP p;
  p.a = (const bf16_t*)a->a; p.b = (const bf16_t*)a->b; p.d = a->d;
  p.m = a->m; p.n = a->n; p.k = a->k; p.lda = a->lda; p.ldb = 0; p.ldd = a->ldd;
  p.alpha = a->alpha;
  p.bias = nullptr; p.res_f32 = nullptr; p.res_bf16 = nullptr; p.ld_res = 0; p.gelu_in = nullptr; p.ld_gelu = 0; p.row_mask = nullptr; p.aux_out = nullptr; p.ld_aux = 0;
  p.act = 0; p.gelu_deriv = 0; p.out_f32 = 1; p.accumulate = a->accumulate; p.a_rowsum = a->a_rowsum;
  p.tail_begin = 0; p.tail_split = 0; p.tail_ktiles = 0; p.tail_ws = nullptr; p.tail_cnt = nullptr;
  p.scale_a = nullptr; p.scale_b = nullptr; p.scale_a_rows = 0;
  p.conv_taps = (const int4*)a->conv_taps; p.cX = a->conv_x; p.cY = a->conv_y; p.cZ = a->conv_z; p.cC = a->conv_c; p.conv_coords = (const int*)a->conv_coords;
  p.cZB = a->conv_zb > 1 ? a->conv_zb : 1;
  if (a->conv_z % p.cZB) return CINEMA_ERR_BAD_ARG;
  const int nkt = (a->k + BK - 1) / BK;
  const int sp = split > nkt ? nkt : split;
  p.ktiles_per_split = (nkt + sp - 1) / sp;
  const int gz = (nkt + p.ktiles_per_split - 1) / p.ktiles_per_split;
  if (!a->workspace || a->workspace_bytes < (long long)gz * a->m * a->n * 4 || (((uintptr_t)a->workspace) & 15)) return CINEMA_ERR_BAD_ARG;
  p.ws = (float*)a->workspace;
  dim3 grid(((a->m + BM - 1) / BM) * ((a->n + BN - 1) / BN), 1, gz);
  hipStream_t st = (hipStream_t)stream;
  CINEMA_LAUNCH(gemm_convw_kernel, grid, dim3(256), 0, st, p);
  long long blocks = ((long long)a->m * a->n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  int slices = 1;
  if (a->accumulate && blocks < 256) {
    slices = (int)(512 / blocks);
    if (slices > (gz + 7) / 8) slices = (gz + 7) / 8;
    if (slices < 1) slices = 1;
  }
  const SplitkP sk{(const float*)a->workspace, gz, a->m, a->n, (float*)a->d, a->ldd, a->accumulate, a->alpha};
  launch_lanes(splitk_reduce_kernel, splitk_reduce_lanes_kernel, 2, dim3((unsigned)blocks, slices), dim3(256), 0, st, sk);
  a->kernel_used = 1024;
  return launch_status();
}

CINEMA_API int cinema_gemm_bf16_grouped(cinema_gemm_args* args, int count, void* stream) {
  if (!args || count < 1 || count > 8) return CINEMA_ERR_BAD_ARG;
  auto al8 = [](int v) { return (v & 7) == 0; };
  auto ptr16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  GroupP g;
  g.count = count;
  g.tile_begin[0] = 0;
  for (int i = 0; i < count; i++) {
    const cinema_gemm_args* a = &args[i];
    if (!a->a || !a->b || !a->d || a->m <= 0 || a->n <= 0 || a->k <= 0) return CINEMA_ERR_BAD_ARG;
    // weight-gradient form only: both operands reduction-strided, fp32 output (optionally accumulated), no fused epilogue terms
    if (a->a_kmajor || a->b_kmajor || !a->out_f32 || a->bias || a->residual_f32 || a->residual_bf16 || a->gelu_in || a->row_mask || a->aux_out || a->act)
      return CINEMA_ERR_UNSUPPORTED;
    if (!(al8(a->lda) && al8(a->ldb) && al8(a->ldd) && al8(a->n) && al8(a->m) && ptr16(a->a) && ptr16(a->b) && ptr16(a->d))) return CINEMA_ERR_UNSUPPORTED;
    GemmP& p = g.p[i];
    p.a = (const bf16_t*)a->a; p.b = (const bf16_t*)a->b; p.d = a->d;
    p.m = a->m; p.n = a->n; p.k = a->k; p.lda = a->lda; p.ldb = a->ldb; p.ldd = a->ldd;
    p.alpha = a->alpha;
    p.bias = nullptr; p.res_bf16 = nullptr; p.gelu_in = nullptr; p.ld_gelu = 0; p.row_mask = nullptr; p.aux_out = nullptr; p.ld_aux = 0;
    p.act = 0; p.gelu_deriv = 0; p.out_f32 = 1; p.accumulate = 0;
    p.res_f32 = a->accumulate ? (const float*)a->d : nullptr;  // one owner per element: plain read-modify-write
    p.ld_res = a->ldd;
    p.ktiles_per_split = (a->k + BK - 1) / BK;
    p.ws = nullptr; p.a_rowsum = a->a_rowsum;
    p.tail_begin = 0; p.tail_split = 0; p.tail_ktiles = 0; p.tail_ws = nullptr; p.tail_cnt = nullptr; p.scale_a = nullptr; p.scale_b = nullptr; p.scale_a_rows = 0; p.conv_taps = nullptr; p.conv_coords = nullptr; p.cZB = 1; p.scale_a_rows = 0;
    g.tile_begin[i + 1] = g.tile_begin[i] + ((a->m + BM - 1) / BM) * ((a->n + BN - 1) / BN);
    args[i].kernel_used = 64;  // the grouped kernel
  }
  for (int i = count + 1; i < 9; i++) g.tile_begin[i] = g.tile_begin[count];
  CINEMA_LAUNCH((gemm_mfma_grouped_kernel<false, false, EPI_F32>), dim3(g.tile_begin[count]), dim3(256), 0, (hipStream_t)stream, g);
  return launch_status();
}

// fp32 rows (optionally gathered through row_idx), n % 4 == 0: thread = 4 consecutive columns, 4 row lanes per block, 4 independent rows in
// flight per thread (the scalar kernel above walked its rows with one dependent 4-byte load at a time: 54 us for the decoder's mask-token
// gradient, 27648 x 512 fp32 = 0.4 TB/s)
__global__ __launch_bounds__(256) void colsum_f32_vec_kernel(const float* x, const int* row_idx, int m, int n, int ldx, float* out, int rows_per_block) {
  __shared__ float4 part[4][64];
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + lane) * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(m, r0 + rows_per_block);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < n) {
    for (int r = r0 + rl; r < r1; r += 16) {
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int rr = r + 4 * q;
        v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr < r1) v[q] = *reinterpret_cast<const float4*>(x + (size_t)(row_idx ? row_idx[rr] : rr) * ldx + col);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
    }
  }
  part[rl][lane] = s;
  __syncthreads();
  if (rl == 0 && col < n) {
    const float4 a = part[0][lane], b = part[1][lane], c = part[2][lane], d = part[3][lane];
    unsafeAtomicAdd(out + col, (a.x + b.x) + (c.x + d.x));
    unsafeAtomicAdd(out + col + 1, (a.y + b.y) + (c.y + d.y));
    unsafeAtomicAdd(out + col + 2, (a.z + b.z) + (c.z + d.z));
    unsafeAtomicAdd(out + col + 3, (a.w + b.w) + (c.w + d.w));
  }
}

CINEMA_API int cinema_colsum(const void* x, int x_dtype, const int* row_idx, int m, int n, int ldx, float* out, void* stream) {
  if (!x || !out || m <= 0 || n <= 0) return CINEMA_ERR_BAD_ARG;
  if (x_dtype == 0 && !row_idx && !(n & 7) && !(ldx & 7) && !(((uintptr_t)x) & 15)) {
    const int col_blocks = (n + 255) / 256;
    int row_chunks = (1024 + col_blocks - 1) / col_blocks;
    if (row_chunks > (m + 63) / 64) row_chunks = (m + 63) / 64;
    const int rpb = (((m + row_chunks - 1) / row_chunks) + 7) / 8 * 8;
    dim3 grid(col_blocks, (m + rpb - 1) / rpb);
    CINEMA_LAUNCH(colsum_bf16_vec_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, m, n, ldx, out, rpb);
    return launch_status();
  }
  if (x_dtype == 1 && !(n & 3) && !(ldx & 3) && !(((uintptr_t)x) & 15)) {
    const int col_blocks = (n + 255) / 256;
    int chunks = (1024 + col_blocks - 1) / col_blocks;
    if (chunks > (m + 31) / 32) chunks = (m + 31) / 32;
    const int rpb = (m + chunks - 1) / chunks;
    dim3 grid(col_blocks, (m + rpb - 1) / rpb);
    CINEMA_LAUNCH(colsum_f32_vec_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, row_idx, m, n, ldx, out, rpb);
    return launch_status();
  }
  int chunks = (m + 511) / 512;
  if (chunks > 256) chunks = 256;
  const int rpb = (m + chunks - 1) / chunks;
  dim3 grid((n + 63) / 64, (m + rpb - 1) / rpb);
  CINEMA_LAUNCH(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, x_dtype, row_idx, m, n, ldx, out, rpb);
  return launch_status();
}
