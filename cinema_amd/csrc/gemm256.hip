// Persistent 256x256x64 bf16 GEMM for gfx950 with the split reduction finished inside the launch.
//
// One workgroup of 8 waves (2 x 4, a wave owns 128 x 64 outputs = 4 x 2 MFMA 32x32x16 blocks) per CU, 128 KiB of LDS as two 64 KiB k-tile
// stages filled by LDS-DMA (the 256-row operand tile is two of gemm_shared.cuh's 128-row sub-tiles, so loaders, swizzles and fragment reads are
// the 128x128 kernel's).  Half the staged bytes per FLOP of the 128x128 tile - the 128x128 loop is bound by that stream.
//
// Work is a list of PIECES: (problem, output tile, k-tile range).  Three schedules:
//   split   - every tile of problem i is cut into split[i] equal k-slices (the host balances the slice lengths over up to 12 problems: the weight
//             gradients of one transformer block in one launch); workgroups walk the pieces grid-stride, slice-major so that neighbours share panels
//   stream  - the (tile, k-tile) units of one problem are dealt to the workgroups as equal contiguous ranges (tile counts like 129 on 256 CUs)
//   split + remainder (CINEMA_P256_REMAINDER=1, off by default: 2 % faster for 9 % more traffic) - every tile is cut into slices of the SAME length L ~ units / CUs, one
//             per workgroup, and the rest of the tile's reduction (shorter than L) goes to workgroups that take several such rests one after the other: the
//             weight gradients of an encoder block are 108 tiles x 343 units = 144.7 units per CU; equal slices give 216 pieces of 172 units on 256 CUs,
//             this schedule 216 slices of 147 + 108 rests of 49 packed three to a workgroup (252 CUs, 147 units each)
// A tile cut into n pieces is finished by its LAST ARRIVER: a piece takes a ticket from the tile's arrival counter; tickets 0..n-2 store their
// accumulators as a fragment-ordered fp32 slot (write-through sc1 stores, 1 KiB per wave-instruction), drain, and bump the tile's publish counter;
// ticket n-1 keeps its accumulators, waits for n-1 publishes (its partners arrived before it and never wait themselves: no deadlock for any dispatch
// order or co-residency), sums the slots in piece order with its own registers at their position (rounding independent of the arrival order) and runs the
// fused epilogue.  The last arriver zeroes both counters: no per-launch memset, no reduce
// launch, no fp32 slab round trip through a second kernel.  Visibility follows the CDNA4 recipe (sc1 payload + per-wave vmcnt(0) + barrier + relaxed
// agent counter; consumer: relaxed poll, ONE agent acquire, barrier, plain loads) and does not depend on placement.
#include "gemm_shared.cuh"

namespace {

constexpr int P_TILE = 256;
constexpr int P_SUB = 16384;                 // one 128-row (or 128-column) operand sub-tile of a k-tile
constexpr int P_STAGE = 4 * P_SUB;           // A0 A1 B0 B1
constexpr int P_CTL = 2 * P_STAGE;           // control words behind the two stages
constexpr int P_SMEM = P_CTL + 64;
constexpr int P_SLOT_FLOATS = P_TILE * P_TILE;
constexpr int P_MAX = 12;
constexpr int P_COUNTER_BYTES = 65536;       // head of the workspace: 2 counters per output tile (zero on entry, left zero)

// what a piece needs of a problem (GemmP carries the fields of every other GEMM mode as well: 12 of them would not fit the kernel-argument segment)
struct SlimP {
  const bf16_t* a; const bf16_t* b; void* d;
  int m, n, k, lda, ldb, ldd;
  const float* bias; const float* res_f32; const bf16_t* gelu_in; bf16_t* aux_out; float* a_rowsum;
  int ld_res, ld_gelu, ld_aux, act, out_f32, gelu_deriv;
  float alpha;
  const float* scale_a; const float* scale_b;   // fp8 weight-gradient form (LOOP 3): per-tensor dequantisation scales of the 8-bit operands, NULL otherwise
};
__device__ __forceinline__ GemmP expand(const SlimP& s) {
  GemmP p;
  p.a = s.a; p.b = s.b; p.d = s.d; p.m = s.m; p.n = s.n; p.k = s.k; p.lda = s.lda; p.ldb = s.ldb; p.ldd = s.ldd; p.alpha = s.alpha;
  p.bias = s.bias; p.res_f32 = s.res_f32; p.res_bf16 = nullptr; p.ld_res = s.ld_res; p.gelu_in = s.gelu_in; p.ld_gelu = s.ld_gelu;
  p.row_mask = nullptr; p.aux_out = s.aux_out; p.ld_aux = s.ld_aux; p.act = s.act; p.gelu_deriv = s.gelu_deriv; p.out_f32 = s.out_f32; p.accumulate = 0;
  p.ktiles_per_split = 0; p.ws = nullptr; p.a_rowsum = s.a_rowsum; p.tail_begin = 0; p.tail_split = 0; p.tail_ktiles = 0; p.tail_ws = nullptr; p.tail_cnt = nullptr;
  p.scale_a = s.scale_a; p.scale_b = s.scale_b; p.scale_a_rows = 0; p.conv_taps = nullptr; p.cX = p.cY = p.cZ = p.cC = 0; p.cZB = 1; p.conv_coords = nullptr;
  return p;
}

struct P256 {
  SlimP p[P_MAX];
  int count, mode;               // mode 0 = split, 1 = stream, 2 = split + remainder
  int tile_begin[P_MAX + 1];     // prefix sums of the 256x256 tile counts (global tile id = counter index)
  int piece_begin[P_MAX + 1];    // split: prefix sums of tiles_i * split_i
  int split[P_MAX], kts[P_MAX];  // split: slices per tile, k-tiles per slice
  int nkt[P_MAX];
  int nbig[P_MAX];               // split + remainder: full slices per tile (kts[i] units each); the rest [nbig * kts, nkt) is the tile's remainder piece (none: rem_per = 0)
  int big_begin[P_MAX + 1];      //   prefix sums of tiles_i * nbig_i: logical workgroup ids of the full slices (= their slot ids); the remainder of global tile t uses slot n_big + t
  int rem_begin[P_MAX + 1];      //   prefix sums of the remainder workgroups per problem
  int rem_per[P_MAX];            //   remainder pieces (consecutive tiles) per remainder workgroup
  long long units;               // stream: tiles_0 * nkt_0
  int per, rem;                  // stream: units per workgroup; the first `rem` workgroups take one more
  float* slots;                  // fp32 partial slots [..][P_SLOT_FLOATS]
  unsigned* counters;
  unsigned* error;               // set to 1 when a bounded spin gives up (never in a healthy run)
};

typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ long long stream_start(const P256& g, int wg) { return (long long)wg * g.per + min(wg, g.rem); }
__device__ __forceinline__ int stream_owner_of(const P256& g, long long u) {  // workgroup whose range holds unit u
  const long long big = (long long)g.rem * (g.per + 1);
  return u < big ? (int)(u / (g.per + 1)) : g.rem + (int)((u - big) / g.per);
}

// acc + the 8 bf16 of a fragment: four v_dot2c_f32_bf16 against (1, 1) instead of eight shift / mask + add pairs (the bias-gradient row sums of the
// weight-gradient pieces ran ~100 VALU instructions per phase on two of the eight waves, and everyone waits for them at the phase barriers)
__device__ __forceinline__ float frag_sum8_dot(const short8v& f, float acc) {
  typedef uint32_t u32x4f __attribute__((ext_vector_type(4)));
  const u32x4f w = __builtin_bit_cast(u32x4f, f);
  const uint32_t ones = 0x3f803f80u;
#pragma unroll
  for (int e = 0; e < 4; e++) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(w[e]), "v"(ones));
  return acc;
}

// ---- main loop, form 0: k-tiles of 64, two 64 KiB stages, one barrier per k-tile (every wave: fragment reads, then MFMAs, hipcc's interleave)
template <bool A_KMAJ, bool B_KMAJ>
__device__ __forceinline__ void p256_loop64(const GemmP& p, int m0, int n0, int kt_begin, int kt_end, float16v (&acc)[4][2], float (&rs)[4], bool do_rowsum,
                                            char* smem) {
  using AIO = TileIO<A_KMAJ>;
  using BIO = TileIO<B_KMAJ>;
  const int lane = threadIdx.x & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wr = wave_u >> 2, wc = wave_u & 3;      // wave grid 2 (m) x 4 (n)
  const int sub = wave_u >> 2, w4 = wave_u & 3;     // loader role: waves 0-3 fill sub-tile 0 of A and B, waves 4-7 sub-tile 1
  const bf16_t* zero_page = reinterpret_cast<const bf16_t*>(g_zero_page);
  const typename AIO::Src4 asrc = AIO::src4(p.a, p.lda, m0 + sub * 128, p.m, lane, w4);
  const typename BIO::Src4 bsrc = BIO::src4(p.b, p.ldb, n0 + sub * 128, p.n, lane, w4);
  const size_t astep = AIO::k_step(p.lda), bstep = BIO::k_step(p.ldb);
  const uint32_t smem_addr = __builtin_amdgcn_readfirstlane(lds_address(smem));
  auto load_tile = [&](int stage, int kt) {
    const uint32_t a_dst = smem_addr + stage * P_STAGE + sub * P_SUB, b_dst = a_dst + 2 * P_SUB;
    if ((kt + 1) * BK <= p.k) {
      AIO::glds_at(a_dst, asrc, (size_t)kt * astep, w4);
      BIO::glds_at(b_dst, bsrc, (size_t)kt * bstep, w4);
    } else {  // ragged last k-tile: per-element range tests, zero page
      AIO::glds(smem + stage * P_STAGE + sub * P_SUB, p.a, p.lda, m0 + sub * 128, p.m, kt * BK, p.k, lane, w4, zero_page);
      BIO::glds(smem + stage * P_STAGE + (2 + sub) * P_SUB, p.b, p.ldb, n0 + sub * 128, p.n, kt * BK, p.k, lane, w4, zero_page);
    }
  };
  load_tile(0, kt_begin);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; kt++) {
    const int cur = (kt - kt_begin) & 1;
    const char* sa = smem + cur * P_STAGE + wr * P_SUB;
    const char* sb = smem + cur * P_STAGE + (2 + (wc >> 1)) * P_SUB;
    const int bcol = (wc & 1) * 64;
    const bool more = kt + 1 < kt_end;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      short8v fa[4], fb[2];
#pragma unroll
      for (int i = 0; i < 4; i++) fa[i] = AIO::frag(sa, i * 32, ks, lane);
      fb[0] = BIO::frag(sb, bcol, ks, lane);
      fb[1] = BIO::frag(sb, bcol + 32, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      if (do_rowsum) {
#pragma unroll
        for (int i = 0; i < 4; i++) rs[i] += frag_sum8(fa[i]);
      }
      if (ks == 0 && more) load_tile(cur ^ 1, kt + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

// ---- main loop, forms 1 / 2: PHASES of 32 k, ring of four 32 KiB half-stages, the two waves of a SIMD half a phase apart.
// A phase of one wave = READ segment (its 12 operand fragments of the phase: 12 ds_read_b128 or 24 ds_read_b64_tr_b16) | barrier | MFMA segment (16 MFMAs at
// priority 1) | barrier.  Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave computes while its partner reads: the matrix pipe sees MFMA
// segments back to back, the LDS sees one group of four readers at a time.  The LDS-DMA of the phase three ahead goes into the half-stage last read in phase
// p - 1 and is waited for with a COUNTED vmcnt, so no wave ever drains its queue inside the loop.
//   form 1 (rounds 3-4): DMA(p + 3) issued between the MFMAs of MFMA(p), waited for at the end of READ(p + 2).
//   form 2 (default since round 4, CINEMA_P256_LOOP=1 selects form 1): DMA(p + 3) issued by the READING wave at the end of READ(p), waited for at the end of
//           READ(p + 2) - see the comment in the loop.  Weight-gradient groups of the step 10-20 % faster (encoder block 213 -> 186 us, ViT-Large block 423 -> 353 us,
//           8192^3 1015 -> 1157 TF), forward / data-gradient layouts 1-3 %, bit-identical results (profiles/r04_aa_p256_loop_ab.txt).
#define P256_BAR() do { asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
constexpr int P_HS = 32768;   // half-stage: A sub-tiles 0, 1 then B sub-tiles 0, 1, 8 KiB each
template <bool A_KMAJ, bool B_KMAJ, int FORM>
__device__ __forceinline__ void p256_loop32(const GemmP& p, int m0, int n0, int ph_begin, int ph_end, float16v (&acc)[4][2], float (&rs)[4], bool do_rowsum,
                                            char* smem) {
  using AIO = TileIO32<A_KMAJ>;
  using BIO = TileIO32<B_KMAJ>;
  const int lane = threadIdx.x & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wr = wave_u >> 2, wc = wave_u & 3;
  const int sub = wave_u >> 2, w4 = wave_u & 3;
  const bf16_t* zero_page = reinterpret_cast<const bf16_t*>(g_zero_page);
  const typename AIO::Src2 asrc = AIO::src2(p.a, p.lda, m0 + sub * 128, p.m, lane, w4);
  const typename BIO::Src2 bsrc = BIO::src2(p.b, p.ldb, n0 + sub * 128, p.n, lane, w4);
  const size_t astep = AIO::k_step(p.lda), bstep = BIO::k_step(p.ldb);
  const uint32_t smem_addr = __builtin_amdgcn_readfirstlane(lds_address(smem));
  // validity of this lane's pieces in a ragged last phase (k0 + 32 > K): k-major operand: the lane's 16-byte chunk covers 8 k; reduction-strided
  // operand: piece pss covers k row (pss * 4 + w4) * 4 + lane / 16
  auto issue = [&](int ph) {
    const uint32_t dst = smem_addr + (ph & 3) * P_HS + sub * 8192 + w4 * 1024;
    const size_t ka = (size_t)ph * astep, kb = (size_t)ph * bstep;
    const bf16_t *a0 = asrc.p[0] + ka, *a1 = asrc.p[1] + ka, *b0 = bsrc.p[0] + kb, *b1 = bsrc.p[1] + kb;
    if ((ph + 1) * 32 > p.k) {
      const int k0 = ph * 32;
      if (A_KMAJ) {
        const int r0 = w4 * 16 + (lane >> 2), r1 = r0 + 64;
        if (k0 + (((lane & 3) ^ ((r0 >> 2) & 3)) << 3) >= p.k) a0 = zero_page;
        if (k0 + (((lane & 3) ^ ((r1 >> 2) & 3)) << 3) >= p.k) a1 = zero_page;
      } else {
        if (k0 + w4 * 4 + (lane >> 4) >= p.k) a0 = zero_page;
        if (k0 + (4 + w4) * 4 + (lane >> 4) >= p.k) a1 = zero_page;
      }
      if (B_KMAJ) {
        const int r0 = w4 * 16 + (lane >> 2), r1 = r0 + 64;
        if (k0 + (((lane & 3) ^ ((r0 >> 2) & 3)) << 3) >= p.k) b0 = zero_page;
        if (k0 + (((lane & 3) ^ ((r1 >> 2) & 3)) << 3) >= p.k) b1 = zero_page;
      } else {
        if (k0 + w4 * 4 + (lane >> 4) >= p.k) b0 = zero_page;
        if (k0 + (4 + w4) * 4 + (lane >> 4) >= p.k) b1 = zero_page;
      }
    }
    glds16x4(dst, dst + 4096, dst + 16384, dst + 16384 + 4096, a0, a1, b0, b1);
  };
  constexpr bool DMA_IN_READ = FORM == 2;
  // Transpose reads of a reduction-strided [32 k][128] sub-tile (TileIO32<false>::frag) with the address split into a per-lane, per-fragment base
  // (computed here, once per piece) and immediates: byte = kr * 256 + ((col / 8) ^ (kr & 3) * 4) * 16 + (col & 7) * 2 with kr = 16 ks + 8 (q4 / 2) + t / 4
  // (+ 4 for the second half), col = 32 f + 16 (q4 & 1) + 4 (t & 3): kr & 3 = t / 4 is a lane constant, so the swizzle turns fragment f into
  // 64 * (f ^ t / 4) and everything else is lane part + ks * 4096 + half * 1024 (as one function of (kr, col) the compiler rebuilt ~25 adds per phase).
  int tr_a[4], tr_b[2];
  {
    const int q4 = lane >> 4, t = lane & 15, mm = t >> 2;
    const int lane_part = (q4 >> 1) * 2048 + mm * 256 + (2 * (q4 & 1) + ((t & 3) >> 1)) * 16 + (t & 1) * 8;
#pragma unroll
    for (int i = 0; i < 4; i++) tr_a[i] = wr * 8192 + lane_part + 64 * (i ^ mm);
#pragma unroll
    for (int j = 0; j < 2; j++) tr_b[j] = 16384 + (wc >> 1) * 8192 + lane_part + 64 * ((2 * (wc & 1) + j) ^ mm);
  }
  auto tr_frag = [&](const char* base, int ks) {
    const short4v lo = lds_tr16_b64(base + ks * 4096), hi = lds_tr16_b64(base + ks * 4096 + 1024);
    short8v out;
    out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; out[3] = lo[3];
    out[4] = hi[0]; out[5] = hi[1]; out[6] = hi[2]; out[7] = hi[3];
    return out;
  };
  const int nph = ph_end - ph_begin;
  // prologue: phases 0..2 in flight, phase 0 landed
  issue(ph_begin);
  if (nph > 1) issue(ph_begin + 1);
  if (nph > 2) issue(ph_begin + 2);
  if (nph > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nph > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  P256_BAR();
  if (wr == 1) P256_BAR();   // waves 4-7 run one barrier behind
  const int bcol = (wc & 1) * 64;
  for (int q = 0; q < nph; q++) {
    const int ph = ph_begin + q;
    const char* hs = smem + (ph & 3) * P_HS;
    const char* sa = hs + wr * 8192;
    const char* sb = hs + 16384 + (wc >> 1) * 8192;
    short8v fa[2][4], fb[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
      for (int j = 0; j < 2; j++) fb[ks][j] = B_KMAJ ? BIO::frag(sb, bcol + j * 32, ks, lane) : tr_frag(hs + tr_b[j], ks);
#pragma unroll
      for (int i = 0; i < 4; i++) fa[ks][i] = A_KMAJ ? AIO::frag(sa, i * 32, ks, lane) : tr_frag(hs + tr_a[i], ks);
    }
    if constexpr (DMA_IN_READ) {
      // form 2: the READING wave issues DMA(ph + 3), after its fragment reads.  The CU accepts an LDS-DMA piece every ~32-36 clocks in this loop (shader-clock probe,
      // tools/p256_phase_probe.py: four waves issuing four pieces each take ~550 clocks), and a wave that waits for the queue issues nothing else: between the MFMAs
      // (form 1) that wait came out of the matrix segment (measured 811 clocks for 16 MFMAs = 512), here it is spent by the wave whose partner on the SIMD computes
      // (MFMA segment 577, READ 700 with the probe's stamps).  The half-stage (ph - 1) & 3 was last read by the other wave group one interval ago; every wave
      // drains its fragment reads BEFORE the barrier that ends its READ segment (free: it waits there anyway), so those reads have completed.  DMA(ph + 1) must
      // have landed before that barrier; DMA(ph + 2) and DMA(ph + 3) stay in flight.
      __builtin_amdgcn_sched_barrier(0);
      if (q + 3 < nph) {
        issue(ph + 3);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if (q + 2 < nph) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      // DMA(ph + 1) must have landed before the barrier that lets anyone read it; DMA(ph + 2) (the youngest, if issued) stays in flight
      if (q + 2 < nph) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    P256_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
      if (!DMA_IN_READ && ks == 0 && q + 3 < nph) issue(ph + 3);
      if (do_rowsum) {
#pragma unroll
        for (int i = 0; i < 4; i++) rs[i] = frag_sum8_dot(fa[ks][i], rs[i]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    P256_BAR();
  }
  if (wr == 0) P256_BAR();
}

// ---- main loop, form 3: WEIGHT GRADIENT ON 8-BIT OPERANDS (BASELINE config 5, "fp8 MFMA path"): dW[n][k] = sa * sb * sum_t dY8[t][n] X8[t][k] with both operands
// stored as the forward / backward passes produce them - row-major [token][feature] bytes (OCP e4m3), per-TENSOR scales - i.e. reduction-strided like the bf16
// weight gradient.  Same schedule as form 1 (ring of four 32 KiB half-stages, two waves per SIMD half a phase apart, counted vmcnt), but a phase is 64 TOKENS:
// a 128-feature sub-tile of a phase is [64 token rows][128 B] = 8 KiB, filled by LDS-DMA pieces of 8 rows x 128 B; v_mfma_scale_f32_32x32x64_f8f6f4 (unit
// block scales) wants 32 consecutive k per lane for one feature, which ds_read_b64_tr_b8 delivers from this layout: a 16-lane group reads an [8 token rows] x
// [16 features] block, lane t supplies row t / 2, bytes 8 (t & 1) .., and receives feature t at the 8 tokens (tools/probe/probe_tr8.hip, verified on the GPU
// together with the MFMA operand layout).  Per phase and wave: 24 transpose reads, 8 MFMAs of twice the bf16 instruction's FLOPs per cycle, and HALF the
// LDS-DMA pieces per FLOP.  Swizzle: 16-byte chunk position = chunk ^ (((row >> 1) & 3) << 1) - the 8 rows x 2 chunk parities of one LDS cycle's two
// 16-lane groups then cover the 64 banks exactly once; applied to the DMA source address and, as a lane constant, to the read address.
typedef int i32x2v __attribute__((ext_vector_type(2)));
typedef int i32x8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x2v lds_tr8_b64(const void* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr8_b64_v2i32((i32x2v __attribute__((address_space(3)))*)(lds_ptr));
}
template <bool DMA_IN_READ>
__device__ __forceinline__ void p256_loop_fp8w(const GemmP& p, int m0, int n0, int ph_begin, int ph_end, float16v (&acc)[4][2], char* smem) {
  const int lane = threadIdx.x & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wr = wave_u >> 2, wc = wave_u & 3;
  const int sub = wave_u >> 2, w4 = wave_u & 3;
  const uint8_t* zero_page = reinterpret_cast<const uint8_t*>(g_zero_page);
  const uint8_t* pa = reinterpret_cast<const uint8_t*>(p.a);
  const uint8_t* pb = reinterpret_cast<const uint8_t*>(p.b);
  // DMA sources of phase 0: piece pss covers token rows (pss * 4 + w4) * 8 + lane / 8 of the phase, the lane's LDS chunk position lane & 7 holds the
  // global chunk (lane & 7) ^ swizzle(row); out-of-range features re-read feature 0 (their outputs are never stored)
  const uint8_t *asrc[2], *bsrc[2];
  int prow[2];
#pragma unroll
  for (int pss = 0; pss < 2; pss++) {
    const int rr = (pss * 4 + w4) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (((rr >> 1) & 3) << 1);
    int ca = m0 + sub * 128 + c * 16, cb = n0 + sub * 128 + c * 16;
    ca = ca < p.m ? ca : 0; cb = cb < p.n ? cb : 0;
    prow[pss] = rr;
    asrc[pss] = pa + (size_t)rr * p.lda + ca;
    bsrc[pss] = pb + (size_t)rr * p.ldb + cb;
  }
  const size_t astep = (size_t)64 * p.lda, bstep = (size_t)64 * p.ldb;
  const uint32_t smem_addr = __builtin_amdgcn_readfirstlane(lds_address(smem));
  auto issue = [&](int ph) {
    const uint32_t dst = smem_addr + (ph & 3) * P_HS + sub * 8192 + w4 * 1024;
    const size_t ka = (size_t)ph * astep, kb = (size_t)ph * bstep;
    const uint8_t *a0 = asrc[0] + ka, *a1 = asrc[1] + ka, *b0 = bsrc[0] + kb, *b1 = bsrc[1] + kb;
    if ((ph + 1) * 64 > p.k) {  // ragged last phase: token rows >= K read the zero page
      const int k0 = ph * 64;
      if (k0 + prow[0] >= p.k) { a0 = zero_page; b0 = zero_page; }
      if (k0 + prow[1] >= p.k) { a1 = zero_page; b1 = zero_page; }
    }
    glds16x4(dst, dst + 4096, dst + 16384, dst + 16384 + 4096, a0, a1, b0, b1);
  };
  int tr_a[4], tr_b[2];
  {
    const int q4 = lane >> 4, t = lane & 15, mm = (t >> 2) & 3;
    const int lane_part = (32 * (q4 >> 1) + (t >> 1)) * 128 + (q4 & 1) * 16 + (t & 1) * 8;
#pragma unroll
    for (int i = 0; i < 4; i++) tr_a[i] = wr * 8192 + lane_part + 32 * (i ^ mm);
#pragma unroll
    for (int j = 0; j < 2; j++) tr_b[j] = 16384 + (wc >> 1) * 8192 + lane_part + 32 * ((2 * (wc & 1) + j) ^ mm);
  }
  auto tr_frag = [&](const char* base) {
    i32x8v out;
#pragma unroll
    for (int h = 0; h < 4; h++) {
      const i32x2v v = lds_tr8_b64(base + h * 1024);
      out[2 * h] = v[0]; out[2 * h + 1] = v[1];
    }
    return out;
  };
  const int nph = ph_end - ph_begin;
  issue(ph_begin);
  if (nph > 1) issue(ph_begin + 1);
  if (nph > 2) issue(ph_begin + 2);
  if (nph > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nph > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  P256_BAR();
  if (wr == 1) P256_BAR();   // waves 4-7 run one barrier behind
  for (int q = 0; q < nph; q++) {
    const int ph = ph_begin + q;
    const char* hs = smem + (ph & 3) * P_HS;
    i32x8v fa[4], fb[2];
#pragma unroll
    for (int j = 0; j < 2; j++) fb[j] = tr_frag(hs + tr_b[j]);
#pragma unroll
    for (int i = 0; i < 4; i++) fa[i] = tr_frag(hs + tr_a[i]);
    if constexpr (DMA_IN_READ) {  // as form 2 of the bf16 loop: the reading wave issues DMA(ph + 3) and drains its fragment reads before the barrier
      __builtin_amdgcn_sched_barrier(0);
      if (q + 3 < nph) {
        issue(ph + 3);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if (q + 2 < nph) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      if (q + 2 < nph) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    P256_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int j = 0; j < 2; j++)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[j], fa[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      if (!DMA_IN_READ && i == 1 && q + 3 < nph) issue(ph + 3);
    }
    __builtin_amdgcn_s_setprio(0);
    P256_BAR();
  }
  if (wr == 0) P256_BAR();
}

// where the pieces of this piece's tile publish: pieces first .. first + n_pieces - 1 in summation order, `me` among them
struct PieceRed {
  float* my_slot;
  const float* slot0; long long stride; int n_big;   // split: pieces 0 .. n_big - 1 at slot0 + piece * stride,
  const float* rem_slot;                             //        piece n_big (the remainder, if any) here
  int first; long long tile_unit0;                   // stream: pieces = workgroups first .., slot by stream_start
  int me;
};

template <bool A_KMAJ, bool B_KMAJ, int EPI, int LOOP>
__device__ __forceinline__ void p256_piece(const P256& g, const GemmP& p, int gtile, int tile, int kt_begin, int kt_end, int n_pieces, const PieceRed& rd, char* smem) {
  float* my_slot = rd.my_slot;
  const int first_other = rd.first, n_other = n_pieces - 1, me = rd.me;
  const long long tile_unit0 = rd.tile_unit0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wr = wave_u >> 2, wc = wave_u & 3;      // wave grid 2 (m) x 4 (n)
  const int tiles_n = (p.n + P_TILE - 1) / P_TILE, tiles_m = (p.m + P_TILE - 1) / P_TILE;
  int tm, tn;
  tile_of(tile, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * P_TILE, n0 = tn * P_TILE;

  float16v acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  float rs[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_rowsum = !A_KMAJ && p.a_rowsum != nullptr && wc == 0 && n0 == 0;
  if constexpr (LOOP == 0) p256_loop64<A_KMAJ, B_KMAJ>(p, m0, n0, kt_begin, kt_end, acc, rs, do_rowsum, smem);
  else if constexpr (LOOP == 3 || LOOP == 10) p256_loop_fp8w<LOOP == 10>(p, m0, n0, kt_begin, kt_end, acc, smem);
  else p256_loop32<A_KMAJ, B_KMAJ, LOOP>(p, m0, n0, kt_begin, kt_end, acc, rs, do_rowsum, smem);

  if (do_rowsum) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float t = rs[i] + __shfl_xor(rs[i], 32, 64);
      const int m = m0 + wr * 128 + i * 32 + lane;
      if (lane < 32 && m < p.m) unsafeAtomicAdd(p.a_rowsum + m, t);
    }
  }

  // ---- split reduction: last arriver finishes the tile
  if (n_pieces > 1) {
    unsigned* cnt = g.counters + 2 * gtile;
    volatile unsigned* ctl = reinterpret_cast<volatile unsigned*>(smem + P_CTL);
    if (tid == 0) ctl[0] = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = ctl[0];
    __syncthreads();
    if (ticket != (unsigned)(n_pieces - 1)) {
      // publish: fragment order, wave w / register quad q / lane l -> float4 index (w * 32 + q) * 64 + l
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(my_slot, 0, P_SLOT_FLOATS * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int idx = (wave_u * 32 + (i * 2 + j) * 4 + q) * 64 + lane;
            u32x4v v = {__float_as_uint(acc[i][j][4 * q]), __float_as_uint(acc[i][j][4 * q + 1]), __float_as_uint(acc[i][j][4 * q + 2]),
                        __float_as_uint(acc[i][j][4 * q + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, idx * 16, 0, 16 /* sc1: write-through */);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (tid == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(n_pieces - 1)) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1u << 26)) { *g.error = 1u; break; }
      }
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // left zero for the next launch on this workspace
      __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    // the pieces are summed in piece order with this workgroup's registers at their own position: the rounding does not depend on who arrived last
    // (one 32x32 block = 4 float4 per lane at a time: a second copy of more accumulators than that spills)
#pragma unroll
    for (int ij = 0; ij < 8; ij++) {
      const int i = ij >> 1, j = ij & 1;
      float4 r[4];
#pragma unroll
      for (int q = 0; q < 4; q++) r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int o = 0; o <= n_other; o++) {
        const int s = first_other + o;   // slice (split) / workgroup (stream) of piece o
        if (s == me) {
#pragma unroll
          for (int q = 0; q < 4; q++) { r[q].x += acc[i][j][4 * q]; r[q].y += acc[i][j][4 * q + 1]; r[q].z += acc[i][j][4 * q + 2]; r[q].w += acc[i][j][4 * q + 3]; }
          continue;
        }
        const float* src;
        if (g.mode != 1) {
          src = s < rd.n_big ? rd.slot0 + (long long)s * rd.stride : rd.rem_slot;   // slice-major piece order: same tile, slice s
        } else {
          const int which = stream_start(g, s) >= tile_unit0 ? 0 : 1;
          src = g.slots + ((long long)s * 2 + which) * P_SLOT_FLOATS;
        }
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4 t[4];
#pragma unroll
        for (int q = 0; q < 4; q++) t[q] = s4[(wave_u * 32 + ij * 4 + q) * 64 + lane];
#pragma unroll
        for (int q = 0; q < 4; q++) { r[q].x += t[q].x; r[q].y += t[q].y; r[q].z += t[q].z; r[q].w += t[q].w; }
      }
#pragma unroll
      for (int q = 0; q < 4; q++) { acc[i][j][4 * q] = r[q].x; acc[i][j][4 * q + 1] = r[q].y; acc[i][j][4 * q + 2] = r[q].z; acc[i][j][4 * q + 3] = r[q].w; }
    }
  }

  // ---- fused epilogue, one 32 x 64 block row of the wave tile at a time through 8 KiB of LDS per wave (both stages are free)
  float* stg = reinterpret_cast<float*>(smem + wave * 8192);
  GemmP q = p;
  q.ws = nullptr;
  if constexpr (LOOP == 3 || LOOP == 10) { q.alpha = p.alpha * p.scale_a[0] * p.scale_b[0]; q.scale_a = nullptr; q.scale_b = nullptr; }  // dequantisation (per-tensor scales in device memory)
#pragma unroll
  for (int i = 0; i < 4; i++) half_epilogue<EPI>(q, acc[i], m0 + wr * 128 + i * 32, n0 + wc * 64, lane, 0, stg);
  __syncthreads();  // the staging area becomes stage 0 / 1 of the next piece
}

// The next piece of logical workgroup `wg` (it = how many it has taken, u = stream position): problem, tile, k range, fan-in and where the tile's pieces
// publish.  One function for the three schedules so that the kernel has ONE inlined copy of the piece body (four copies spilled 60-280 registers).
struct PieceSel { int i, gtile, tile, kt_begin, kt_end, n_pieces; PieceRed rd; };
__device__ __forceinline__ bool next_piece(const P256& g, int wg, int G, int it, long long& u, long long u_end, PieceSel& o) {
  o.rd.first = 0; o.rd.tile_unit0 = 0; o.rd.rem_slot = nullptr; o.rd.slot0 = nullptr; o.rd.stride = 0; o.rd.n_big = 0;
  if (g.mode == 0) {
    const int q = wg + it * G;
    if (q >= g.piece_begin[g.count]) return false;
    int i = 0;
#pragma unroll
    for (int j = 1; j < P_MAX; j++) i += (j < g.count && q >= g.piece_begin[j]) ? 1 : 0;
    const int tiles = g.tile_begin[i + 1] - g.tile_begin[i];
    const int local = q - g.piece_begin[i];
    const int s = local / tiles, tile = local - s * tiles;   // slice-major
    o.i = i; o.tile = tile; o.gtile = g.tile_begin[i] + tile;
    o.kt_begin = s * g.kts[i]; o.kt_end = min(g.nkt[i], o.kt_begin + g.kts[i]); o.n_pieces = g.split[i];
    o.rd.my_slot = g.slots + (long long)q * P_SLOT_FLOATS;
    o.rd.stride = (long long)tiles * P_SLOT_FLOATS; o.rd.slot0 = o.rd.my_slot - (long long)s * o.rd.stride; o.rd.n_big = o.n_pieces; o.rd.me = s;
    return true;
  }
  if (g.mode == 2) {
    const int n_big = g.big_begin[g.count];
    if (wg < n_big) {  // one full slice
      if (it > 0) return false;
      int i = 0;
#pragma unroll
      for (int j = 1; j < P_MAX; j++) i += (j < g.count && wg >= g.big_begin[j]) ? 1 : 0;
      const int tiles = g.tile_begin[i + 1] - g.tile_begin[i];
      const int local = wg - g.big_begin[i];
      const int s = local / tiles, tile = local - s * tiles;   // slice-major
      o.i = i; o.tile = tile; o.gtile = g.tile_begin[i] + tile;
      o.kt_begin = s * g.kts[i]; o.kt_end = min(g.nkt[i], o.kt_begin + g.kts[i]); o.n_pieces = g.nbig[i] + (g.rem_per[i] > 0 ? 1 : 0);
      o.rd.my_slot = g.slots + (long long)wg * P_SLOT_FLOATS;
      o.rd.stride = (long long)tiles * P_SLOT_FLOATS; o.rd.slot0 = o.rd.my_slot - (long long)s * o.rd.stride; o.rd.n_big = g.nbig[i];
      o.rd.rem_slot = g.slots + (long long)(n_big + o.gtile) * P_SLOT_FLOATS; o.rd.me = s;
      return true;
    }
    const int rc = wg - n_big;  // the remainders of rem_per consecutive tiles, one after the other
    int i = 0;
#pragma unroll
    for (int j = 1; j < P_MAX; j++) i += (j < g.count && rc >= g.rem_begin[j]) ? 1 : 0;
    const int tiles = g.tile_begin[i + 1] - g.tile_begin[i];
    const int tile = (rc - g.rem_begin[i]) * g.rem_per[i] + it;
    if (it >= g.rem_per[i] || tile >= tiles) return false;
    o.i = i; o.tile = tile; o.gtile = g.tile_begin[i] + tile;
    o.kt_begin = g.nbig[i] * g.kts[i]; o.kt_end = g.nkt[i]; o.n_pieces = g.nbig[i] + 1;
    o.rd.my_slot = g.slots + (long long)(n_big + o.gtile) * P_SLOT_FLOATS;
    o.rd.stride = (long long)tiles * P_SLOT_FLOATS; o.rd.slot0 = g.slots + (long long)(g.big_begin[i] + tile) * P_SLOT_FLOATS; o.rd.n_big = g.nbig[i];
    o.rd.rem_slot = o.rd.my_slot; o.rd.me = g.nbig[i];
    return true;
  }
  // stream: contiguous unit range [u, u_end) of problem 0
  if (u >= u_end) return false;
  const int nkt = g.nkt[0];
  const int tile = (int)(u / nkt);
  const long long t0 = (long long)tile * nkt;
  o.i = 0; o.tile = tile; o.gtile = tile;
  o.kt_begin = (int)(u - t0); o.kt_end = (int)min((long long)nkt, u_end - t0);
  const int w_first = stream_owner_of(g, t0), w_last = stream_owner_of(g, t0 + nkt - 1);
  o.n_pieces = w_last - w_first + 1;
  const int which = stream_start(g, wg) >= t0 ? 0 : 1;   // this workgroup's first piece, or a later one (only its last piece can be partial then)
  o.rd.my_slot = g.slots + ((long long)wg * 2 + which) * P_SLOT_FLOATS;
  o.rd.first = w_first; o.rd.tile_unit0 = t0; o.rd.me = wg;
  u = t0 + o.kt_end;
  return true;
}

template <bool A_KMAJ, bool B_KMAJ, int EPI, int LOOP>
__global__ __launch_bounds__(512, 2) void gemm_p256_kernel(P256 g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int G = (int)gridDim.x;
  const int wg = xcd_remap((int)blockIdx.x, G);  // neighbouring logical ids share an XCD (and with it operand panels in L2)
  long long u = 0, u_end = 0;
  if (g.mode == 1) { u = stream_start(g, wg); u_end = stream_start(g, wg + 1); }
  PieceSel o;
  for (int it = 0; next_piece(g, wg, G, it, u, u_end, o); it++) {
    const GemmP p = expand(g.p[o.i]);
    p256_piece<A_KMAJ, B_KMAJ, EPI, LOOP>(g, p, o.gtile, o.tile, o.kt_begin, o.kt_end, o.n_pieces, o.rd, smem);
  }
}

template <bool A_KMAJ, bool B_KMAJ, int EPI, int LOOP>
int launch_p256(const P256& g, int grid, hipStream_t st) {
  static bool attr_set[16] = {};
  auto fn = gemm_p256_kernel<A_KMAJ, B_KMAJ, EPI, LOOP>;
  const hipError_t e = dyn_lds_attr_once(attr_set, reinterpret_cast<const void*>(fn), P_SMEM);
  if (e != hipSuccess) return (int)e;
  launch_any(fn, dim3(grid), dim3(512), (size_t)P_SMEM, st, g);
  return launch_status();
}

int cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0; hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    if (cus < 8) cus = 8;
  }
  return cus;
}

}  // namespace

// Grouped / persistent 256x256 GEMM: up to 12 problems of ONE operand layout and epilogue class in one launch (see the header of this file).
//   schedule 0 "split": balanced k-slices per tile (weight gradients; also whole-K tiles when the tile count fills the chip),
//   schedule 1 "stream": one problem, equal contiguous (tile, k-tile) ranges per workgroup.
// workspace: >= cinema_gemm_p256_workspace_bytes(); its first 64 KiB hold the tile counters and must be ZERO before the first use (the kernel
// leaves them zero); one workspace per stream (concurrent launches must not share one).
CINEMA_API long long cinema_gemm_p256_workspace_bytes(void) { return (long long)P_COUNTER_BYTES + 2LL * cu_count() * P_SLOT_FLOATS * 4; }

static int p256_launch_host(cinema_gemm_args* args, int count, int schedule, void* workspace, long long workspace_bytes, void* stream, bool fp8);
CINEMA_API int cinema_gemm_bf16_p256(cinema_gemm_args* args, int count, int schedule, void* workspace, long long workspace_bytes, void* stream) {
  return p256_launch_host(args, count, schedule, workspace, workspace_bytes, stream, false);
}
// Weight gradients on 8-bit operands (form 3 of the main loop): args[i].a = dY8 [rows = k][lda] bytes, args[i].b = X8 [rows][ldb] bytes (OCP e4m3, row-major
// [token][feature]: a_kmajor = b_kmajor = 0), m / n = feature counts (multiples of 16, lda / ldb multiples of 16 bytes, 16-byte aligned), scale_a / scale_b =
// per-tensor dequantisation scales (device scalars); D fp32 [m][ldd] (+)= scale_a * scale_b * dY8^T X8.  No a_rowsum (bias gradients: cinema_colsum), no other
// epilogue term.  Schedule, workspace and in-launch reduction as cinema_gemm_bf16_p256 (split schedule).
CINEMA_API int cinema_gemm_fp8_wgrad_p256(cinema_gemm_args* args, int count, void* workspace, long long workspace_bytes, void* stream) {
  return p256_launch_host(args, count, 0, workspace, workspace_bytes, stream, true);
}
static int p256_launch_host(cinema_gemm_args* args, int count, int schedule, void* workspace, long long workspace_bytes, void* stream, bool fp8) {
  if (!args || count < 1 || count > P_MAX || !workspace || (((uintptr_t)workspace) & 255)) return CINEMA_ERR_BAD_ARG;
  if (schedule == 1 && count != 1) return CINEMA_ERR_BAD_ARG;
  auto al8 = [](long long v) { return (v & 7) == 0; };
  auto ptr16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  // main-loop form (p256_loop64 / p256_loop32) and with it the k extent of a schedule unit; CINEMA_P256_LOOP=0 selects the plain k-tile loop
  const char* loop_s = getenv("CINEMA_P256_LOOP");   // read per call (the A/B tools toggle it in one process)
  const int loop_env = loop_s ? atoi(loop_s) : 2;
  const int loop_form = fp8 ? 3 : loop_env;
  const int unit_k = fp8 ? 64 : (loop_form ? 32 : BK);
  P256 g;
  g.count = count; g.mode = schedule ? 1 : 0;
  g.tile_begin[0] = 0; g.piece_begin[0] = 0;
  const int ak = args[0].a_kmajor, bk = args[0].b_kmajor;
  if (!ak && bk) return CINEMA_ERR_UNSUPPORTED;
  int epi = -1;
  long long units = 0;
  for (int i = 0; i < count; i++) {
    const cinema_gemm_args* a = &args[i];
    if (!a->a || !a->b || !a->d || a->m <= 0 || a->n <= 0 || a->k <= 0) return CINEMA_ERR_BAD_ARG;
    if (a->a_kmajor != ak || a->b_kmajor != bk) return CINEMA_ERR_UNSUPPORTED;
    if (a->residual_bf16 || a->row_mask || a->conv_taps) return CINEMA_ERR_UNSUPPORTED;
    if (fp8) {
      auto al16 = [](long long v) { return (v & 15) == 0; };
      if (!a->scale_a || !a->scale_b || a->scale_a_rows) return CINEMA_ERR_BAD_ARG;
      if (a->a_kmajor || a->b_kmajor || !a->out_f32 || a->a_rowsum || a->bias || a->residual_f32 || a->gelu_in || a->aux_out || a->act) return CINEMA_ERR_UNSUPPORTED;
      if (!al16(a->m) || !al16(a->n) || !al16(a->lda) || !al16(a->ldb)) return CINEMA_ERR_UNSUPPORTED;
    } else if (a->scale_a || a->scale_b) return CINEMA_ERR_UNSUPPORTED;
    if (a->accumulate && !a->out_f32) return CINEMA_ERR_BAD_ARG;
    if (a->accumulate && a->residual_f32) return CINEMA_ERR_UNSUPPORTED;
    bool ok = al8(a->lda) && al8(a->ldb) && al8(a->ldd) && al8(a->n) && ptr16(a->a) && ptr16(a->b) && ptr16(a->d);
    ok = ok && (a->a_kmajor ? al8(a->k) : al8(a->m)) && (a->b_kmajor ? al8(a->k) : true);
    ok = ok && (!a->bias || ptr16(a->bias)) && (!a->residual_f32 || (al8(a->ld_res) && ptr16(a->residual_f32)));
    ok = ok && (!a->gelu_in || (al8(a->ld_gelu) && ptr16(a->gelu_in))) && (!a->aux_out || (al8(a->ld_aux) && ptr16(a->aux_out)));
    if (!ok) return CINEMA_ERR_UNSUPPORTED;
    if (a->a_rowsum && a->a_kmajor) return CINEMA_ERR_UNSUPPORTED;
    SlimP& p = g.p[i];
    p.a = (const bf16_t*)a->a; p.b = (const bf16_t*)a->b; p.d = a->d;
    p.m = a->m; p.n = a->n; p.k = a->k; p.lda = a->lda; p.ldb = a->ldb; p.ldd = a->ldd;
    p.alpha = a->alpha;
    p.bias = a->bias; p.res_f32 = a->residual_f32; p.ld_res = a->ld_res;
    p.gelu_in = a->gelu_in; p.ld_gelu = a->ld_gelu; p.aux_out = a->aux_out; p.ld_aux = a->ld_aux;
    p.act = a->act; p.gelu_deriv = a->gelu_deriv; p.out_f32 = a->out_f32; p.a_rowsum = a->a_rowsum;
    p.scale_a = fp8 ? a->scale_a : nullptr; p.scale_b = fp8 ? a->scale_b : nullptr;
    if (a->accumulate) { p.res_f32 = (const float*)a->d; p.ld_res = a->ldd; }  // one owner per element: plain read-modify-write
    int e;
    if (!p.out_f32 && !p.res_f32 && !p.gelu_in && p.act == 0 && !p.aux_out) e = EPI_BF16;
    else if (!p.out_f32 && !p.res_f32 && !p.gelu_in && p.act == 1) e = EPI_BF16_GELU;
    else if (!p.out_f32 && !p.res_f32 && p.gelu_in && p.act == 0 && !p.aux_out && !p.bias) e = EPI_BF16_GELU_GRAD;
    else if (p.out_f32 && !p.gelu_in && p.act == 0 && !p.aux_out) e = EPI_F32;
    else return CINEMA_ERR_UNSUPPORTED;
    if (epi >= 0 && e != epi) return CINEMA_ERR_UNSUPPORTED;
    epi = e;
    const int tiles = ((a->m + P_TILE - 1) / P_TILE) * ((a->n + P_TILE - 1) / P_TILE);
    g.nkt[i] = (a->k + unit_k - 1) / unit_k;
    g.tile_begin[i + 1] = g.tile_begin[i] + tiles;
    units += (long long)tiles * g.nkt[i];
  }
  if (g.tile_begin[count] * 8 > P_COUNTER_BYTES) return CINEMA_ERR_UNSUPPORTED;
  const int G = cu_count();
  int grid;
  long long slots;
  if (g.mode == 0) {
    // smallest slice length x (in k-tiles) whose piece count fits one round of the chip; more tiles than CUs: whole-K tiles, several rounds
    const int total_tiles = g.tile_begin[count];
    long long x = (units + G - 1) / G;
    if (x < 256 / unit_k) x = 256 / unit_k;
    int maxk = 0;
    for (int i = 0; i < count; i++) maxk = g.nkt[i] > maxk ? g.nkt[i] : maxk;
    if (args[0].split_k == 1 || total_tiles >= G) x = maxk;
    for (;; x++) {
      long long pieces = 0;
      for (int i = 0; i < count; i++) pieces += (long long)(g.tile_begin[i + 1] - g.tile_begin[i]) * ((g.nkt[i] + x - 1) / x);
      if (pieces <= G || x >= maxk) break;
    }
    for (int i = 0; i < count; i++) {
      const int sp = (int)((g.nkt[i] + x - 1) / x);
      g.kts[i] = (g.nkt[i] + sp - 1) / sp;
      g.split[i] = (g.nkt[i] + g.kts[i] - 1) / g.kts[i];
      g.piece_begin[i + 1] = g.piece_begin[i] + (g.tile_begin[i + 1] - g.tile_begin[i]) * g.split[i];
    }
    for (int i = count; i < P_MAX; i++) { g.piece_begin[i + 1] = g.piece_begin[count]; g.tile_begin[i + 1] = g.tile_begin[count]; g.split[i] = 1; g.kts[i] = 1; g.nkt[i] = 1; }
    grid = g.piece_begin[count] < G ? g.piece_begin[count] : G;
    slots = g.piece_begin[count];
    bool any_split = false;
    int longest = 0;
    for (int i = 0; i < count; i++) { any_split = any_split || g.split[i] > 1; longest = g.kts[i] > longest ? g.kts[i] : longest; }
    if (!any_split) slots = 0;
    g.units = units; g.per = 0; g.rem = 0;
    for (int i = 0; i < P_MAX; i++) { g.nbig[i] = 0; g.rem_per[i] = 0; g.big_begin[i + 1] = 0; g.rem_begin[i + 1] = 0; }
    g.big_begin[0] = 0; g.rem_begin[0] = 0;
    // split + remainder: slices of one length L >= units / CUs and the rests packed several to a workgroup; taken (CINEMA_P256_REMAINDER=1) when its longest
    // workgroup is >= 4 % shorter than the equal slices'.  OFF by default - measured on the encoder block's weight gradients (216 equal slices of 172 units vs
    // 216 x 147 + 108 rests of 49): 210.7 -> 206.5 us only (the launch is not bound by its longest workgroup: with 252 instead of 216 busy CUs the per-CU rate
    // drops), for 9 % more HBM traffic (three partial tiles per output tile instead of two).
    const int rem_env = getenv("CINEMA_P256_REMAINDER") ? atoi(getenv("CINEMA_P256_REMAINDER")) : 0;  // read per call (tests toggle it)
    const int min_rest = 256 / unit_k;  // a shorter rest is not worth a piece of its own: the slices of that problem are stretched instead
    if (rem_env && any_split && args[0].split_k != 1 && total_tiles < G) {
      for (long long L = (units + G - 1) / G; L < longest; L++) {
        int nb[P_MAX], len[P_MAX], per[P_MAX], used = 0, worst = 0;
        for (int i = 0; i < count; i++) {
          const int tiles = g.tile_begin[i + 1] - g.tile_begin[i];
          nb[i] = (int)(g.nkt[i] / L); len[i] = (int)L; per[i] = 0;
          int rest = g.nkt[i] - nb[i] * (int)L;
          if (nb[i] > 0 && rest < min_rest) { len[i] = (g.nkt[i] + nb[i] - 1) / nb[i]; nb[i] = (g.nkt[i] + len[i] - 1) / len[i]; rest = 0; }
          if (rest > 0) { per[i] = (int)(L / rest) > 0 ? (int)(L / rest) : 1; used += (tiles + per[i] - 1) / per[i]; worst = per[i] * rest > worst ? per[i] * rest : worst; }
          used += tiles * nb[i];
          if (nb[i] > 0 && len[i] > worst) worst = len[i];
        }
        if (used > G) continue;
        if (worst * 100 > longest * 96) break;  // the first fit is the shortest; not enough of a gain
        for (int i = 0; i < count; i++) {
          const int tiles = g.tile_begin[i + 1] - g.tile_begin[i];
          g.nbig[i] = nb[i]; g.kts[i] = len[i]; g.rem_per[i] = per[i];
          g.big_begin[i + 1] = g.big_begin[i] + tiles * nb[i];
          g.rem_begin[i + 1] = g.rem_begin[i] + (per[i] > 0 ? (tiles + per[i] - 1) / per[i] : 0);
        }
        for (int i = count; i < P_MAX; i++) { g.big_begin[i + 1] = g.big_begin[count]; g.rem_begin[i + 1] = g.rem_begin[count]; }
        g.mode = 2;
        grid = g.big_begin[count] + g.rem_begin[count];
        slots = g.big_begin[count] + total_tiles;
        break;
      }
    }
  } else {
    grid = units < G ? (int)units : G;
    g.units = units; g.per = (int)(units / grid); g.rem = (int)(units % grid);
    for (int i = 0; i < P_MAX; i++) { g.split[i] = 1; g.kts[i] = g.nkt[0]; }
    for (int i = 1; i < P_MAX; i++) { g.piece_begin[i + 1] = 0; g.tile_begin[i + 1] = g.tile_begin[1]; g.nkt[i] = 1; }
    g.piece_begin[1] = 0;
    slots = 2LL * grid;
  }
  if (workspace_bytes < P_COUNTER_BYTES + slots * P_SLOT_FLOATS * 4LL) return CINEMA_ERR_BAD_ARG;
  g.counters = (unsigned*)workspace;
  g.error = g.counters + P_COUNTER_BYTES / 4 - 1;   // last counter word (tile ids never reach it: checked above)
  g.slots = (float*)((char*)workspace + P_COUNTER_BYTES);
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < count; i++) args[i].kernel_used = (fp8 ? 4096 : 2048) + (ak && bk ? 1 : (ak ? 2 : 3)) + 8 * epi;
  if (fp8) return loop_env >= 2 ? launch_p256<false, false, EPI_F32, 10>(g, grid, st) : launch_p256<false, false, EPI_F32, 3>(g, grid, st);
#define P256_LAYOUT(E)                                                        \
  do {                                                                        \
    if (loop_form >= 2) {                                                     \
      if (ak && bk) return launch_p256<true, true, E, 2>(g, grid, st);        \
      if (ak && !bk) return launch_p256<true, false, E, 2>(g, grid, st);      \
      return launch_p256<false, false, E, 2>(g, grid, st);                    \
    }                                                                         \
    if (loop_form) {                                                          \
      if (ak && bk) return launch_p256<true, true, E, 1>(g, grid, st);        \
      if (ak && !bk) return launch_p256<true, false, E, 1>(g, grid, st);      \
      return launch_p256<false, false, E, 1>(g, grid, st);                    \
    }                                                                         \
    if (ak && bk) return launch_p256<true, true, E, 0>(g, grid, st);          \
    if (ak && !bk) return launch_p256<true, false, E, 0>(g, grid, st);        \
    return launch_p256<false, false, E, 0>(g, grid, st);                      \
  } while (0)
  switch (epi) {
    case EPI_BF16: P256_LAYOUT(EPI_BF16);
    case EPI_BF16_GELU: P256_LAYOUT(EPI_BF16_GELU);
    case EPI_BF16_GELU_GRAD: P256_LAYOUT(EPI_BF16_GELU_GRAD);
    default: P256_LAYOUT(EPI_F32);
  }
#undef P256_LAYOUT
}
