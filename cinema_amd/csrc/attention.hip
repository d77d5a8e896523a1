// Flash attention forward / backward for gfx950 (no mask, no dropout), head_dim 32 / 64 on MFMA 32x32x16 bf16.
//
// Layout trick used by all three MFMA kernels: scores are produced TRANSPOSED so that the softmax row statistics are
// lane-local, and the probability accumulator registers are re-used *as is* as the k-slot operand of the second
// MFMA (the MFMA only needs both operands to agree on which reduction index sits in which k-slot):
//   acc reg r of lane l  <->  row (r&3)+8*(r>>2)+4*(l>>5), col l&31;   regs 8s..8s+7 = the 8 k-slots of MFMA step s.
// The operand that is strided along the reduction index (V^T, K^T, dO^T, Q^T) is read from its row-major LDS tile
// with ds_read_b64_tr_b16.  Tiles are staged through registers into XOR-swizzled LDS, double buffered.
#include <cstdlib>
#include "common.cuh"
#include <initializer_list>
#include <type_traits>
#include "../../include/cinema_hip.h"

namespace {

struct AttnP {
  const bf16_t* q; int ldq; const bf16_t* k; int ldk; const bf16_t* v; int ldv;
  bf16_t* o; int ldo; float* lse;
  const bf16_t* d_o; int lddo; float* delta;
  bf16_t* o_lo;  // optional second half of the output, same addressing as o: bf16(O - float(bf16(O))).  delta = rowsum(dO (o + o_lo)) then carries O to 2^-17 (see attn_bwd_dq_mfma)
  bf16_t* dq; int lddq; bf16_t* dk; int lddk; bf16_t* dv; int lddv;
  int b, h, tq, tk, hd;
  float scale;   // softmax scale (head_dim^-0.5)
  float c2;      // scale * log2(e)
};

typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
// Drain every outstanding vector-memory operation of the wave (the LDS-DMA pieces of a tile and ordinary loads alike).  Issued through the BUILTIN, not inline asm:
// hipcc's wait-count pass then KNOWS that nothing is pending behind this point.  With an asm wait it did not, and in every loop that keeps global loads in
// registers across iterations (Q / dO / K / V row fragments) it guarded their first use in each iteration with s_waitcnt vmcnt(3..0) - which, the counter being
// in order, waited for the LDS-DMA of the NEXT tile issued a few instructions earlier: the double buffering of every attention kernel was serialised behind a
// global round trip per tile (round 5 finding, from the ISA of the new one-pass kernel: 2.3 us per tile).  simm16 = vmcnt 0, expcnt 7, lgkmcnt 15.
#define WAIT_VM0() do { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); } while (0)
constexpr float NEG_INF = -1e30f;
__device__ __forceinline__ float bf_lo32(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi32(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// ---- stage a [ROWS][HD] bf16 tile (rows = tokens row0.., this head's HD columns) global -> regs -> swizzled LDS
template <int HD, int ROWS>
struct TileStage {
  static constexpr int CPR = HD / 8;                 // 16-byte chunks per row
  static constexpr int NCH = ROWS * CPR;
  static constexpr int PASSES = (NCH + 255) / 256;
  static constexpr int BYTES = ROWS * HD * 2;
  uint4 r[PASSES];
  __device__ __forceinline__ void load(const bf16_t* base, int ld, int row0, int nrows, int tid) {
#pragma unroll
    for (int pss = 0; pss < PASSES; pss++) {
      const int cid = pss * 256 + tid;
      const int row = cid / CPR, c = cid % CPR;
      // rows past the end re-read the last valid row (no divergent branch around the load): every consumer masks those
      // positions (score -> -inf / probability -> 0), so only finiteness of the data matters
      const int rr = min(row0 + row, nrows - 1);
      r[pss] = *reinterpret_cast<const uint4*>(base + (size_t)rr * ld + c * 8);
    }
  }
  static_assert(NCH % 256 == 0, "tile chunks must be a multiple of the block size");
  // LDS-DMA variant: no staging registers (the register-staged tiles were spilled to scratch inside the key/query loop of the
  // HD = 64 kernels), asynchronous; 1 KiB lane-linear pieces, the XOR swizzle of swz_off<HD*2>() applied to the SOURCE chunk.
  // Callers drain with s_waitcnt vmcnt(0) before the barrier that publishes the tile.
  static __device__ __forceinline__ void glds(char* lds, const bf16_t* base, int ld, int row0, int nrows, int lane, int wave) {
    constexpr int RB = HD * 2, RPP = 1024 / RB, PIECES = BYTES / 1024;  // bytes per row, rows per piece
    static_assert(PIECES % 4 == 0, "pieces are dealt to the block's 4 waves");
    const uint32_t a0 = __builtin_amdgcn_readfirstlane(lds_address(lds));
#pragma unroll
    for (int pss = 0; pss < PIECES / 4; pss++) {
      const int blk = pss * 4 + wave;
      const int row = blk * RPP + lane / CPR;
      const int c = (lane % CPR) ^ ((row / (256 / RB)) & (CPR - 1));
      const int rr = min(row0 + row, nrows - 1);
      glds16(__builtin_amdgcn_readfirstlane(a0 + blk * 1024), base + (size_t)rr * ld + c * 8);
    }
  }
  __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
    for (int pss = 0; pss < PASSES; pss++) {
      const int cid = pss * 256 + tid;
      const int row = cid / CPR, c = cid % CPR;
      if (cid < NCH) *reinterpret_cast<uint4*>(lds + swz_off<HD * 2>(row, c)) = r[pss];
    }
  }
};

// K-major fragment: lane l -> tile row base+(l&31), reduction (head-dim) elements 16*ks + 8*(l>>5) .. +7
template <int HD>
__device__ __forceinline__ short8v frag_km(const char* lds, int base, int ks, int lane) {
  return *reinterpret_cast<const short8v*>(lds + swz_off<HD * 2>(base + (lane & 31), ks * 2 + (lane >> 5)));
}
// Transposed fragment: lane l -> tile COLUMN dcol0 + (l&31) (a head-dim index), reduction = tile rows
// rbase + 4*(l>>5) + {0..3} and + 8 + {0..3}  (matching acc-register k-slots, see header comment).
template <int HD>
__device__ __forceinline__ short8v frag_tr(const char* lds, int rbase, int dcol0, int lane) {
  const int q4 = lane >> 4, t = lane & 15;
  const int col = dcol0 + 16 * (q4 & 1) + 4 * (t & 3);
  const int row = rbase + 4 * (q4 >> 1) + (t >> 2);
  const short4v lo = lds_tr16_b64(lds + swz_off<HD * 2>(row, col >> 3) + (col & 7) * 2);
  const short4v hi = lds_tr16_b64(lds + swz_off<HD * 2>(row + 8, col >> 3) + (col & 7) * 2);
  short8v out;
  out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; out[3] = lo[3];
  out[4] = hi[0]; out[5] = hi[1]; out[6] = hi[2]; out[7] = hi[3];
  return out;
}
// pack accumulator regs 8s..8s+7 to the bf16 k-slot operand
__device__ __forceinline__ short8v pack_slots(const float16v& a, int s) {
  union { uint32_t u[4]; short8v v; } x;
#pragma unroll
  for (int i = 0; i < 4; i++) x.u[i] = pack_bf2(a[8 * s + 2 * i], a[8 * s + 2 * i + 1]);
  return x.v;
}
template <int HD>
__device__ __forceinline__ void load_row_frags(short8v (&f)[HD / 16], const bf16_t* rowptr, int lane) {
#pragma unroll
  for (int ks = 0; ks < HD / 16; ks++) f[ks] = *reinterpret_cast<const short8v*>(rowptr + ks * 16 + 8 * (lane >> 5));
}
__device__ __forceinline__ void zero16(float16v& a) {
#pragma unroll
  for (int r = 0; r < 16; r++) a[r] = 0.f;
}

// ================================================================================================
// forward: block = 4 waves x 32 queries; K/V tiles of 64 keys
// ================================================================================================
// (tile, head, batch) of this workgroup: the linear dispatch id goes through xcd_remap() so that the tiles of one
// (batch, head) pair - which share K/V (forward, dQ) or Q/dO (dK/dV) - run on ONE XCD and meet in its L2 (PMC: the plain
// grid fetched the shared operands once per XCD, 4x the operand bytes on the decoder shapes).
struct BlockCoord { int t, h, b; };
__device__ __forceinline__ BlockCoord block_coord() {
  const int gx = gridDim.x, gy = gridDim.y;
  const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int l = xcd_remap(lin, gx * gy * (int)gridDim.z);
  BlockCoord c;
  c.t = l % gx;
  const int r = l / gx;
  c.h = r % gy;
  c.b = r / gy;
  return c;
}

// V2 (round 4): the softmax denominator comes out of the matrix pipe (one more MFMA per 16 keys with an all-ones operand: the row sums of the bf16 probabilities
// the P V product uses, instead of 32 dependent v_add_f32 per tile) and the running maximum is raised lazily (only when some query's tile maximum exceeds it by
// more than 2^6: the accumulators are then rescaled, otherwise neither the rescale multiplies nor the alpha exponential run) - the kernel is bound by VALU +
// v_exp_f32 issue, the matrix pipe has the slack.
template <int HD, bool V2>
__global__ __launch_bounds__(256, 4) void attn_fwd_mfma(AttnP p) {
  using Stage = TileStage<HD, 64>;
  constexpr int TB = Stage::BYTES;
  __shared__ __attribute__((aligned(16))) char smem[4 * TB];  // [2 stages][K | V]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
  const BlockCoord bc = block_coord();
  const int b = bc.b, h = bc.h;
  const int qrow = bc.t * 128 + wave * 32 + (lane & 31);
  const int qc = qrow < p.tq ? qrow : p.tq - 1;
  const bf16_t* kbase = p.k + (size_t)b * p.tk * p.ldk + h * HD;
  const bf16_t* vbase = p.v + (size_t)b * p.tk * p.ldv + h * HD;

  short8v qf[HD / 16];
  load_row_frags<HD>(qf, p.q + ((size_t)b * p.tq + qc) * p.ldq + h * HD, lane);

  float16v o[HD / 32];
#pragma unroll
  for (int i = 0; i < HD / 32; i++) zero16(o[i]);
  float m_run = NEG_INF, l_run = 0.f;
  constexpr bool SUM_MFMA = V2 && HD == 32;  // (at head_dim 64 the 16 extra accumulator registers do not fit the 128-register budget of 4 waves per SIMD: plain sums there)
  float16v lacc;  // SUM_MFMA: every row of this accumulator holds the running denominator of the lane's query
  zero16(lacc);
  short8v ones;
#pragma unroll
  for (int e = 0; e < 8; e++) ones[e] = (short)0x3F80;  // bf16 1.0

  const int nkt = (p.tk + 63) / 64;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  Stage::glds(smem, kbase, p.ldk, 0, p.tk, lane, wave_u);
  Stage::glds(smem + TB, vbase, p.ldv, 0, p.tk, lane, wave_u);
  WAIT_VM0();
  __syncthreads();

  for (int kt = 0; kt < nkt; kt++) {
    const char* ks_ = smem + (kt & 1) * 2 * TB;
    const char* vs_ = ks_ + TB;
    const bool more = kt + 1 < nkt;
    if (more) {  // the other stage was last read before the barrier that ended the previous iteration
      char* nk = smem + ((kt + 1) & 1) * 2 * TB;
      Stage::glds(nk, kbase, p.ldk, (kt + 1) * 64, p.tk, lane, wave_u);
      Stage::glds(nk + TB, vbase, p.ldv, (kt + 1) * 64, p.tk, lane, wave_u);
    }
    float16v s[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      zero16(s[u]);
#pragma unroll
      for (int ks = 0; ks < HD / 16; ks++) s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km<HD>(ks_, 32 * u, ks, lane), qf[ks], s[u], 0, 0, 0);
    }
    // online softmax (base-2 domain); this lane owns query column lane&31, keys spread over regs (+ partner lane^32)
    float mx = NEG_INF;
    if (kt == nkt - 1 && (p.tk & 63)) {  // only the ragged last tile needs the key-range mask (wave-uniform branch)
      const int key0 = kt * 64 + 4 * g;
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int key = key0 + 32 * u + (r & 3) + 8 * (r >> 2);
          if (key >= p.tk) s[u][r] = NEG_INF;
        }
    }
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[u][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if constexpr (V2) {
      const float mxs = mx * p.c2;  // tile maximum in the scaled base-2 domain (c2 > 0)
      if (__builtin_amdgcn_ballot_w64(mxs > m_run + 6.0f) != 0ull) {  // wave-uniform: somebody's maximum moved by more than 2^6 (always true in the first tile)
        const float m_new = fmaxf(m_run, mxs);
        const float alpha = fast_exp2(m_run - m_new);
        m_run = m_new;
        if (SUM_MFMA) lacc[0] *= alpha; else l_run *= alpha;
#pragma unroll
        for (int i = 0; i < HD / 32; i++)
#pragma unroll
          for (int r = 0; r < 16; r++) o[i][r] *= alpha;
      }
      float ps = 0.f;
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          s[u][r] = fast_exp2(fmaf(s[u][r], p.c2, -m_run));  // <= 2^6: no overflow anywhere downstream
          if (!SUM_MFMA) ps += s[u][r];
        }
      if (!SUM_MFMA) l_run += ps;
    } else {
      const float m_new = fmaxf(m_run, mx * p.c2);  // running max in the scaled base-2 domain (c2 > 0)
      const float alpha = fast_exp2(m_run - m_new);
      float ps = 0.f;
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const float e = fast_exp2(fmaf(s[u][r], p.c2, -m_new));  // one FMA + one v_exp_f32 per score
          s[u][r] = e;
          ps += e;
        }
      l_run = l_run * alpha + ps;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < HD / 32; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[i][r] *= alpha;
    }
    // O^T[d][q] += V^T[d][key] * P^T[key][q]
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int st = 0; st < 2; st++) {
        const short8v pf = pack_slots(s[u], st);
#pragma unroll
        for (int dt = 0; dt < HD / 32; dt++)
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<HD>(vs_, 32 * u + 16 * st, 32 * dt, lane), pf, o[dt], 0, 0, 0);
        if constexpr (SUM_MFMA) lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lacc, 0, 0, 0);  // row sums of the same bf16 probabilities
      }
    WAIT_VM0();  // the next tile's DMA has landed
    __syncthreads();
  }
  const float l_tot = SUM_MFMA ? lacc[0] : l_run + __shfl_xor(l_run, 32, 64);  // (the MFMA reduces over all 16 k-slots: both lane halves hold the full sum)
  if (qrow < p.tq) {
    const float inv = 1.f / l_tot;
    bf16_t* op = p.o + ((size_t)b * p.tq + qrow) * p.ldo + h * HD;
#pragma unroll
    for (int dt = 0; dt < HD / 32; dt++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint2 pk;
        const float v0 = o[dt][4 * j] * inv, v1 = o[dt][4 * j + 1] * inv, v2 = o[dt][4 * j + 2] * inv, v3 = o[dt][4 * j + 3] * inv;
        pk.x = pack_bf2(v0, v1);
        pk.y = pack_bf2(v2, v3);
        *reinterpret_cast<uint2*>(op + dt * 32 + 8 * j + 4 * g) = pk;
        if (p.o_lo) {  // what the bf16 rounding took away, as a second bf16 (training: the backward pass forms delta from both halves)
          uint2 lo;
          lo.x = pack_bf2(v0 - bf_lo32(pk.x), v1 - bf_hi32(pk.x));
          lo.y = pack_bf2(v2 - bf_lo32(pk.y), v3 - bf_hi32(pk.y));
          *reinterpret_cast<uint2*>(p.o_lo + (op - p.o) + dt * 32 + 8 * j + 4 * g) = lo;
        }
      }
    if (g == 0 && p.lse) p.lse[((size_t)b * p.h + h) * p.tq + qrow] = m_run + log2f(l_tot);
  }
}

// ================================================================================================
// backward, dQ: block = 4 waves x 32 queries; loop over K/V tiles of 64 keys
// ================================================================================================
template <int HD>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_mfma(AttnP p) {
  using Stage = TileStage<HD, 64>;
  constexpr int TB = Stage::BYTES;
  __shared__ __attribute__((aligned(16))) char smem[4 * TB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
  const BlockCoord bc = block_coord();
  const int b = bc.b, h = bc.h;
  const int qrow = bc.t * 128 + wave * 32 + (lane & 31);
  const int qc = qrow < p.tq ? qrow : p.tq - 1;
  const bool active = bc.t * 128 + wave * 32 < p.tq;  // wave-uniform: a wave past the last query only helps with the loads
  const bf16_t* kbase = p.k + (size_t)b * p.tk * p.ldk + h * HD;
  const bf16_t* vbase = p.v + (size_t)b * p.tk * p.ldv + h * HD;
  short8v qf[HD / 16], dof[HD / 16];
  load_row_frags<HD>(qf, p.q + ((size_t)b * p.tq + qc) * p.ldq + h * HD, lane);
  load_row_frags<HD>(dof, p.d_o + ((size_t)b * p.tq + qc) * p.lddo + h * HD, lane);
  const size_t sidx = ((size_t)b * p.h + h) * p.tq + qc;
  const float lse = p.lse[sidx];
  // delta = rowsum(dO * O), computed here from the row fragments (the lane pair l, l^32 holds the whole row) and stored for the dK/dV kernel
  // that runs next: this used to be a launch of its own that read O and dO once more
  float dl = 0.f;
  {
    short8v of[HD / 16];
    load_row_frags<HD>(of, p.o + ((size_t)b * p.tq + qc) * p.ldo + h * HD, lane);
#pragma unroll
    for (int ks = 0; ks < HD / 16; ks++)
#pragma unroll
      for (int e = 0; e < 8; e++)
        dl = fmaf(__uint_as_float((uint32_t)(uint16_t)of[ks][e] << 16), __uint_as_float((uint32_t)(uint16_t)dof[ks][e] << 16), dl);
    if (p.o_lo) {
      // delta from the STORED bf16 O alone carries a relative error of 2^-9 per element; it shifts every dS of the query by P eps, i.e. dQ by eps * sum_j P_ij K_j.
      // In the late encoder blocks the keys are a large common vector plus small differences (|mean key| / |key - mean| = 5.4 at ViT-Large block 22) and the
      // softmax is nearly uniform: that error was 13 % of dQ and 5 % of attn.q.weight's gradient, while the bf16 rounding of P and dS costs 0.25 %
      // (tools/attn_dq_error.py, profiles/r06_i_attn_dq_error.txt).  With the second half of O the error of delta is 2^-17.
      load_row_frags<HD>(of, p.o_lo + ((size_t)b * p.tq + qc) * p.ldo + h * HD, lane);
#pragma unroll
      for (int ks = 0; ks < HD / 16; ks++)
#pragma unroll
        for (int e = 0; e < 8; e++)
          dl = fmaf(__uint_as_float((uint32_t)(uint16_t)of[ks][e] << 16), __uint_as_float((uint32_t)(uint16_t)dof[ks][e] << 16), dl);
    }
    dl += __shfl_xor(dl, 32, 64);
    if (g == 0 && qrow < p.tq) p.delta[sidx] = dl;
  }

  float16v dq[HD / 32];
#pragma unroll
  for (int i = 0; i < HD / 32; i++) zero16(dq[i]);

  const int nkt = (p.tk + 63) / 64;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  Stage::glds(smem, kbase, p.ldk, 0, p.tk, lane, wave_u);
  Stage::glds(smem + TB, vbase, p.ldv, 0, p.tk, lane, wave_u);
  WAIT_VM0();
  __syncthreads();
  for (int kt = 0; kt < nkt; kt++) {
    const char* ks_ = smem + (kt & 1) * 2 * TB;
    const char* vs_ = ks_ + TB;
    const bool more = kt + 1 < nkt;
    if (more) {  // the other stage was last read before the barrier that ended the previous iteration
      char* nk = smem + ((kt + 1) & 1) * 2 * TB;
      Stage::glds(nk, kbase, p.ldk, (kt + 1) * 64, p.tk, lane, wave_u);
      Stage::glds(nk + TB, vbase, p.ldv, (kt + 1) * 64, p.tk, lane, wave_u);
    }
    const int key0 = kt * 64 + 4 * g;
    // idle wave / empty 32-key half of the last tile: wave-uniform trip count (kept rolled: unrolled with the guards inside, the hd=64
    // kernel went from 130 to 168 registers and spilled)
    const int nu = !active ? 0 : (kt * 64 + 32 < p.tk ? 2 : 1);
#pragma unroll 1
    for (int u = 0; u < nu; u++) {
      float16v s, dp;
      zero16(s); zero16(dp);
#pragma unroll
      for (int ks = 0; ks < HD / 16; ks++) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km<HD>(ks_, 32 * u, ks, lane), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km<HD>(vs_, 32 * u, ks, lane), dof[ks], dp, 0, 0, 0);
      }
      if (kt == nkt - 1 && (p.tk & 63)) {  // only the ragged last tile needs the key-range test (wave-uniform branch)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int key = key0 + 32 * u + (r & 3) + 8 * (r >> 2);
          const float pr = key < p.tk ? fast_exp2(fmaf(s[r], p.c2, -lse)) : 0.f;
          s[r] = pr * (dp[r] - dl);  // dS^T
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] = fast_exp2(fmaf(s[r], p.c2, -lse)) * (dp[r] - dl);
      }
#pragma unroll
      for (int st = 0; st < 2; st++) {
        const short8v dsf = pack_slots(s, st);
#pragma unroll
        for (int dt = 0; dt < HD / 32; dt++)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<HD>(ks_, 32 * u + 16 * st, 32 * dt, lane), dsf, dq[dt], 0, 0, 0);
      }
    }
    WAIT_VM0();  // the next tile's DMA has landed
    __syncthreads();
  }
  if (qrow < p.tq) {
    bf16_t* op = p.dq + ((size_t)b * p.tq + qrow) * p.lddq + h * HD;
#pragma unroll
    for (int dt = 0; dt < HD / 32; dt++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint2 pk;
        pk.x = pack_bf2(dq[dt][4 * j] * p.scale, dq[dt][4 * j + 1] * p.scale);
        pk.y = pack_bf2(dq[dt][4 * j + 2] * p.scale, dq[dt][4 * j + 3] * p.scale);
        *reinterpret_cast<uint2*>(op + dt * 32 + 8 * j + 4 * g) = pk;
      }
  }
}

// ================================================================================================
// backward, dK/dV: block = 4 waves x 32 keys; loop over Q/dO tiles of 64 queries
// ================================================================================================
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_mfma(AttnP p) {
  using Stage = TileStage<HD, 64>;
  constexpr int TB = Stage::BYTES;
  constexpr int ST = 2 * TB + 512;  // Q | dO | lse[64] | delta[64]
  __shared__ __attribute__((aligned(16))) char smem[2 * ST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
  const BlockCoord bc = block_coord();
  const int b = bc.b, h = bc.h;
  const int krow = bc.t * 128 + wave * 32 + (lane & 31);
  const int kc = krow < p.tk ? krow : p.tk - 1;
  const bool active = bc.t * 128 + wave * 32 < p.tk;  // wave-uniform: a wave past the last key only helps with the loads
  const bf16_t* qbase = p.q + (size_t)b * p.tq * p.ldq + h * HD;
  const bf16_t* dobase = p.d_o + (size_t)b * p.tq * p.lddo + h * HD;
  const float* lsebase = p.lse + ((size_t)b * p.h + h) * p.tq;
  const float* dlbase = p.delta + ((size_t)b * p.h + h) * p.tq;
  short8v kf[HD / 16], vf[HD / 16];
  load_row_frags<HD>(kf, p.k + ((size_t)b * p.tk + kc) * p.ldk + h * HD, lane);
  load_row_frags<HD>(vf, p.v + ((size_t)b * p.tk + kc) * p.ldv + h * HD, lane);

  float16v dk[HD / 32], dv[HD / 32];
#pragma unroll
  for (int i = 0; i < HD / 32; i++) { zero16(dk[i]); zero16(dv[i]); }

  const int nqt = (p.tq + 63) / 64;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  float st_l = 0.f;  // threads 0..63: lse, 64..127: delta
  auto load_stats = [&](int q0) {
    if (tid < 128) {
      const int qi = q0 + (tid & 63);
      st_l = qi < p.tq ? (tid < 64 ? lsebase[qi] : dlbase[qi]) : (tid < 64 ? 1e30f : 0.f);  // lse=+big -> p=0 for padded queries
    }
  };
  auto store_stats = [&](char* base) {
    if (tid < 128) reinterpret_cast<float*>(base + 2 * TB)[tid] = st_l;
  };
  Stage::glds(smem, qbase, p.ldq, 0, p.tq, lane, wave_u);
  Stage::glds(smem + TB, dobase, p.lddo, 0, p.tq, lane, wave_u);
  load_stats(0);
  store_stats(smem);
  WAIT_VM0();
  __syncthreads();
  for (int qt = 0; qt < nqt; qt++) {
    const char* qs_ = smem + (qt & 1) * ST;
    const char* dos_ = qs_ + TB;
    const float* stats = reinterpret_cast<const float*>(qs_ + 2 * TB);
    const bool more = qt + 1 < nqt;
    if (more) {
      char* nb = smem + ((qt + 1) & 1) * ST;
      Stage::glds(nb, qbase, p.ldq, (qt + 1) * 64, p.tq, lane, wave_u);
      Stage::glds(nb + TB, dobase, p.lddo, (qt + 1) * 64, p.tq, lane, wave_u);
      load_stats((qt + 1) * 64);
    }
    const int nu = !active ? 0 : (qt * 64 + 32 < p.tq ? 2 : 1);  // idle wave / empty 32-query half of the last tile (wave-uniform)
#pragma unroll 1
    for (int u = 0; u < nu; u++) {
      float16v s, dp;
      zero16(s); zero16(dp);
#pragma unroll
      for (int ks = 0; ks < HD / 16; ks++) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km<HD>(qs_, 32 * u, ks, lane), kf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_km<HD>(dos_, 32 * u, ks, lane), vf[ks], dp, 0, 0, 0);
      }
      // acc reg r <-> query 32u + (r&3) + 8*(r>>2) + 4g of this tile; column = this lane's key
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float4 l4 = *reinterpret_cast<const float4*>(stats + 32 * u + 8 * j + 4 * g);
        const float4 d4 = *reinterpret_cast<const float4*>(stats + 64 + 32 * u + 8 * j + 4 * g);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const float pr = fast_exp2(fmaf(s[4 * j + i], p.c2, -lv[i]));
          s[4 * j + i] = pr;                             // P
          dp[4 * j + i] = pr * (dp[4 * j + i] - dv4[i]);  // dS
        }
      }
#pragma unroll
      for (int st = 0; st < 2; st++) {
        const short8v pf = pack_slots(s, st);
        const short8v dsf = pack_slots(dp, st);
#pragma unroll
        for (int dt = 0; dt < HD / 32; dt++) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<HD>(dos_, 32 * u + 16 * st, 32 * dt, lane), pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<HD>(qs_, 32 * u + 16 * st, 32 * dt, lane), dsf, dk[dt], 0, 0, 0);
        }
      }
    }
    if (more) store_stats(smem + ((qt + 1) & 1) * ST);
    WAIT_VM0();
    __syncthreads();
  }
  if (krow < p.tk) {
    bf16_t* kp = p.dk + ((size_t)b * p.tk + krow) * p.lddk + h * HD;
    bf16_t* vp = p.dv + ((size_t)b * p.tk + krow) * p.lddv + h * HD;
#pragma unroll
    for (int dt = 0; dt < HD / 32; dt++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint2 pk;
        pk.x = pack_bf2(dk[dt][4 * j] * p.scale, dk[dt][4 * j + 1] * p.scale);
        pk.y = pack_bf2(dk[dt][4 * j + 2] * p.scale, dk[dt][4 * j + 3] * p.scale);
        *reinterpret_cast<uint2*>(kp + dt * 32 + 8 * j + 4 * g) = pk;
        pk.x = pack_bf2(dv[dt][4 * j], dv[dt][4 * j + 1]);
        pk.y = pack_bf2(dv[dt][4 * j + 2], dv[dt][4 * j + 3]);
        *reinterpret_cast<uint2*>(vp + dt * 32 + 8 * j + 4 * g) = pk;
      }
  }
}

// ================================================================================================
// backward, ONE pass (head_dim 32, tk <= 768): one workgroup of 8 waves per (batch, head); P and dS are computed once.
// ================================================================================================
// The two-kernel backward above evaluates exp(S - lse) twice per score (once with the query in the lane for dQ, once with the key in the lane
// for dK / dV) and reads Q, K, V, dO twice.  Here the keys of one (batch, head) are dealt to the 8 waves in blocks of 32 (wave w owns blocks
// w, w + 8, w + 16): dK / dV of those keys live in the wave's accumulators for the whole kernel, K sits in LDS, and the workgroup walks the
// queries in tiles of 64 (Q / dO tiles by LDS-DMA, double buffered).  Per (key block, 32 queries) a wave runs S = Q K^T and dP = dO V^T
// (key in the lane, queries in the registers), P = exp2(S c2 - lse), dS = P (dP - delta), dV += P^T dO and dK += dS^T Q with the
// accumulator registers as the k-slot operand - and dQ, whose reduction runs over the KEYS, i.e. over the lanes of dS: the 32 x 32 bf16
// block dS goes through 2 KiB of wave-private LDS (four ds_write_b64 per lane, [key][query] rows) and comes back query-in-the-lane through
// ds_read_b64_tr_b16, the same transpose read that feeds K^T.  The waves' partial dQ^T tiles (fp32) are summed through LDS in a fixed order
// (deterministic, no atomics) and written once.  delta = rowsum(dO * O) is computed per query tile from the rows themselves.
// Row statistics of a 64-query tile (lse and delta = rowsum(dO * O)) for the one-pass backward kernels: thread t < 64 * HD / 8 holds one 16-byte chunk of O and
// of dO of query q0 + t / (HD / 8) and the query's lse.  load() is called at the TOP of an iteration on a fresh object and store() at its END, so that the global
// loads fly under the tile's matrix work.  (Round 3-5 history: the first form had two waves load whole rows and sum them at once - a global round trip in front
// of their share of the tile; a form with the loaded registers carried around the loop made hipcc copy them into the loop-carried registers right behind the
// loads, i.e. wait for them at the top of every iteration: 2.3 us per tile, measured.)
template <int HD>
struct TileStats {
  static constexpr int CPR = HD / 8;
  uint4 o, d, ol;
  float lse;
  // olo_off: element offset of the optional second half of O from obase (0: none) - delta then comes from both halves (see attn_bwd_dq_mfma)
  __device__ __forceinline__ void load(const bf16_t* obase, int ldo, const bf16_t* dobase, int lddo, const float* lsebase, int q0, int tq, int tid, ptrdiff_t olo_off = 0) {
    ol = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 64 * CPR) {
      const int qc = min(q0 + tid / CPR, tq - 1), c = tid % CPR;
      o = *reinterpret_cast<const uint4*>(obase + (size_t)qc * ldo + c * 8);
      if (olo_off) ol = *reinterpret_cast<const uint4*>(obase + olo_off + (size_t)qc * ldo + c * 8);
      d = *reinterpret_cast<const uint4*>(dobase + (size_t)qc * lddo + c * 8);
      lse = lsebase[qc];
    }
  }
  __device__ __forceinline__ float delta() const {  // rowsum(dO * O) of the thread's query: complete in every lane of the query's CPR-lane group
    float s = (bf_lo32(o.x) + bf_lo32(ol.x)) * bf_lo32(d.x) + (bf_hi32(o.x) + bf_hi32(ol.x)) * bf_hi32(d.x) + (bf_lo32(o.y) + bf_lo32(ol.y)) * bf_lo32(d.y) +
              (bf_hi32(o.y) + bf_hi32(ol.y)) * bf_hi32(d.y) + (bf_lo32(o.z) + bf_lo32(ol.z)) * bf_lo32(d.z) + (bf_hi32(o.z) + bf_hi32(ol.z)) * bf_hi32(d.z) +
              (bf_lo32(o.w) + bf_lo32(ol.w)) * bf_lo32(d.w) + (bf_hi32(o.w) + bf_hi32(ol.w)) * bf_hi32(d.w);
#pragma unroll
    for (int m = 1; m < CPR; m <<= 1) s += __shfl_xor(s, m, 64);
    return s;
  }
  __device__ __forceinline__ void store(char* stats_base, int q0, int tq, int tid) const {
    if (tid < 64 * CPR) {
      const float s = delta();
      if (tid % CPR == 0) {
        float* st = reinterpret_cast<float*>(stats_base);
        st[tid / CPR] = q0 + tid / CPR < tq ? lse : 1e30f;  // lse = +big -> P = 0 for padded queries (their delta, a repeat of the last row's, is never used)
        st[64 + tid / CPR] = s;
      }
    }
  }
};

constexpr int FUSED_MAXKB = 3;
template <int HD>
struct FusedGeom {
  static constexpr int RB = HD * 2;                         // bytes per Q / K / dO row
  static constexpr int KS_BYTES = 8 * FUSED_MAXKB * 32 * RB;
  static constexpr int TB = 64 * RB;
  static constexpr int ST = 2 * TB + 512;                   // Q | dO | lse[64] | delta[64]
  static constexpr int DST_BYTES = 8 * 32 * 64;             // per wave: one 32 x 32 bf16 dS tile
  static constexpr int RED_BYTES = 8 * 64 * HD * 4;
  static constexpr int SMEM = KS_BYTES + 2 * ST + DST_BYTES + RED_BYTES;
};
template <int HD>
__global__ __launch_bounds__(512, 2) void attn_bwd_fused_mfma(AttnP p) {
  static_assert(HD == 32, "one 32-wide head-dim block per accumulator");
  using G = FusedGeom<HD>;
  using Stage = TileStage<HD, 64>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks_ = smem;
  char* stage0 = smem + G::KS_BYTES;
  char* dst_all = stage0 + 2 * G::ST;
  float* red = reinterpret_cast<float*>(dst_all + G::DST_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int b = blockIdx.x / p.h, h = blockIdx.x % p.h;
  const int nkb = (p.tk + 31) / 32;
  const bf16_t* qbase = p.q + (size_t)b * p.tq * p.ldq + h * HD;
  const bf16_t* dobase = p.d_o + (size_t)b * p.tq * p.lddo + h * HD;
  const bf16_t* obase = p.o + (size_t)b * p.tq * p.ldo + h * HD;
  const bf16_t* kbase = p.k + (size_t)b * p.tk * p.ldk + h * HD;
  const bf16_t* vbase = p.v + (size_t)b * p.tk * p.ldv + h * HD;
  const float* lsebase = p.lse + ((size_t)b * p.h + h) * p.tq;
  char* dstw = dst_all + wave_u * 2048;

  // K -> LDS (16 rows per 1 KiB piece, swizzle on the source chunk), V rows of this wave's keys -> registers
  {
    const uint32_t a0 = __builtin_amdgcn_readfirstlane(lds_address(ks_));
    for (int blk = wave_u; blk < nkb * 2; blk += 8) {
      const int row = blk * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      const int rr = min(row, p.tk - 1);
      glds16(__builtin_amdgcn_readfirstlane(a0 + blk * 1024), kbase + (size_t)rr * p.ldk + c * 8);
    }
  }
  short8v vf[FUSED_MAXKB][HD / 16];
#pragma unroll
  for (int kb = 0; kb < FUSED_MAXKB; kb++) {
    const int key = min(32 * (wave_u + 8 * kb) + (lane & 31), p.tk - 1);
    load_row_frags<HD>(vf[kb], vbase + (size_t)key * p.ldv, lane);
  }
  float16v dk[FUSED_MAXKB], dv[FUSED_MAXKB];
#pragma unroll
  for (int kb = 0; kb < FUSED_MAXKB; kb++) { zero16(dk[kb]); zero16(dv[kb]); }

  const int nqt = (p.tq + 63) / 64;
  float st_l = 0.f;  // threads 0..63: lse, 64..127: delta of the next tile's queries
  auto load_stats = [&](int q0) {
    if (tid < 64) {
      const int qi = q0 + tid;
      st_l = qi < p.tq ? lsebase[qi] : 1e30f;  // lse = +big -> P = 0 for padded queries
    } else if (tid < 128) {
      const int qi = q0 + tid - 64;
      float s = 0.f;
      if (qi < p.tq) {
        const uint4* op = reinterpret_cast<const uint4*>(obase + (size_t)qi * p.ldo);
        const uint4* dp = reinterpret_cast<const uint4*>(dobase + (size_t)qi * p.lddo);
#pragma unroll
        for (int c = 0; c < HD / 8; c++) {
          const uint4 a = op[c], d = dp[c];
          s += bf_lo32(a.x) * bf_lo32(d.x) + bf_hi32(a.x) * bf_hi32(d.x) + bf_lo32(a.y) * bf_lo32(d.y) + bf_hi32(a.y) * bf_hi32(d.y) +
               bf_lo32(a.z) * bf_lo32(d.z) + bf_hi32(a.z) * bf_hi32(d.z) + bf_lo32(a.w) * bf_lo32(d.w) + bf_hi32(a.w) * bf_hi32(d.w);
        }
        if (p.o_lo) {  // the second half of O (see attn_bwd_dq_mfma)
          const uint4* lp = reinterpret_cast<const uint4*>(p.o_lo + (obase - p.o) + (size_t)qi * p.ldo);
#pragma unroll
          for (int c = 0; c < HD / 8; c++) {
            const uint4 a = lp[c], d = dp[c];
            s += bf_lo32(a.x) * bf_lo32(d.x) + bf_hi32(a.x) * bf_hi32(d.x) + bf_lo32(a.y) * bf_lo32(d.y) + bf_hi32(a.y) * bf_hi32(d.y) +
                 bf_lo32(a.z) * bf_lo32(d.z) + bf_hi32(a.z) * bf_hi32(d.z) + bf_lo32(a.w) * bf_lo32(d.w) + bf_hi32(a.w) * bf_hi32(d.w);
          }
        }
      }
      st_l = s;
    }
  };
  auto store_stats = [&](char* base) {
    if (tid < 128) reinterpret_cast<float*>(base + 2 * G::TB)[tid] = st_l;
  };
  auto load_tiles = [&](char* base, int q0) {  // waves 0-3: the four pieces of the Q tile, waves 4-7: of the dO tile
    if (wave_u < 4) Stage::glds(base, qbase, p.ldq, q0, p.tq, lane, wave_u);
    else Stage::glds(base + G::TB, dobase, p.lddo, q0, p.tq, lane, wave_u - 4);
  };
  load_tiles(stage0, 0);
  load_stats(0);
  store_stats(stage0);

  for (int qt = 0; qt < nqt; qt++) {
    WAIT_VM0();
    __syncthreads();  // tile qt (and, first time, K) landed; everyone is done with the other stage and with `red`
    const char* qs_ = stage0 + (qt & 1) * G::ST;
    const char* dos_ = qs_ + G::TB;
    const float* stats = reinterpret_cast<const float*>(qs_ + 2 * G::TB);
    const bool more = qt + 1 < nqt;
    if (more) {
      load_tiles(stage0 + ((qt + 1) & 1) * G::ST, (qt + 1) * 64);
      load_stats((qt + 1) * 64);
    }
    const int nu = qt * 64 + 32 < p.tq ? 2 : 1;  // empty 32-query half of the last tile (workgroup-uniform)
#pragma unroll
    for (int u = 0; u < 2; u++) {
      float16v dq;
      zero16(dq);
      if (u < nu) {
        // Q / dO row fragments of these 32 queries: once per half tile, not once per key block
        short8v qf[HD / 16], dof[HD / 16];
#pragma unroll
        for (int ks = 0; ks < HD / 16; ks++) { qf[ks] = frag_km<HD>(qs_, 32 * u, ks, lane); dof[ks] = frag_km<HD>(dos_, 32 * u, ks, lane); }
#pragma unroll
        for (int kb = 0; kb < FUSED_MAXKB; kb++) {
          const int blk = wave_u + 8 * kb;
          if (blk < nkb) {
            const char* kt_ = ks_ + blk * 32 * G::RB;
            float16v s, dp;
            zero16(s); zero16(dp);
#pragma unroll
            for (int ks = 0; ks < HD / 16; ks++) {
              s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], frag_km<HD>(kt_, 0, ks, lane), s, 0, 0, 0);
              dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof[ks], vf[kb][ks], dp, 0, 0, 0);
            }
            // row statistics straight from LDS (broadcast reads; held in registers across the key blocks they cost 32 VGPRs and spilled):
            // acc reg r <-> query 32u + (r&3) + 8*(r>>2) + 4g
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const float4 l4 = *reinterpret_cast<const float4*>(stats + 32 * u + 8 * j + 4 * g);
              const float4 d4 = *reinterpret_cast<const float4*>(stats + 64 + 32 * u + 8 * j + 4 * g);
              const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
              for (int i = 0; i < 4; i++) {
                const float pr = fast_exp2(fmaf(s[4 * j + i], p.c2, -lv[i]));
                s[4 * j + i] = pr;
                dp[4 * j + i] = pr * (dp[4 * j + i] - dl[i]);
              }
            }
            if (blk * 32 + 32 > p.tk) {  // ragged last key block (wave-uniform): keys past the end contribute nothing
              const bool key_ok = blk * 32 + (lane & 31) < p.tk;
#pragma unroll
              for (int r = 0; r < 16; r++) { s[r] = key_ok ? s[r] : 0.f; dp[r] = key_ok ? dp[r] : 0.f; }
            }
#pragma unroll
            for (int st = 0; st < 2; st++) {
              const short8v pf = pack_slots(s, st);
              union { short8v v; uint32_t u32[4]; } dsf;
              dsf.v = pack_slots(dp, st);
              dv[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<HD>(dos_, 32 * u + 16 * st, 0, lane), pf, dv[kb], 0, 0, 0);
              dk[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<HD>(qs_, 32 * u + 16 * st, 0, lane), dsf.v, dk[kb], 0, 0, 0);
              // dS -> [key][query] rows of the wave's scratch tile: this lane's key, queries 8c + 4g .. + 3 (c = 2 st, 2 st + 1)
              const int j = lane & 31;
              *reinterpret_cast<uint2*>(dstw + swz_off<64>(j, 2 * st) + 8 * g) = make_uint2(dsf.u32[0], dsf.u32[1]);
              *reinterpret_cast<uint2*>(dstw + swz_off<64>(j, 2 * st + 1) + 8 * g) = make_uint2(dsf.u32[2], dsf.u32[3]);
            }
            // dQ^T[d][q] += K^T[d][key] dS^T[key][q]: both operands by transpose reads (K tile rows = keys; scratch tile rows = keys)
#pragma unroll
            for (int st = 0; st < 2; st++)
              dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<HD>(kt_, 16 * st, 0, lane), frag_tr<32>(dstw, 16 * st, 0, lane), dq, 0, 0, 0);
          }
        }
      }
      // partial dQ^T of this wave -> red[wave][q][d] (16-byte chunk index XORed with q & 7: a ds_write_b128 lane group covers consecutive q)
      const int q = 32 * u + (lane & 31);
#pragma unroll
      for (int c = 0; c < 4; c++)
        *reinterpret_cast<float4*>(red + ((size_t)(wave_u * 64 + q) * 8 + ((2 * c + g) ^ (q & 7))) * 4) =
            make_float4(dq[4 * c], dq[4 * c + 1], dq[4 * c + 2], dq[4 * c + 3]);
    }
    if (more) store_stats(stage0 + ((qt + 1) & 1) * G::ST);
    __syncthreads();
    {  // fixed-order sum over the 8 waves: wave w finishes queries 8w .. 8w + 7 of the tile, a lane 4 head-dim values of one query
      const int row = wave_u * 8 + (lane >> 3), chunk = lane & 7;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int ww = 0; ww < 8; ww++) {
        const float4 t = *reinterpret_cast<const float4*>(red + ((size_t)(ww * 64 + row) * 8 + (chunk ^ (row & 7))) * 4);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
      }
      const int qg = qt * 64 + row;
      if (qg < p.tq) {
        uint2 pk;
        pk.x = pack_bf2(acc.x * p.scale, acc.y * p.scale);
        pk.y = pack_bf2(acc.z * p.scale, acc.w * p.scale);
        *reinterpret_cast<uint2*>(p.dq + ((size_t)b * p.tq + qg) * p.lddq + h * HD + chunk * 4) = pk;
      }
    }
  }
#pragma unroll
  for (int kb = 0; kb < FUSED_MAXKB; kb++) {
    const int krow = 32 * (wave_u + 8 * kb) + (lane & 31);
    if (krow < p.tk) {
      bf16_t* kp = p.dk + ((size_t)b * p.tk + krow) * p.lddk + h * HD;
      bf16_t* vp = p.dv + ((size_t)b * p.tk + krow) * p.lddv + h * HD;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint2 pk;
        pk.x = pack_bf2(dk[kb][4 * j] * p.scale, dk[kb][4 * j + 1] * p.scale);
        pk.y = pack_bf2(dk[kb][4 * j + 2] * p.scale, dk[kb][4 * j + 3] * p.scale);
        *reinterpret_cast<uint2*>(kp + 8 * j + 4 * g) = pk;
        pk.x = pack_bf2(dv[kb][4 * j], dv[kb][4 * j + 1]);
        pk.y = pack_bf2(dv[kb][4 * j + 2], dv[kb][4 * j + 3]);
        *reinterpret_cast<uint2*>(vp + 8 * j + 4 * g) = pk;
      }
    }
  }
}

// ================================================================================================
// backward, ONE pass for head_dim 64 and any number of keys (round 5): passes of 256 keys, dQ formed from SHARED dS tiles
// ================================================================================================
// The one-pass kernel above keeps 3 key blocks per wave and sums eight partial dQ tiles through LDS; at head_dim 64 a wave has registers for ONE key block
// (dK + dV = 64 accumulators) and eight fp32 partial dQ tiles of 64 x 64 would be 128 KB.  This kernel therefore splits the work differently:
//   * a workgroup (8 waves) owns a contiguous range of key blocks of one (batch, head) and walks it in PASSES of 256 keys (wave w <-> key block w of the pass:
//     its K / V row fragments live in registers for the pass, dK / dV in its accumulators; the pass's K tile also sits in LDS for the dQ product);
//   * per pass it walks ALL queries in tiles of 64 (Q / dO by LDS-DMA, double buffered; lse and delta = rowsum(dO * O) per tile as above); phase A, every wave:
//     S = Q K^T, dP = dO V^T, P = exp2(S c2 - lse), dS = P (dP - delta) ONCE for its 32 keys, dV += P^T dO, dK += dS^T Q, and dS (bf16) -> a SHARED LDS
//     tile [256 keys][64 queries] (double buffered);
//   * phase B, one tile later (so that it overlaps phase A of the next tile: one barrier per tile): dQ^T[d][q] = sum over the pass's keys K^T[d][key] dS^T[key][q]
//     = four 32 x 32 output blocks x 16 k-steps, both operands by transpose reads (ds_read_b64_tr_b16).  Waves 0-3 form the four blocks of the EVEN tiles, waves
//     4-7 those of the odd tiles - every block is reduced over all 256 keys by ONE wave, so there is no cross-wave sum, no atomics, and the two waves of a SIMD
//     alternate (the SIMD's MFMA work per tile is the same in every iteration);
//   * dQ across passes: the block owner adds the previous passes' sum, kept in a fragment-ordered fp32 scratch (16 B per lane, fully coalesced, read back by the
//     same lane: plain program order), and the LAST pass writes bf16 dQ.  Keys split over G workgroups per (batch, head) (few (batch, head) pairs, many keys:
//     config 4 / 5): every workgroup publishes its sum with write-through stores, takes a ticket, and the last arriver adds the G sums in split order
//     (deterministic) - the p256 protocol (csrc/gemm256.hip) without any waiting: whoever sees ticket G - 1 knows that all others have published.
// P and dS are evaluated once per score (5 matmuls + 1 exponential instead of 7 + 2 of the dQ / dK-dV kernel pair).
struct OnePassP {
  AttnP a;
  float* part;          // [G][b*h][ntiles][4 blocks][4][64 lanes] float4: dQ sums in fragment order (nullptr: one pass, G = 1)
  unsigned* counters;   // [b*h] arrival tickets (zero on entry, left zero); G > 1 only
  int G;
};
template <int HD>
struct OnePassGeom {
  static constexpr int RB = HD * 2;
  static constexpr int KEYS = 256;
  static constexpr int KS_BYTES = KEYS * RB;
  static constexpr int TB = 64 * RB;
  static constexpr int ST = 2 * TB + 512;       // Q | dO | lse[64] | delta[64]
  static constexpr int DS_BYTES = KEYS * 128;   // [key][64 queries] bf16
  static constexpr int SMEM = KS_BYTES + 2 * ST + 2 * DS_BYTES;
  static constexpr int TILE_FLOATS = 64 * HD;   // one query tile of dQ
};
template <int HD>
__global__ __launch_bounds__(512, 2) void attn_bwd_onepass_mfma(OnePassP pp) {
  static_assert(HD == 64, "two 32-wide head-dim blocks x two 32-query halves = the four dQ blocks of a tile");
  using G = OnePassGeom<HD>;
  using Stage = TileStage<HD, 64>;
  const AttnP& p = pp.a;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks_ = smem;
  char* stage0 = smem + G::KS_BYTES;
  char* ds0 = stage0 + 2 * G::ST;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int l = xcd_remap((int)blockIdx.x, (int)gridDim.x);  // the G workgroups of one (batch, head) share Q / dO: neighbours on one XCD
  const int bh = l / pp.G, sp = l - bh * pp.G;
  const int b = bh / p.h, h = bh - b * p.h;
  const int nkb = (p.tk + 31) / 32;
  const int per = (nkb + pp.G - 1) / pp.G;
  const int kb0 = sp * per, kb1 = min(nkb, kb0 + per);   // (the host picks G so that no range is empty)
  const int npass = (kb1 - kb0 + 7) / 8;
  const int nqt = (p.tq + 63) / 64;
  const bf16_t* qbase = p.q + (size_t)b * p.tq * p.ldq + h * HD;
  const bf16_t* dobase = p.d_o + (size_t)b * p.tq * p.lddo + h * HD;
  const bf16_t* obase = p.o + (size_t)b * p.tq * p.ldo + h * HD;
  const bf16_t* kbase = p.k + (size_t)b * p.tk * p.ldk + h * HD;
  const bf16_t* vbase = p.v + (size_t)b * p.tk * p.ldv + h * HD;
  const float* lsebase = p.lse + ((size_t)b * p.h + h) * p.tq;
  float4* part4 = pp.part ? reinterpret_cast<float4*>(pp.part + ((size_t)sp * (p.b * p.h) + bh) * nqt * G::TILE_FLOATS) : nullptr;
  const int tl = wave_u & 3, dtb = tl & 1, qh = tl >> 1;  // the dQ block this wave forms in phase B: head-dim half, query half

  // delta = rowsum(dO * O) of every query of this (batch, head), once per workgroup, into p.delta (the workgroups that share the pair write the same values):
  // all 512 threads, 16 bytes of O and dO each per 64 queries.  The tile loop then needs two scalars per query (lse, delta), loaded by waves 0 / 1 at the top of
  // an iteration and put into the next stage's statistics at its end.
  float* dlbase = p.delta + ((size_t)b * p.h + h) * p.tq;
  for (int q0 = 0; q0 < p.tq; q0 += 64) {
    TileStats<HD> ts;
    ts.load(obase, p.ldo, dobase, p.lddo, lsebase, q0, p.tq, tid, p.o_lo ? p.o_lo - p.o : 0);
    const float dl = ts.delta();
    const int qi = q0 + (tid >> 3);
    if ((tid & 7) == 0 && qi < p.tq) dlbase[qi] = dl;
  }
  WAIT_VM0();  // (the first pass's opening barrier makes the values visible to waves 0 and 1)
  const float* stat_src = tid < 64 ? lsebase : dlbase;   // threads 0-63 fetch lse, 64-127 delta: one pointer and one value register per thread
  const float stat_pad = tid < 64 ? 1e30f : 0.f;         // lse = +big -> P = 0 for padded queries
  auto load_tiles = [&](char* base, int q0) {  // waves 0-3: the Q tile, waves 4-7: the dO tile (two 1 KiB pieces per wave)
    if (wave_u < 4) Stage::glds(base, qbase, p.ldq, q0, p.tq, lane, wave_u);
    else Stage::glds(base + G::TB, dobase, p.lddo, q0, p.tq, lane, wave_u - 4);
  };

  for (int ps = 0; ps < npass; ps++) {
    const int blk0 = kb0 + 8 * ps;
    const int blk = blk0 + wave_u;
    const bool active = blk < kb1;                       // wave-uniform
    const bool has_prev = ps > 0, last_pass = ps == npass - 1;
    const bool final_out = last_pass && pp.G == 1;
    __syncthreads();  // everybody is done with the previous pass's K tile, dS tiles and stages
    {  // K rows of the pass -> LDS: 32 pieces of 8 rows, swizzle on the source chunk
      const uint32_t a0 = __builtin_amdgcn_readfirstlane(lds_address(ks_));
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int piece = wave_u + 8 * i;
        const int row = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int rr = min(blk0 * 32 + row, p.tk - 1);
        glds16(__builtin_amdgcn_readfirstlane(a0 + piece * 1024), kbase + (size_t)rr * p.ldk + c * 8);
      }
    }
    if (!active) {   // phase B reduces over all 256 key rows of the dS tiles: the rows of a key block nobody owns in this pass stay zero
      const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; i++) *reinterpret_cast<uint4*>(ds0 + (i >> 2) * G::DS_BYTES + wave_u * 4096 + ((i & 3) * 64 + lane) * 16) = z;
    }
    short8v vf[HD / 16];
    load_row_frags<HD>(vf, vbase + (size_t)min(blk * 32 + (lane & 31), p.tk - 1) * p.ldv, lane);
    const char* kt_ = ks_ + wave_u * 32 * G::RB;  // this wave's 32 key rows of the pass's K tile
    float16v dk[HD / 32], dv[HD / 32];
#pragma unroll
    for (int i = 0; i < HD / 32; i++) { zero16(dk[i]); zero16(dv[i]); }
    load_tiles(stage0, 0);
    if (tid < 128) reinterpret_cast<float*>(stage0 + 2 * G::TB)[tid] = (tid & 63) < p.tq ? stat_src[tid & 63] : stat_pad;

    for (int it = 0; it <= nqt; it++) {
      WAIT_VM0();
      __syncthreads();  // tile `it` (first time: K) landed; dS tile it - 1 is complete; everyone is done with the other stage and with dS tile it - 2
      const bool more = it + 1 < nqt;
      float nx_stat = stat_pad;  // the next tile's statistics (threads 0-63: lse, 64-127: delta); loaded before the LDS-DMA, stored at the end of the iteration
      if (more) {
        const int qn = (it + 1) * 64 + (tid & 63);
        if (tid < 128 && qn < p.tq) nx_stat = stat_src[qn];
        load_tiles(stage0 + ((it + 1) & 1) * G::ST, (it + 1) * 64);
      }
      const int tb = it - 1;  // the tile whose dQ is formed in this iteration
      const bool do_b = tb >= 0 && (tb & 1) == (wave_u >> 2) && tb * 64 + 32 * qh < p.tq;  // wave-uniform
      if (it < nqt && active) {
        const char* qs_ = stage0 + (it & 1) * G::ST;
        const char* dos_ = qs_ + G::TB;
        const float* stats = reinterpret_cast<const float*>(qs_ + 2 * G::TB);
        char* dsw = ds0 + (it & 1) * G::DS_BYTES + wave_u * 4096;  // this wave's 32 key rows of the shared dS tile
        const int nu = it * 64 + 32 < p.tq ? 2 : 1;  // empty 32-query half of the last tile (workgroup-uniform)
#pragma unroll 1
        for (int u = 0; u < nu; u++) {
          {
            float16v s, dp;
            zero16(s); zero16(dp);
            {
              // every LDS read of the S / dP products is issued before the first MFMA (two reads + wait + MFMA per k-step left the LDS latency in front of
              // every MFMA: the kernel ran two waves per SIMD at the latency of its LDS reads)
              short8v qa[HD / 16], da[HD / 16], ka[HD / 16];
#pragma unroll
              for (int ks = 0; ks < HD / 16; ks++) { qa[ks] = frag_km<HD>(qs_, 32 * u, ks, lane); ka[ks] = frag_km<HD>(kt_, 0, ks, lane); da[ks] = frag_km<HD>(dos_, 32 * u, ks, lane); }
#pragma unroll
              for (int ks = 0; ks < HD / 16; ks++) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks], ka[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[ks], vf[ks], dp, 0, 0, 0);
              }
            }
            // the transposed Q / dO fragments of the dV / dK products (first 16 queries of the half): in flight under the exponentials below
            short8v tdo[2][HD / 32], tq[2][HD / 32];
#pragma unroll
            for (int dt = 0; dt < HD / 32; dt++) { tdo[0][dt] = frag_tr<HD>(dos_, 32 * u, 32 * dt, lane); tq[0][dt] = frag_tr<HD>(qs_, 32 * u, 32 * dt, lane); }
            // acc reg r <-> query 32u + (r&3) + 8*(r>>2) + 4g of the tile; column = this lane's key
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const float4 l4 = *reinterpret_cast<const float4*>(stats + 32 * u + 8 * j + 4 * g);
              const float4 d4 = *reinterpret_cast<const float4*>(stats + 64 + 32 * u + 8 * j + 4 * g);
              const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
              for (int i = 0; i < 4; i++) {
                const float pr = fast_exp2(fmaf(s[4 * j + i], p.c2, -lv[i]));
                s[4 * j + i] = pr;
                dp[4 * j + i] = pr * (dp[4 * j + i] - dl[i]);
              }
            }
            if (blk * 32 + 32 > p.tk) {  // ragged last key block (wave-uniform): keys past the end contribute nothing
              const bool key_ok = blk * 32 + (lane & 31) < p.tk;
#pragma unroll
              for (int r = 0; r < 16; r++) { s[r] = key_ok ? s[r] : 0.f; dp[r] = key_ok ? dp[r] : 0.f; }
            }
#pragma unroll
            for (int dt = 0; dt < HD / 32; dt++) { tdo[1][dt] = frag_tr<HD>(dos_, 32 * u + 16, 32 * dt, lane); tq[1][dt] = frag_tr<HD>(qs_, 32 * u + 16, 32 * dt, lane); }
#pragma unroll
            for (int st = 0; st < 2; st++) {
              const short8v pf = pack_slots(s, st);
              union { short8v v; uint32_t u32[4]; } dsf;
              dsf.v = pack_slots(dp, st);
#pragma unroll
              for (int dt = 0; dt < HD / 32; dt++) {
                dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tdo[st][dt], pf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[st][dt], dsf.v, dk[dt], 0, 0, 0);
              }
              // dS -> [key][query] rows of the shared tile: this lane's key, queries 32u + 16st + 4g .. + 3 and + 8 (16-byte chunks 4u + 2st, + 1)
              const int j = lane & 31;
              *reinterpret_cast<uint2*>(dsw + swz_off<128>(j, 4 * u + 2 * st) + 8 * g) = make_uint2(dsf.u32[0], dsf.u32[1]);
              *reinterpret_cast<uint2*>(dsw + swz_off<128>(j, 4 * u + 2 * st + 1) + 8 * g) = make_uint2(dsf.u32[2], dsf.u32[3]);
            }
          }
        }
      }
      if (do_b) {
        // dQ^T[d][q] (d = 32 dtb .. + 31 in the registers, q = 32 qh + (lane & 31)) over the pass's keys: K^T and dS^T by transpose reads
        const char* dsr = ds0 + (tb & 1) * G::DS_BYTES;
        float4 prev[4];  // the earlier passes' sum of this block: issued in front of the 16 matrix steps that hide most of its latency (held across phase A it spilled)
        if (has_prev) {
#pragma unroll
          for (int c = 0; c < 4; c++) prev[c] = part4[((size_t)(tb * 4 + tl) * 4 + c) * 64 + lane];
        }
        float16v dq;
        zero16(dq);
#pragma unroll
        for (int kb4 = 0; kb4 < 4; kb4++) {   // (a runtime trip count left one k-step per iteration behind its own four LDS reads: 16 exposed LDS latencies)
          short8v fk[4], fd[4];
#pragma unroll
          for (int i = 0; i < 4; i++) { fk[i] = frag_tr<HD>(ks_, 16 * (4 * kb4 + i), 32 * dtb, lane); fd[i] = frag_tr<64>(dsr, 16 * (4 * kb4 + i), 32 * qh, lane); }
#pragma unroll
          for (int i = 0; i < 4; i++) dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[i], fd[i], dq, 0, 0, 0);
        }
        if (has_prev) {
#pragma unroll
          for (int c = 0; c < 4; c++) { dq[4 * c] += prev[c].x; dq[4 * c + 1] += prev[c].y; dq[4 * c + 2] += prev[c].z; dq[4 * c + 3] += prev[c].w; }
        }
        if (final_out) {
          const int qg = tb * 64 + 32 * qh + (lane & 31);
          if (qg < p.tq) {
            bf16_t* op = p.dq + ((size_t)b * p.tq + qg) * p.lddq + h * HD + 32 * dtb;
#pragma unroll
            for (int j = 0; j < 4; j++) {
              uint2 pk;
              pk.x = pack_bf2(dq[4 * j] * p.scale, dq[4 * j + 1] * p.scale);
              pk.y = pack_bf2(dq[4 * j + 2] * p.scale, dq[4 * j + 3] * p.scale);
              *reinterpret_cast<uint2*>(op + 8 * j + 4 * g) = pk;
            }
          }
        } else if (last_pass) {  // G > 1: what the last arriver reads - write-through
          const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(part4 + (size_t)(tb * 4 + tl) * 4 * 64, 0, 4 * 64 * 16, 0x00020000);
#pragma unroll
          for (int c = 0; c < 4; c++) {
            u32x4v v = {__float_as_uint(dq[4 * c]), __float_as_uint(dq[4 * c + 1]), __float_as_uint(dq[4 * c + 2]), __float_as_uint(dq[4 * c + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (c * 64 + lane) * 16, 0, 16 /* sc1 */);
          }
        } else {
#pragma unroll
          for (int c = 0; c < 4; c++) part4[((size_t)(tb * 4 + tl) * 4 + c) * 64 + lane] = make_float4(dq[4 * c], dq[4 * c + 1], dq[4 * c + 2], dq[4 * c + 3]);
        }
      }
      if (more && tid < 128) reinterpret_cast<float*>(stage0 + ((it + 1) & 1) * G::ST + 2 * G::TB)[tid] = nx_stat;
    }
    if (active) {
      const int krow = blk * 32 + (lane & 31);
      if (krow < p.tk) {
        bf16_t* kp = p.dk + ((size_t)b * p.tk + krow) * p.lddk + h * HD;
        bf16_t* vp = p.dv + ((size_t)b * p.tk + krow) * p.lddv + h * HD;
#pragma unroll
        for (int dt = 0; dt < HD / 32; dt++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            uint2 pk;
            pk.x = pack_bf2(dk[dt][4 * j] * p.scale, dk[dt][4 * j + 1] * p.scale);
            pk.y = pack_bf2(dk[dt][4 * j + 2] * p.scale, dk[dt][4 * j + 3] * p.scale);
            *reinterpret_cast<uint2*>(kp + dt * 32 + 8 * j + 4 * g) = pk;
            pk.x = pack_bf2(dv[dt][4 * j], dv[dt][4 * j + 1]);
            pk.y = pack_bf2(dv[dt][4 * j + 2], dv[dt][4 * j + 3]);
            *reinterpret_cast<uint2*>(vp + dt * 32 + 8 * j + 4 * g) = pk;
          }
      }
    }
  }
  if (pp.G > 1) {
    // publish (the last pass's stores were write-through), take a ticket; the last arriver sums the G partial sums in split order
    WAIT_VM0();
    __syncthreads();
    volatile unsigned* ctl = reinterpret_cast<volatile unsigned*>(smem);
    if (tid == 0) ctl[0] = __hip_atomic_fetch_add(pp.counters + bh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ctl[0] != (unsigned)(pp.G - 1)) return;
    if (tid == 0) {
      __hip_atomic_store(pp.counters + bh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // left zero for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    const size_t split_stride = (size_t)(p.b * p.h) * nqt * (G::TILE_FLOATS / 4);  // float4s between the sums of two splits
    const float4* base4 = reinterpret_cast<const float4*>(pp.part) + (size_t)bh * nqt * (G::TILE_FLOATS / 4);
    for (int t = wave_u >> 2; t < nqt; t += 2) {
      const int qg = t * 64 + 32 * qh + (lane & 31);
      if (t * 64 + 32 * qh >= p.tq) continue;  // (wave-uniform)
      float4 acc[4];
#pragma unroll
      for (int c = 0; c < 4; c++) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s2 = 0; s2 < pp.G; s2++) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const float4 v = base4[s2 * split_stride + ((size_t)(t * 4 + tl) * 4 + c) * 64 + lane];
          acc[c].x += v.x; acc[c].y += v.y; acc[c].z += v.z; acc[c].w += v.w;
        }
      }
      if (qg < p.tq) {
        bf16_t* op = p.dq + ((size_t)b * p.tq + qg) * p.lddq + h * HD + 32 * dtb;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          uint2 pk;
          pk.x = pack_bf2(acc[j].x * p.scale, acc[j].y * p.scale);
          pk.y = pack_bf2(acc[j].z * p.scale, acc[j].w * p.scale);
          *reinterpret_cast<uint2*>(op + 8 * j + 4 * g) = pk;
        }
      }
    }
  }
}

// delta[b,h,q] = sum_d dO*O
__global__ void attn_delta_kernel(AttnP p) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)p.b * p.h * p.tq;
  if (idx >= total) return;
  const int q = (int)(idx % p.tq);
  const int h = (int)((idx / p.tq) % p.h);
  const int b = (int)(idx / ((long long)p.tq * p.h));
  const bf16_t* op = p.o + ((size_t)b * p.tq + q) * p.ldo + h * p.hd;
  const bf16_t* dop = p.d_o + ((size_t)b * p.tq + q) * p.lddo + h * p.hd;
  float s = 0.f;
  for (int d = 0; d < p.hd; d++) s += bf2f(op[d]) * bf2f(dop[d]);
  if (p.o_lo) {
    const bf16_t* lp = p.o_lo + (op - p.o);
    for (int d = 0; d < p.hd; d++) s += bf2f(lp[d]) * bf2f(dop[d]);
  }
  p.delta[idx] = s;
}

// vectorised delta: one wave per (b, q) token row; 16-byte loads of O and dO across all heads; lanes of one head
// (hd/8 adjacent lanes) reduce with xor shuffles.  hd in {8,16,32,64,128}.
__global__ __launch_bounds__(256) void attn_delta_vec_kernel(AttnP p) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)p.b * p.tq) return;
  const int q = (int)(row % p.tq), b = (int)(row / p.tq);
  const int lanes_per_head = p.hd >> 3;
  const int n_chunks = p.h * lanes_per_head;
  const bf16_t* op = p.o + (size_t)row * p.ldo;
  const bf16_t* dop = p.d_o + (size_t)row * p.lddo;
  for (int c0 = 0; c0 < n_chunks; c0 += 64) {
    const int c = c0 + lane;
    float s = 0.f;
    if (c < n_chunks) {
      const uint4 a = *reinterpret_cast<const uint4*>(op + c * 8), d = *reinterpret_cast<const uint4*>(dop + c * 8);
      s = bf2f((bf16_t)(a.x & 0xffff)) * bf2f((bf16_t)(d.x & 0xffff)) + bf2f((bf16_t)(a.x >> 16)) * bf2f((bf16_t)(d.x >> 16)) +
          bf2f((bf16_t)(a.y & 0xffff)) * bf2f((bf16_t)(d.y & 0xffff)) + bf2f((bf16_t)(a.y >> 16)) * bf2f((bf16_t)(d.y >> 16)) +
          bf2f((bf16_t)(a.z & 0xffff)) * bf2f((bf16_t)(d.z & 0xffff)) + bf2f((bf16_t)(a.z >> 16)) * bf2f((bf16_t)(d.z >> 16)) +
          bf2f((bf16_t)(a.w & 0xffff)) * bf2f((bf16_t)(d.w & 0xffff)) + bf2f((bf16_t)(a.w >> 16)) * bf2f((bf16_t)(d.w >> 16));
    }
    for (int o = lanes_per_head >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (c < n_chunks && (c % lanes_per_head) == 0) p.delta[((size_t)b * p.h + c / lanes_per_head) * p.tq + q] = s;
  }
}

// ================================================================================================
// generic kernels (any head_dim <= 128): one wave per output row, scores staged in LDS
// ================================================================================================
__global__ __launch_bounds__(256) void attn_fwd_generic(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* sc = reinterpret_cast<float*>(dyn_smem) + (size_t)wave * p.tk;
  const long long row = (long long)blockIdx.x * 4 + wave;
  if (row >= (long long)p.b * p.h * p.tq) return;
  const int q = (int)(row % p.tq), h = (int)((row / p.tq) % p.h), b = (int)(row / ((long long)p.tq * p.h));
  const bf16_t* qp = p.q + ((size_t)b * p.tq + q) * p.ldq + h * p.hd;
  float mx = NEG_INF;
  for (int j = lane; j < p.tk; j += 64) {
    const bf16_t* kp = p.k + ((size_t)b * p.tk + j) * p.ldk + h * p.hd;
    float s = 0.f;
    for (int d = 0; d < p.hd; d++) s += bf2f(qp[d]) * bf2f(kp[d]);
    s *= p.c2;
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float l = 0.f;
  for (int j = lane; j < p.tk; j += 64) {
    const float e = exp2f(sc[j] - mx);
    sc[j] = e;
    l += e;
  }
  l = wave_sum(l);
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < p.hd; d += 64) {
    float acc = 0.f;
    // P is rounded to bf16 before the PV product, like the MFMA kernel and like a bf16 SDPA
    for (int j = 0; j < p.tk; j++) acc += bf2f(f2bf(sc[j])) * bf2f(p.v[((size_t)b * p.tk + j) * p.ldv + h * p.hd + d]);
    const float val = acc / l;
    const bf16_t hi = f2bf(val);
    p.o[((size_t)b * p.tq + q) * p.ldo + h * p.hd + d] = hi;
    if (p.o_lo) p.o_lo[((size_t)b * p.tq + q) * p.ldo + h * p.hd + d] = f2bf(val - bf2f(hi));
  }
  if (lane == 0 && p.lse) p.lse[row] = mx + log2f(l);
}

__global__ __launch_bounds__(256) void attn_bwd_q_generic(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* ds = reinterpret_cast<float*>(dyn_smem) + (size_t)wave * p.tk;
  const long long row = (long long)blockIdx.x * 4 + wave;
  if (row >= (long long)p.b * p.h * p.tq) return;
  const int q = (int)(row % p.tq), h = (int)((row / p.tq) % p.h), b = (int)(row / ((long long)p.tq * p.h));
  const bf16_t* qp = p.q + ((size_t)b * p.tq + q) * p.ldq + h * p.hd;
  const bf16_t* dop = p.d_o + ((size_t)b * p.tq + q) * p.lddo + h * p.hd;
  const float lse = p.lse[row], dl = p.delta[row];
  for (int j = lane; j < p.tk; j += 64) {
    const bf16_t* kp = p.k + ((size_t)b * p.tk + j) * p.ldk + h * p.hd;
    const bf16_t* vp = p.v + ((size_t)b * p.tk + j) * p.ldv + h * p.hd;
    float s = 0.f, dp = 0.f;
    for (int d = 0; d < p.hd; d++) { s += bf2f(qp[d]) * bf2f(kp[d]); dp += bf2f(dop[d]) * bf2f(vp[d]); }
    ds[j] = bf2f(f2bf(exp2f(s * p.c2 - lse) * (dp - dl)));
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < p.hd; d += 64) {
    float acc = 0.f;
    for (int j = 0; j < p.tk; j++) acc += ds[j] * bf2f(p.k[((size_t)b * p.tk + j) * p.ldk + h * p.hd + d]);
    p.dq[((size_t)b * p.tq + q) * p.lddq + h * p.hd + d] = f2bf(acc * p.scale);
  }
}

__global__ __launch_bounds__(256) void attn_bwd_kv_generic(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* pb = reinterpret_cast<float*>(dyn_smem) + (size_t)wave * 2 * p.tq;
  float* dsb = pb + p.tq;
  const long long row = (long long)blockIdx.x * 4 + wave;
  if (row >= (long long)p.b * p.h * p.tk) return;
  const int j = (int)(row % p.tk), h = (int)((row / p.tk) % p.h), b = (int)(row / ((long long)p.tk * p.h));
  const bf16_t* kp = p.k + ((size_t)b * p.tk + j) * p.ldk + h * p.hd;
  const bf16_t* vp = p.v + ((size_t)b * p.tk + j) * p.ldv + h * p.hd;
  for (int i = lane; i < p.tq; i += 64) {
    const bf16_t* qp = p.q + ((size_t)b * p.tq + i) * p.ldq + h * p.hd;
    const bf16_t* dop = p.d_o + ((size_t)b * p.tq + i) * p.lddo + h * p.hd;
    const size_t si = ((size_t)b * p.h + h) * p.tq + i;
    float s = 0.f, dp = 0.f;
    for (int d = 0; d < p.hd; d++) { s += bf2f(qp[d]) * bf2f(kp[d]); dp += bf2f(dop[d]) * bf2f(vp[d]); }
    const float pr = exp2f(s * p.c2 - p.lse[si]);
    pb[i] = bf2f(f2bf(pr));
    dsb[i] = bf2f(f2bf(pr * (dp - p.delta[si])));
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < p.hd; d += 64) {
    float ak = 0.f, av = 0.f;
    for (int i = 0; i < p.tq; i++) {
      av += pb[i] * bf2f(p.d_o[((size_t)b * p.tq + i) * p.lddo + h * p.hd + d]);
      ak += dsb[i] * bf2f(p.q[((size_t)b * p.tq + i) * p.ldq + h * p.hd + d]);
    }
    p.dk[((size_t)b * p.tk + j) * p.lddk + h * p.hd + d] = f2bf(ak * p.scale);
    p.dv[((size_t)b * p.tk + j) * p.lddv + h * p.hd + d] = f2bf(av);
  }
}

bool mfma_ok(int hd, int force_generic, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
  if (force_generic || (hd != 32 && hd != 64)) return false;
  for (int v : lds) if (v & 7) return false;
  for (const void* q : ptrs) if (((uintptr_t)q) & 15) return false;
  return true;
}

}  // namespace

CINEMA_API int cinema_attention_fwd(const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v, int ldv, uint16_t* o, uint16_t* o_lo, int ldo,
                                    float* lse, int b, int h, int tq, int tk, int hd, float scale, int force_generic, void* stream) {
  if (!q || !k || !v || !o || b <= 0 || h <= 0 || tq <= 0 || tk <= 0 || hd <= 0 || hd > 128) return CINEMA_ERR_BAD_ARG;
  if (o_lo && (((uintptr_t)o_lo) & 15)) return CINEMA_ERR_BAD_ARG;
  AttnP p{};
  p.q = q; p.ldq = ldq; p.k = k; p.ldk = ldk; p.v = v; p.ldv = ldv; p.o = o; p.o_lo = o_lo; p.ldo = ldo; p.lse = lse;
  p.b = b; p.h = h; p.tq = tq; p.tk = tk; p.hd = hd; p.scale = scale; p.c2 = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  if (mfma_ok(hd, force_generic, {ldq, ldk, ldv, ldo}, {q, k, v, o})) {
    dim3 grid((tq + 127) / 128, h, b);
    const int v2 = getenv("CINEMA_ATTN_FWD_V2") ? atoi(getenv("CINEMA_ATTN_FWD_V2")) : 1;  // read per call: tests / tools A/B the forms in one process
    if (hd == 64) { if (v2) CINEMA_LAUNCH((attn_fwd_mfma<64, true>), grid, dim3(256), 0, st, p); else CINEMA_LAUNCH((attn_fwd_mfma<64, false>), grid, dim3(256), 0, st, p); }
    else { if (v2) CINEMA_LAUNCH((attn_fwd_mfma<32, true>), grid, dim3(256), 0, st, p); else CINEMA_LAUNCH((attn_fwd_mfma<32, false>), grid, dim3(256), 0, st, p); }
    return launch_status();
  }
  const size_t smem = (size_t)4 * tk * sizeof(float);
  if (smem > 150 * 1024) return CINEMA_ERR_UNSUPPORTED;
  const long long rows = (long long)b * h * tq;
  CINEMA_LAUNCH(attn_fwd_generic, dim3((unsigned)((rows + 3) / 4)), dim3(256), smem, st, p);
  return launch_status();
}

// ---- one-pass backward for head_dim 64 (attn_bwd_onepass_mfma): how the keys of a (batch, head) are dealt to workgroups
struct OnePassPlan { int G; int npass; long long part_floats; };
static int onepass_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n;
}
static OnePassPlan onepass_plan(int b, int h, int tq, int tk) {
  // one workgroup per compute unit (132 KB of LDS): with fewer (batch, head) pairs than units the keys of a pair are split over G workgroups - never below one
  // full pass of 8 key blocks per workgroup, and so that no workgroup's range is empty
  const int nkb = (tk + 31) / 32, bh = b * h;
  int g = onepass_cu_count() / bh;
  if (g < 1) g = 1;
  const char* force = getenv("CINEMA_ATTN_ONEPASS_G");
  if (force && atoi(force) > 0) g = atoi(force);
  const int max_g = (nkb + 7) / 8;
  if (g > max_g) g = max_g;
  while (g > 1 && (g - 1) * ((nkb + g - 1) / g) >= nkb) g--;
  OnePassPlan pl;
  pl.G = g;
  pl.npass = ((nkb + g - 1) / g + 7) / 8;
  pl.part_floats = (g > 1 || pl.npass > 1) ? (long long)g * bh * ((tq + 63) / 64) * OnePassGeom<64>::TILE_FLOATS : 0;
  return pl;
}

CINEMA_API long long cinema_attention_bwd_workspace_bytes(int b, int h, int tq, int tk, int hd) {
  if (b <= 0 || h <= 0 || tq <= 0 || tk <= 0 || hd != 64) return 0;
  return onepass_plan(b, h, tq, tk).part_floats * 4;
}

static int attention_bwd_impl(const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v, int ldv, const uint16_t* o, const uint16_t* o_lo,
                              int ldo, const uint16_t* d_o, int lddo, const float* lse, float* delta, uint16_t* dq, int lddq,
                              uint16_t* dk, int lddk, uint16_t* dv, int lddv, int b, int h, int tq, int tk, int hd, float scale,
                              int force_generic, float* workspace, long long workspace_bytes, unsigned* counters, int n_counters, void* stream) {
  if (!q || !k || !v || !o || !d_o || !lse || !delta || !dq || !dk || !dv || b <= 0 || h <= 0 || tq <= 0 || tk <= 0 || hd <= 0 || hd > 128)
    return CINEMA_ERR_BAD_ARG;
  AttnP p{};
  if (o_lo && (((uintptr_t)o_lo) & 15)) return CINEMA_ERR_BAD_ARG;
  p.q = q; p.ldq = ldq; p.k = k; p.ldk = ldk; p.v = v; p.ldv = ldv; p.o = const_cast<uint16_t*>(o); p.o_lo = const_cast<uint16_t*>(o_lo); p.ldo = ldo;
  p.lse = const_cast<float*>(lse); p.d_o = d_o; p.lddo = lddo; p.delta = delta;
  p.dq = dq; p.lddq = lddq; p.dk = dk; p.lddk = lddk; p.dv = dv; p.lddv = lddv;
  p.b = b; p.h = h; p.tq = tq; p.tk = tk; p.hd = hd; p.scale = scale; p.c2 = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  const long long nq = (long long)b * h * tq;
  const bool pow2 = (hd & (hd - 1)) == 0 && hd >= 8;
  const bool mfma = mfma_ok(hd, force_generic, {ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv}, {q, k, v, d_o, dq, dk, dv}) && !(((uintptr_t)o) & 15);
  if (mfma) {
    // delta is produced by the dQ kernel (from its O / dO row fragments) and consumed by the dK/dV kernel behind it
  } else if (!o_lo && pow2 && !(ldo & 7) && !(lddo & 7) && !(((uintptr_t)o) & 15) && !(((uintptr_t)d_o) & 15)) {
    const long long rows = (long long)b * tq;
    CINEMA_LAUNCH(attn_delta_vec_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p);
  } else {
    CINEMA_LAUNCH(attn_delta_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, p);
  }
  if (mfma && mfma_ok(hd, force_generic, {ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv}, {q, k, v, d_o, dq, dk, dv})) {
    // one-pass backward (hd 32, at most 3 key blocks of 32 per wave): CINEMA_ATTN_FUSED=0 keeps the two-kernel form
    const char* fused_txt = getenv("CINEMA_ATTN_FUSED");  // read per call: tests switch between the two forms
    const int fused_env = fused_txt ? atoi(fused_txt) : 1;
    if (fused_env && hd == 32 && tk <= 8 * FUSED_MAXKB * 32 && !(ldo & 7)) {
      static bool attr_set[16] = {};
      const hipError_t e = dyn_lds_attr_once(attr_set, reinterpret_cast<const void*>(attn_bwd_fused_mfma<32>), FusedGeom<32>::SMEM);
      if (e != hipSuccess) return (int)e;
      CINEMA_LAUNCH(attn_bwd_fused_mfma<32>, dim3((unsigned)(b * h)), dim3(512), (size_t)FusedGeom<32>::SMEM, st, p);
      return launch_status();
    }
    // one-pass backward for head_dim 64 (any number of keys): needs its scratch when the keys take several passes or are split over workgroups.
    // OFF by default (CINEMA_ATTN_ONEPASS=1 selects it): parity-green and deterministic, and after three rounds of tuning (statistics off the critical path,
    // LDS reads issued in batches ahead of their MFMAs, phase B unrolled: 194 -> 131 us at the config-2 encoder) ON PAR with the dQ + dK/dV pair, not ahead of it:
    // config 2 encoder 131-137 vs 137-142 us, config 5 encoder 324-352 vs 328-397, config 4 (3073 tokens, three passes on 240 workgroups) 522-548 vs 459-490;
    // the config-2 step moves by -0.06 ms (profiles/r05_i_attn_bwd*.txt).  At one key block per wave every wave re-reads the Q / dO tiles from LDS in two
    // layouts for 32 keys, which is what the kernel pair does as well - the saved second evaluation of P pays for the dS tile traffic and phase B, no more.
    const char* op_txt = getenv("CINEMA_ATTN_ONEPASS");
    if (fused_env && (op_txt ? atoi(op_txt) : 0) && hd == 64 && !(ldo & 7)) {
      const OnePassPlan pl = onepass_plan(b, h, tq, tk);
      const bool ws_ok = pl.part_floats == 0 || (workspace && workspace_bytes >= pl.part_floats * 4 && !(((uintptr_t)workspace) & 15));
      const bool cnt_ok = pl.G == 1 || (counters && n_counters >= b * h);
      if (ws_ok && cnt_ok) {
        static bool attr_set64[16] = {};
        const hipError_t e = dyn_lds_attr_once(attr_set64, reinterpret_cast<const void*>(attn_bwd_onepass_mfma<64>), OnePassGeom<64>::SMEM);
        if (e != hipSuccess) return (int)e;
        OnePassP pp{};
        pp.a = p; pp.part = pl.part_floats ? workspace : nullptr; pp.counters = counters; pp.G = pl.G;
        CINEMA_LAUNCH(attn_bwd_onepass_mfma<64>, dim3((unsigned)(b * h * pl.G)), dim3(512), (size_t)OnePassGeom<64>::SMEM, st, pp);
        return launch_status();
      }
    }
    dim3 gq((tq + 127) / 128, h, b), gk((tk + 127) / 128, h, b);
    if (hd == 64) {
      CINEMA_LAUNCH(attn_bwd_dq_mfma<64>, gq, dim3(256), 0, st, p);
      CINEMA_LAUNCH(attn_bwd_dkv_mfma<64>, gk, dim3(256), 0, st, p);
    } else {
      CINEMA_LAUNCH(attn_bwd_dq_mfma<32>, gq, dim3(256), 0, st, p);
      CINEMA_LAUNCH(attn_bwd_dkv_mfma<32>, gk, dim3(256), 0, st, p);
    }
    return launch_status();
  }
  if ((size_t)4 * tk * sizeof(float) > 150 * 1024 || (size_t)8 * tq * sizeof(float) > 150 * 1024) return CINEMA_ERR_UNSUPPORTED;
  CINEMA_LAUNCH(attn_bwd_q_generic, dim3((unsigned)((nq + 3) / 4)), dim3(256), (size_t)4 * tk * sizeof(float), st, p);
  const long long nk = (long long)b * h * tk;
  CINEMA_LAUNCH(attn_bwd_kv_generic, dim3((unsigned)((nk + 3) / 4)), dim3(256), (size_t)8 * tq * sizeof(float), st, p);
  return launch_status();
}

CINEMA_API int cinema_attention_bwd(const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v, int ldv, const uint16_t* o, const uint16_t* o_lo,
                                    int ldo, const uint16_t* d_o, int lddo, const float* lse, float* delta, uint16_t* dq, int lddq,
                                    uint16_t* dk, int lddk, uint16_t* dv, int lddv, int b, int h, int tq, int tk, int hd, float scale,
                                    int force_generic, void* stream) {
  return attention_bwd_impl(q, ldq, k, ldk, v, ldv, o, o_lo, ldo, d_o, lddo, lse, delta, dq, lddq, dk, lddk, dv, lddv, b, h, tq, tk, hd, scale, force_generic,
                            nullptr, 0, nullptr, 0, stream);
}

CINEMA_API int cinema_attention_bwd_ws(const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v, int ldv, const uint16_t* o, const uint16_t* o_lo,
                                       int ldo, const uint16_t* d_o, int lddo, const float* lse, float* delta, uint16_t* dq, int lddq,
                                       uint16_t* dk, int lddk, uint16_t* dv, int lddv, int b, int h, int tq, int tk, int hd, float scale,
                                       int force_generic, float* workspace, long long workspace_bytes, unsigned* counters, int n_counters, void* stream) {
  return attention_bwd_impl(q, ldq, k, ldk, v, ldv, o, o_lo, ldo, d_o, lddo, lse, delta, dq, lddq, dk, lddk, dv, lddv, b, h, tq, tk, hd, scale, force_generic,
                            workspace, workspace_bytes, counters, n_counters, stream);
}
