// igemm2.h -- second-generation implicit GEMM on the gfx950 f32 matrix cores: the same descriptor
// (IgemmProb, igemm.h) and the same arithmetic (v_mfma_f32_32x32x2_f32, exact fp32) as igemm_kernel,
// but built for problems whose operands can be fetched 16 bytes at a time:
//
//   * every global access of the staging loop is a dwordx4 load (4 floats along whichever index the
//     tensor is contiguous in), so a 32-deep slab costs 2-12 loads per thread instead of 16 scalar
//     gathers with 16 address computations;
//   * operands that are contiguous along the reduction index are kept K-contiguous in LDS
//     ([row][32*WK + 4]); a lane then owns 16 consecutive k of its row (lanes 0-31: k 0..15, lanes
//     32-63: k 16..31 of the slab) and fetches them with four conflict-free ds_read_b128 -- the
//     k-pairing of the sixteen 32x32x2 MFMAs of a slab is {m, 16+m}, which both operands follow;
//     operands contiguous along the output index stay k-major ([k][cols + 4], ds_read_b32);
//   * LDS is double buffered: one barrier per slab, global loads of slab s+1 are in flight while
//     the MFMAs of slab s run;
//   * workgroup shapes (CFG): 64x64 (2x2 waves), 128x32 (4x1 waves, narrow-N convolutions), 32x64 with
//     the reduction split 2 ways (1x2 waves x 2, 48 KB of LDS: three per CU) and 32x64 with the four
//     waves splitting the reduction (small-M dense layers: fills the chip
//     without a global split-K pass; partial accumulators are summed through LDS in wave order);
//   * the bias-gradient "ones row" of a weight-gradient problem is not an extra, nearly empty row
//     tile any more: the workgroups of row tile 0 sum the staged dY slab column-wise (VALU, from
//     LDS) while the matrix pipe works, and write the row the descriptor asks for.
//
// Eligibility (16-byte alignment, 4-runs in the addressing tables, single-part operands) is decided
// on the host per launch (engine.hip: v2_plan); everything else keeps running on igemm_kernel.
#pragma once
#include "igemm.h"

namespace grl {

enum { I2_P_ALONG_R = 0, I2_P_ALONG_I = 1 };
enum { I2_Q_ALONG_R = 0, I2_Q_ALONG_J = 1 };
enum { I2F_ONES = 1, I2F_KTAIL = 2 };

template <int CFG> struct I2Cfg;
template <> struct I2Cfg<0> { static constexpr int BM = 64, BN = 64, WM = 2, WN = 2, WK = 1, FM = 1, FN = 1; };
template <> struct I2Cfg<1> { static constexpr int BM = 128, BN = 32, WM = 4, WN = 1, WK = 1, FM = 1, FN = 1; };
template <> struct I2Cfg<2> { static constexpr int BM = 32, BN = 64, WM = 1, WN = 1, WK = 4, FM = 1, FN = 2; };
template <> struct I2Cfg<3> { static constexpr int BM = 32, BN = 64, WM = 1, WN = 2, WK = 2, FM = 1, FN = 1; };

static inline int i2_bm(int cfg) { return cfg == 0 ? 64 : (cfg == 1 ? 128 : 32); }   // cfg 2, 3: 32
static inline int i2_bn(int cfg) { return cfg == 0 ? 64 : (cfg == 1 ? 32 : 64); }

#ifndef GRL_GEMM_TYPES_ONLY
#ifdef GRL_HOSTEMU
#include "igemm2_ref1.h"   // tests/hostemu: the emulation build only
#else

#ifndef I2_ABLATE
#define I2_ABLATE 0   // development only (scripts/igemm_bench.hip): 1 no global loads, 2 no LDS stores, 4 no barrier
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define I2_OOB ((int)0x80000000)   // byte offset beyond num_records: the buffer load returns zeros

__device__ __forceinline__ __amdgpu_buffer_rsrc_t i2_rsrc(const float* p) {
  // descriptor inputs made provably wave-uniform (no waterfall loop around the buffer loads)
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 i2_ld(__amdgpu_buffer_rsrc_t rs, int byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}

// FLAGS: I2F_ONES  -- the problems carry a bias-gradient ones row (p_ones_i == M-1), Q along j, WK == 1
//        I2F_KTAIL -- K % 4 != 0 (affine operands along r): elements past r_end are zeroed one by one
// LDS floats one tile of an instantiation needs
template <int PL, int QL, int CFG>
struct I2Lds {
  using C = I2Cfg<CFG>;
  static constexpr int BKT = 32 * C::WK;
  static constexpr int PSZ = PL == I2_P_ALONG_R ? C::BM * BKT : BKT * (C::BM + 4);   // K-contiguous rows are swizzled, not padded
  static constexpr int QSZ = QL == I2_Q_ALONG_R ? C::BN * BKT : BKT * (C::BN + 4);
  static constexpr int RED = C::WK > 1 ? C::WK * C::BM * (C::BN + 4) : C::BM * (C::BN + 4);
  static constexpr int value = 2 * (PSZ + QSZ) > RED ? 2 * (PSZ + QSZ) : RED;
};

#ifdef GRL_TILE_TRACE
// Measurement build only (scripts/tile_trace.sh): every workgroup of every igemm2 launch records, in the 6-word slot its
// descriptor copy points to (dbg_t), {start, end (100 MHz device clock), HW_ID | XCC_ID << 32 | slabs << 40, tile,
// first barrier passed, reduction loop done} -- the schedule of the launch as the hardware ran it.
#define I2_TRACE(k) do { if (threadIdx.x == 0 && pb->dbg_t) ((unsigned long long*)pb->dbg_t)[k] = wall_clock64(); } while (0)
__device__ __forceinline__ void i2_trace_record(const IgemmProb* pb, const int4 tl, unsigned long long t0, int cfg) {
  if (threadIdx.x != 0 || !pb->dbg_t) return;
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID[3:0]
  const int chunk = min(pb->K - tl.y * pb->k_chunk, pb->k_chunk);
  unsigned long long* d = (unsigned long long*)pb->dbg_t;
  d[0] = t0; d[1] = wall_clock64();
  d[2] = (unsigned long long)hw | ((unsigned long long)xcc << 32) | ((unsigned long long)((chunk + 31) / 32) << 40) |
         ((unsigned long long)cfg << 56);
  d[3] = (unsigned long long)(unsigned)tl.x | ((unsigned long long)(unsigned)tl.y << 16) |
         ((unsigned long long)(unsigned)tl.z << 32) | ((unsigned long long)gridDim.x << 48);
}
#else
#define I2_TRACE(k) do { } while (0)
#endif

// one tile; `lds` is the workgroup's staging area (I2Lds<...>::value floats, 16-byte aligned).  A function, not the
// kernel, so that one launch can carry tiles of two instantiations (igemm2_pair_kernel).
template <int PL, int QL, int PM, int QM, int CFG, int FLAGS>
__device__ __forceinline__ void igemm2_tile(const IgemmProb* __restrict__ pb, const int4 tl, float* __restrict__ lds) {
  using C = I2Cfg<CFG>;
  constexpr int BM = C::BM, BN = C::BN, WK = C::WK, FM = C::FM, FN = C::FN, WN = C::WN, WM = C::WM;
  constexpr bool ONES = (FLAGS & I2F_ONES) != 0, KTAIL = (FLAGS & I2F_KTAIL) != 0;
  constexpr int BKT = 32 * WK;                 // reduction depth staged per barrier
  // Row strides of the two layouts of an operand.  K-contiguous rows (operand along r) are NOT padded: the 16-byte
  // chunks of a row are XOR-swizzled with row bits instead (i2_swz), which keeps the four ds_read_b128 per slab
  // conflict-free and makes the 128x32 shape fit four workgroups into a CU's 160 KB (2 x (128 + 32) x 32 floats =
  // exactly 40 KB; with 4 floats of padding per row it was 46 KB, three per CU, and conv2_bwd's 1024 tiles ran as a
  // wave of 768 plus a third-full wave of 256).
  constexpr int LDPK = BKT, LDPM = BM + 4;
  constexpr int LDQK = BKT, LDQN = BN + 4;
  // chunk c (16 bytes) of row `row` lives at chunk c ^ swz(row).  A ds_read_b128 is served in groups of 16 lanes
  // (MI355X_MICROARCH.md, LDS: {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) reading one chunk index from 16 rows: with
  // 32-float rows two consecutive rows cover the 64 banks, so the swizzle takes row bits 1..3; with 64-float rows and
  // longer every row starts at bank 0 and the low four row bits are needed.
  auto swz = [](int row) { return BKT == 32 ? ((row >> 1) & 7) : (row & 15); };
  constexpr int PSZ = PL == I2_P_ALONG_R ? BM * LDPK : BKT * LDPM;
  constexpr int QSZ = QL == I2_Q_ALONG_R ? BN * LDQK : BKT * LDQN;
  constexpr int BUF = PSZ + QSZ;
  constexpr int RED = WK > 1 ? WK * BM * (BN + 4) : BM * (BN + 4);   // k-split partials / output tile of the wide epilogue
  static_assert((2 * BUF > RED ? 2 * BUF : RED) == I2Lds<PL, QL, CFG>::value, "I2Lds out of step with the tile layout");

#ifdef I2_TIMING
#define I2_STAMP(k) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1)) { \
    unsigned long long* d_ = (unsigned long long*)pb->dbg_t; if (d_) d_[(blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 16 : 8)) + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define I2_STAMP(k) do { } while (0)
#endif
  I2_STAMP(0);
  const int N = pb->N, K = pb->K;
  const int ones_i = pb->p_ones_i;
  const int M = ONES ? pb->M - 1 : pb->M;   // rows tiled; the ones row is handled by row tile 0
  const int i0 = tl.z * BM, j0 = tl.w * BN;
  const int r_begin = tl.y * pb->k_chunk;
  const int r_end = min(K, r_begin + pb->k_chunk);

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wk = wave / (WM * WN), wm = (wave / WN) % WM, wn = wave % WN;

  const __amdgpu_buffer_rsrc_t rsP = i2_rsrc(pb->p_base[0]);
  const __amdgpu_buffer_rsrc_t rsQ = i2_rsrc(pb->q_base[0]);
  const gci32 pTi = (gci32)pb->p_tab_i;
  const gci32 pTr = (gci32)pb->p_tab_r;
  const gcu64 pVm = (gcu64)pb->p_vmask_i;
  const gcu8 pTap = (gcu8)pb->p_tap_r;
  const gci32 qTr = (gci32)pb->q_tab_r;
  const int pLi = pb->p_ld_i[0], pLr = pb->p_ld_r[0];
  const int qLr = pb->q_ld_r[0], qLj = pb->q_ld_j[0];

  // ------------------------------------------------------------------ staging coordinates
  // operand along r: a thread owns k-quad *_q of rows/cols *_l + e*STEP.
  // operand along i/j: it owns row/col-quad *_q of the k's *_l + e*STEP.
  constexpr int PNQ = PL == I2_P_ALONG_R ? BKT / 4 : BM / 4;
  constexpr int PSTEP = 256 / PNQ;
  constexpr int NVP = (PL == I2_P_ALONG_R ? BM : BKT) / PSTEP;
  const int p_q = t % PNQ, p_l = t / PNQ;
  constexpr int NFP = PL == I2_P_ALONG_R ? NVP : 1;
  int p_fix[NFP];            // element offset of the fixed (row) term, or I2_OOB/4-safe sentinel via p_fok
  bool p_fok[NFP];
  uint64_t p_vm[NFP];
  if (PL == I2_P_ALONG_R) {
#pragma unroll
    for (int e = 0; e < NVP; ++e) {
      const int i = i0 + p_l + e * PSTEP;
      p_fok[e] = i < M;
      const int ic = p_fok[e] ? i : i0;
      p_fix[e] = PM != PM_AFFINE ? pTi[ic] : ic * pLi;
      p_vm[e] = PM == PM_TABLE_MASK ? pVm[ic] : ~0ull;
    }
  } else {
    const int i = i0 + 4 * p_q;
    p_fok[0] = i < M;
    const int ic = p_fok[0] ? i : i0;
    p_fix[0] = PM != PM_AFFINE ? pTi[ic] : ic;
    p_vm[0] = ~0ull;
  }
  constexpr int QNQ = QL == I2_Q_ALONG_R ? BKT / 4 : BN / 4;
  constexpr int QSTEP = 256 / QNQ;
  constexpr int NVQ = (QL == I2_Q_ALONG_R ? BN : BKT) / QSTEP;
  const int q_q = t % QNQ, q_l = t / QNQ;
  constexpr int NFQ = QL == I2_Q_ALONG_R ? NVQ : 1;
  int q_fix[NFQ];
  bool q_fok[NFQ];
  if (QL == I2_Q_ALONG_R) {
#pragma unroll
    for (int e = 0; e < NVQ; ++e) {
      const int j = j0 + q_l + e * QSTEP;
      q_fok[e] = j < N;
      q_fix[e] = (q_fok[e] ? j : j0) * qLj;
    }
  } else {
    const int j = j0 + 4 * q_q;
    q_fok[0] = j < N;
    q_fix[0] = q_fok[0] ? j : j0;
  }

  // Table entries (vector global loads: they share vmcnt with the data loads).  Each register set of
  // staged data has its own table registers, fetched right after the set's data loads were issued and
  // consumed two slabs later by the set's next refill: at that point the only younger memory operations
  // are the other set's loads, so the wait in front of the consumer is the one the LDS stores of this set
  // need anyway.  (One shared copy, fetched last and consumed next, made that wait vmcnt(0): every slab
  // drained all loads in flight.)
  constexpr int NTP = PL == I2_P_ALONG_R ? 1 : NVP;
  constexpr int NTQ = QL == I2_Q_ALONG_R ? 1 : NVQ;
  struct Tabs { int p[NTP]; int tap; int q[NTQ]; };
  auto fetch_tabs = [&](int r0, Tabs& tb) {
    if (PM != PM_AFFINE) {
#pragma unroll
      for (int e = 0; e < NTP; ++e) {
        const int r = PL == I2_P_ALONG_R ? r0 + 4 * p_q : r0 + p_l + e * PSTEP;
        const int rc = r < r_end ? r : r_begin;
        tb.p[e] = pTr[rc];
        if (PM == PM_TABLE_MASK && PL == I2_P_ALONG_R) tb.tap = pTap[rc];
      }
    }
    if (QM == QM_TABLE) {
#pragma unroll
      for (int e = 0; e < NTQ; ++e) {
        const int r = QL == I2_Q_ALONG_R ? r0 + 4 * q_q : r0 + q_l + e * QSTEP;
        tb.q[e] = qTr[r < r_end ? r : r_begin];
      }
    }
  };
  Tabs tbA, tbB;
  tbA.tap = tbB.tap = 0;

  // two register sets: while slab s runs from LDS, slab s+1 (one set) is written to the other LDS buffer
  // and slab s+2 (other set) is still in flight -- loads have two slabs of MFMAs to land
  f32x4 pv[NVP], qv[NVQ], pvB[NVP], qvB[NVQ];
  unsigned p_kb = 0xfu, q_kb = 0xfu, p_kbB = 0xfu, q_kbB = 0xfu;   // KTAIL: validity of the 4 elements of an along-r vector

  // one vector of slab r0 (compile-time e): offset select + buffer load, nothing else
  auto load_p = [&](int r0, int e, f32x4 (&pv)[NVP], unsigned& p_kb, const Tabs& tb) {
    if (I2_ABLATE & 1) { pv[e] = f32x4{1.f, 2.f, 3.f, 4.f}; return; }
    if (PL == I2_P_ALONG_R) {
      const int r = r0 + 4 * p_q;
      bool ok = r < r_end && p_fok[e];
      if (PM == PM_TABLE_MASK) ok = ok && ((p_vm[e] >> tb.tap) & 1ull);
      const int colterm = PM != PM_AFFINE ? tb.p[0] : r;
      pv[e] = i2_ld(rsP, ok ? (p_fix[e] + colterm) * 4 : I2_OOB);
      if (KTAIL && e == 0) p_kb = r + 4 <= r_end ? 0xfu : (0xfu >> min(4, max(0, r + 4 - r_end)));
    } else {
      const int r = r0 + p_l + e * PSTEP;
      const bool ok = r < r_end && p_fok[0];
      const int colterm = PM != PM_AFFINE ? tb.p[e] : r * pLr;
      pv[e] = i2_ld(rsP, ok ? (p_fix[0] + colterm) * 4 : I2_OOB);
    }
  };
  auto load_q = [&](int r0, int e, f32x4 (&qv)[NVQ], unsigned& q_kb, const Tabs& tb) {
    if (I2_ABLATE & 1) { qv[e] = f32x4{1.f, 2.f, 3.f, 4.f}; return; }
    if (QL == I2_Q_ALONG_R) {
      const int r = r0 + 4 * q_q;
      const bool ok = r < r_end && q_fok[e];
      const int rowterm = QM == QM_TABLE ? tb.q[0] : r;
      qv[e] = i2_ld(rsQ, ok ? (rowterm + q_fix[e]) * 4 : I2_OOB);
      if (KTAIL && e == 0) q_kb = r + 4 <= r_end ? 0xfu : (0xfu >> min(4, max(0, r + 4 - r_end)));
    } else {
      const int r = r0 + q_l + e * QSTEP;
      const bool ok = r < r_end && q_fok[0];
      const int rowterm = QM == QM_TABLE ? tb.q[e] : r * qLr;
      qv[e] = i2_ld(rsQ, ok ? (rowterm + q_fix[0]) * 4 : I2_OOB);
    }
  };
  auto ktail_fix = [&](f32x4 v, unsigned m) {
    v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f;
    v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
    return v;
  };
  // LDS position of reduction index k inside its 32-deep block: lane half h, MFMA step m consume
  // position 16h + m, and that position holds k = 2m + h -- the k-order of a slab (pairs (0,1), (2,3),
  // ...) is then the one igemm_kernel uses, so both kernels produce bit-identical sums.
  auto kpos = [](int k) { return (k & ~31) | ((k & 1) << 4) | ((k & 31) >> 1); };
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto put_kquad = [&](float* rowp, int sw, int q, f32x4 v) {   // logical k = 4q..4q+3 of a K-contiguous row (swizzle sw)
    // The (x, z) / (y, w) pairing needs register copies.  The empty asm pins them (and the wait for the
    // load that produced v) to this point of the MFMA chain: left alone, the compiler hoists the copies to
    // the loop back edge, where they wait for loads issued a few instructions earlier.
    asm volatile("" : "+v"(v));
    // unswizzled: positions (q >> 3) * 32 + 2 * (q & 7) and 16 further, i.e. chunks c0 and c0 + 4, halves (q & 1)
    const int c0 = ((q >> 3) << 3) | ((q & 7) >> 1);
    *(f32x2*)(rowp + ((c0 ^ sw) << 2) + 2 * (q & 1)) = f32x2{v.x, v.z};
    *(f32x2*)(rowp + (((c0 | 4) ^ sw) << 2) + 2 * (q & 1)) = f32x2{v.y, v.w};
  };
  auto store_p = [&](float* buf, int e, const f32x4 (&pv)[NVP], unsigned p_kb) {
    if (I2_ABLATE & 2) { asm volatile("" ::"v"(pv[e].x)); return; }
    if (PL == I2_P_ALONG_R) {
      put_kquad(buf + (p_l + e * PSTEP) * LDPK, swz(p_l + e * PSTEP), p_q, KTAIL ? ktail_fix(pv[e], p_kb) : pv[e]);
    } else {
      *(f32x4*)(buf + kpos(p_l + e * PSTEP) * LDPM + 4 * p_q) = pv[e];
    }
  };
  auto store_q = [&](float* buf, int e, const f32x4 (&qv)[NVQ], unsigned q_kb) {
    if (I2_ABLATE & 2) { asm volatile("" ::"v"(qv[e].x)); return; }
    float* Qs = buf + PSZ;
    if (QL == I2_Q_ALONG_R) {
      put_kquad(Qs + (q_l + e * QSTEP) * LDQK, swz(q_l + e * QSTEP), q_q, KTAIL ? ktail_fix(qv[e], q_kb) : qv[e]);
    } else {
      *(f32x4*)(Qs + kpos(q_l + e * QSTEP) * LDQN + 4 * q_q) = qv[e];
    }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[a][b][x] = 0.f;

  const bool do_ones = ONES && tl.z == 0;   // bias-gradient row: column sums of the staged Q slabs
  float csum = 0.f;

  // ---------------------------------------------------------------- pipeline
  // registers hold slab s+1 while the MFMAs of slab s run from LDS buffer s&1; inside that MFMA chain
  // the registers are written to buffer (s+1)&1 and immediately refilled with the loads of slab s+2.
  // Every piece of staging work sits in the shadow of one MFMA of the (dependent, in-order) chain;
  // per slab only the fragment reads after the barrier are exposed.
  const int nslab = (r_end - r_begin + BKT - 1) / BKT;
  // epilogue constants, requested here: read after the reduction loop they are one more scalar round trip (~0.5 us)
  // in front of the output-offset / mask loads of every tile
  const gf32 cbase = (gf32)(pb->c + (long)tl.y * pb->slab_stride);
  const int ldc = pb->ldc;
  const gci32 cT = (gci32)pb->c_tab_i;
  const gcf32 bias = (gcf32)pb->bias;
  const gcf32 rmask = (gcf32)pb->relu_mask;
  const gci32 mT = (gci32)pb->m_tab_i;
  const int act = pb->act;
  const float alpha = pb->act_alpha;
  const int accumulate = pb->accumulate;
  const float oscale = pb->out_scale != 0.f ? pb->out_scale : 1.f;
  // ... and so are the scatter offsets of the wide epilogue (conv backward-data: output rows through c_tab_i): fetched
  // after the loop they sit in front of the ReLU-mask loads that need them -- two dependent round trips per tile
  const bool cvec = (pb->vflags & VF_C_VEC) != 0;
  constexpr int EP_NC4 = BN / 4, EP_RSTEP = 256 / EP_NC4, EP_NQ = BM / EP_RSTEP;
  int ct_pre[EP_NQ], mt_pre[EP_NQ];
#pragma unroll
  for (int e = 0; e < EP_NQ; ++e) ct_pre[e] = mt_pre[e] = 0;
  if (cT && cvec) {
#pragma unroll
    for (int e = 0; e < EP_NQ; ++e) {
      const int i = i0 + t / EP_NC4 + e * EP_RSTEP;
      ct_pre[e] = cT[i < M ? i : 0];
    }
  }
  if (mT && cvec) {      // (the ReLU mask's own row offsets, IgemmProb.m_tab_i)
#pragma unroll
    for (int e = 0; e < EP_NQ; ++e) {
      const int i = i0 + t / EP_NC4 + e * EP_RSTEP;
      mt_pre[e] = mT[i < M ? i : 0];
    }
  }
  I2_STAMP(1);
  // prologue: slab 0 goes through its own registers so that slab 1 can be requested before slab 0 has
  // landed (one memory round trip less before the first MFMA)
  {
    f32x4 pv0[NVP], qv0[NVQ];
    unsigned p_kb0 = 0xfu, q_kb0 = 0xfu;
    Tabs tb0;
    tb0.tap = 0;
    fetch_tabs(r_begin, tb0);
    fetch_tabs(r_begin + BKT, tbA);
    fetch_tabs(r_begin + 2 * BKT, tbB);
#pragma unroll
    for (int e = 0; e < NVP; ++e) load_p(r_begin, e, pv0, p_kb0, tb0);
#pragma unroll
    for (int e = 0; e < NVQ; ++e) load_q(r_begin, e, qv0, q_kb0, tb0);
#pragma unroll
    for (int e = 0; e < NVP; ++e) load_p(r_begin + BKT, e, pv, p_kb, tbA);
#pragma unroll
    for (int e = 0; e < NVQ; ++e) load_q(r_begin + BKT, e, qv, q_kb, tbA);
    __builtin_amdgcn_sched_barrier(0);
    // same issue order as in the loop (data A, tables A, data B, tables B): the compiler's static wait
    // counts are the minimum over all paths into the loop head, the entry path included
    fetch_tabs(r_begin + 3 * BKT, tbA);          // for the first refill of set A (slab 3)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < NVP; ++e) load_p(r_begin + 2 * BKT, e, pvB, p_kbB, tbB);
#pragma unroll
    for (int e = 0; e < NVQ; ++e) load_q(r_begin + 2 * BKT, e, qvB, q_kbB, tbB);
    __builtin_amdgcn_sched_barrier(0);
    fetch_tabs(r_begin + 4 * BKT, tbB);          // for the first refill of set B (slab 4)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < NVP; ++e) store_p(lds, e, pv0, p_kb0);
#pragma unroll
    for (int e = 0; e < NVQ; ++e) store_q(lds, e, qv0, q_kb0);
  }
  __syncthreads();

  I2_STAMP(2);
  I2_TRACE(4);
  constexpr int NMF = 16 * FM * FN;            // MFMAs of one slab per wave
  constexpr int NST = NVP + NVQ;               // staged vectors per thread and slab
  static_assert(2 * NST + 1 <= NMF - 3 - (ONES ? 1 : 0), "staging work must fit into the MFMA gaps of a slab");
  auto slab_body = [&](int s, f32x4 (&pv)[NVP], f32x4 (&qv)[NVQ], unsigned& p_kb, unsigned& q_kb, Tabs& tb) {
    float* cur = lds + (s & 1) * BUF;
    float* nxt = lds + ((s + 1) & 1) * BUF;
    const int r2 = r_begin + (s + 3) * BKT;   // the set written to LDS now (slab s+1) is refilled with slab s+3
    // ---------------- fragments: LDS -> registers in four chunks of 4 k-steps.  Only chunk 0 is
    // read before the chain starts (all four waves leave the barrier together, so whatever is read
    // here queues on the LDS four deep); chunk c+1 is requested in the first gap of chunk c.
    const float* Ps = cur;
    const float* Qs = cur + PSZ;
    float av[FM][16], bv[FN][16];
    auto read_chunk = [&](int c) {
#pragma unroll
      for (int a = 0; a < FM; ++a) {
        const int row = (wm * FM + a) * 32 + li;
        if (PL == I2_P_ALONG_R) {
          const f32x4 v = *(const f32x4*)(Ps + row * LDPK + (((wk * 8 + 4 * lh + c) ^ swz(row)) << 2));
          av[a][4 * c] = v.x; av[a][4 * c + 1] = v.y; av[a][4 * c + 2] = v.z; av[a][4 * c + 3] = v.w;
        } else {
#pragma unroll
          for (int m = 4 * c; m < 4 * c + 4; ++m) av[a][m] = Ps[(wk * 32 + 16 * lh + m) * LDPM + row];
        }
      }
#pragma unroll
      for (int b = 0; b < FN; ++b) {
        const int col = (wn * FN + b) * 32 + li;
        if (QL == I2_Q_ALONG_R) {
          const f32x4 v = *(const f32x4*)(Qs + col * LDQK + (((wk * 8 + 4 * lh + c) ^ swz(col)) << 2));
          bv[b][4 * c] = v.x; bv[b][4 * c + 1] = v.y; bv[b][4 * c + 2] = v.z; bv[b][4 * c + 3] = v.w;
        } else {
#pragma unroll
          for (int m = 4 * c; m < 4 * c + 4; ++m) bv[b][m] = Qs[(wk * 32 + 16 * lh + m) * LDQN + col];
        }
      }
    };
    read_chunk(0);
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- MFMA chain with one piece of other work per gap
    constexpr int GPC = 4 * FM * FN;            // gaps per chunk
    int piece = 0;                              // compile-time after unrolling
#pragma unroll
    for (int g = 0; g < NMF; ++g) {
      const int m = g / (FM * FN), a = (g / FN) % FM, b = g % FN;
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][m], bv[b][m], acc[a][b], 0, 0, 0);
      if (g % GPC == 0 && g / GPC < 3) {
        read_chunk(g / GPC + 1);
      } else if (ONES && g == 1) {
        if (do_ones) {
          const int col = t & (BN - 1), rq = t / BN;          // 256/BN row groups
          constexpr int RPG = 32 / (256 / BN);
#pragma unroll
          for (int u = 0; u < RPG; ++u) csum += Qs[(rq * RPG + u) * LDQN + col];
        }
      } else {
        if (piece < NVP) store_p(nxt, piece, pv, p_kb);
        else if (piece < NST) store_q(nxt, piece - NVP, qv, q_kb);
        else if (piece < NST + NVP) load_p(r2, piece - NST, pv, p_kb, tb);
        else if (piece < 2 * NST) load_q(r2, piece - NST - NVP, qv, q_kb, tb);
        else if (piece == 2 * NST) fetch_tabs(r2 + 2 * BKT, tb);   // this set's next refill (two slabs on)
        ++piece;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(I2_ABLATE & 4)) __syncthreads();
  };
  // The loop body is exactly two slabs and has ONE path back to its head.  (With an `if (s + 1 < nslab)`
  // around the second slab the control-flow graph also contains "first half -> latch -> first half"; on
  // that path the loads just issued into set A are the youngest, so the compiler's static s_waitcnt
  // before the LDS stores of set A must cover it and becomes vmcnt(3..0): every iteration then waited
  // for the loads of set B, issued less than one slab earlier, instead of only for its own.)
  int s = 0;
  for (; s + 1 < nslab; s += 2) {
    slab_body(s, pv, qv, p_kb, q_kb, tbA);
    slab_body(s + 1, pvB, qvB, p_kbB, q_kbB, tbB);
  }
  if (s < nslab) slab_body(s, pv, qv, p_kb, q_kb, tbA);

  I2_STAMP(3);
  I2_TRACE(5);
  // ------------------------------------------------------------------ epilogue
  auto emit = [&](int i, int j, float a, float bj) {
    long off;
    if (cT) {
      const int o = cT[i];
      if (o < 0) return;
      off = (long)o + j;
    } else {
      off = (long)i * ldc + j;
    }
    float v = a * oscale + bj;
    if (accumulate) v += cbase[off];
    if (act == ACT_RELU) v = fmaxf(v, 0.f);
    else if (act == ACT_LEAKY) v = v > 0.f ? v : alpha * v;
    if (rmask) v = rmask[mT ? (long)mT[i] + j : off] > 0.f ? v : alpha * v;
    cbase[off] = v;
  };

  // The 16 results of a lane go to 16 different rows.  Output offsets (scatter table), ReLU masks and
  // accumulate operands are fetched for all of them before the first store: the stores may alias the
  // loads as far as the compiler knows, so interleaving them would serialise 16 dependent round trips.
  auto emit16 = [&](const int (&iv)[16], int j, const float (&val)[16], float bj) {
    long off[16];
    bool ok[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      ok[x] = iv[x] < M;
      const int ic = ok[x] ? iv[x] : 0;
      if (cT) {
        const int o = cT[ic];
        ok[x] = ok[x] && o >= 0;
        off[x] = (long)(o >= 0 ? o : 0) + j;
      } else {
        off[x] = (long)ic * ldc + j;
      }
    }
    float mk[16], prev[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      mk[x] = (rmask && ok[x]) ? rmask[mT ? (long)mT[iv[x]] + j : off[x]] : 1.f;
      prev[x] = (accumulate && ok[x]) ? cbase[off[x]] : 0.f;
    }
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      float v = val[x] * oscale + bj + prev[x];
      if (act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (act == ACT_LEAKY) v = v > 0.f ? v : alpha * v;
      v = mk[x] > 0.f ? v : alpha * v;
      if (ok[x]) cbase[off[x]] = v;
    }
  };

  // Row-per-lane dword stores drain at ~7 B/clk/CU (store-issue bound): a 64x64 tile then spends
  // longer in its epilogue than in three slabs of MFMAs.  When the output allows 16-byte accesses
  // (VF_C_VEC) the tile goes through LDS once and leaves as dwordx4 stores, 4 columns per lane.
  constexpr int LDC_S = BN + 4;
  auto wide_out = [&](float* tile) {   // tile[BM][LDC_S] holds acc (already summed over the k split)
    constexpr int NC4 = BN / 4, RSTEP = 256 / NC4, NQ = BM / RSTEP;
    static_assert(NC4 == EP_NC4 && NQ == EP_NQ, "prefetched scatter offsets follow the wide epilogue's row mapping");
    const int c4 = t % NC4, rl0 = t / NC4;
    const int j = j0 + 4 * c4;
    if (j >= N) return;
    f32x4 bj = {0.f, 0.f, 0.f, 0.f};
    if (bias) bj = *(const GRL_GLOBAL f32x4*)(bias + j);
    long off[NQ];
    bool ok[NQ];
#pragma unroll
    for (int e = 0; e < NQ; ++e) {
      const int i = i0 + rl0 + e * RSTEP;
      ok[e] = i < M;
      const int ic = ok[e] ? i : 0;
      if (cT) {
        const int o = ct_pre[e];
        ok[e] = ok[e] && o >= 0;
        off[e] = (long)(o >= 0 ? o : 0) + j;
      } else {
        off[e] = (long)ic * ldc + j;
      }
    }
    f32x4 mk[NQ], prev[NQ];
#pragma unroll
    for (int e = 0; e < NQ; ++e) {
      mk[e] = f32x4{1.f, 1.f, 1.f, 1.f};
      prev[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rmask && ok[e]) mk[e] = *(const GRL_GLOBAL f32x4*)(rmask + (mT ? (long)mt_pre[e] + j : off[e]));
      if (accumulate && ok[e]) prev[e] = *(const GRL_GLOBAL f32x4*)(cbase + off[e]);
    }
#pragma unroll
    for (int e = 0; e < NQ; ++e) {
      f32x4 v = *(const f32x4*)(tile + (rl0 + e * RSTEP) * LDC_S + 4 * c4);
      v = v * oscale + bj + prev[e];
      if (act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      else if (act == ACT_LEAKY) {
        v.x = v.x > 0.f ? v.x : alpha * v.x; v.y = v.y > 0.f ? v.y : alpha * v.y;
        v.z = v.z > 0.f ? v.z : alpha * v.z; v.w = v.w > 0.f ? v.w : alpha * v.w;
      }
      v.x = mk[e].x > 0.f ? v.x : alpha * v.x; v.y = mk[e].y > 0.f ? v.y : alpha * v.y;
      v.z = mk[e].z > 0.f ? v.z : alpha * v.z; v.w = mk[e].w > 0.f ? v.w : alpha * v.w;
      if (ok[e]) *(GRL_GLOBAL f32x4*)(cbase + off[e]) = v;
    }
  };

  if (WK == 1) {
    if (cvec) {
      // all slabs are consumed (trailing barrier of the loop): the staging LDS is free
      float* tile = lds;
#pragma unroll
      for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
          for (int x = 0; x < 16; ++x) {
            const int rl = (wm * FM + a) * 32 + (x & 3) + 8 * (x >> 2) + 4 * lh;
            const int cl = (wn * FN + b) * 32 + li;
            tile[rl * LDC_S + cl] = acc[a][b][x];
          }
      __syncthreads();
      wide_out(tile);
      if (ONES && do_ones) __syncthreads();   // the column sums below reuse lds[0..255]
    } else {
#pragma unroll
    for (int b = 0; b < FN; ++b) {
      const int j = j0 + (wn * FN + b) * 32 + li;
      if (j >= N) continue;
      const float bj = bias ? bias[j] : 0.f;
#pragma unroll
      for (int a = 0; a < FM; ++a) {
        int iv[16];
        float val[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) {
          iv[x] = i0 + (wm * FM + a) * 32 + (x & 3) + 8 * (x >> 2) + 4 * lh;
          val[x] = acc[a][b][x];
        }
        emit16(iv, j, val, bj);
      }
    }
    }
    if (ONES && do_ones) {
      lds[t] = csum;
      __syncthreads();
      if (t < BN) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 256 / BN; ++g) s += lds[g * BN + t];
        const int j = j0 + t;
        if (j < N) emit(ones_i, j, s, 0.f);
      }
    }
  } else {
    // the waves hold partial sums over disjoint k ranges: add them in wave order through LDS
    float* red = lds;
    constexpr int LDR = BN + 4;
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
      for (int b = 0; b < FN; ++b)
#pragma unroll
        for (int x = 0; x < 16; ++x) {
          const int rl = (wm * FM + a) * 32 + (x & 3) + 8 * (x >> 2) + 4 * lh;
          const int cl = (wn * FN + b) * 32 + li;
          red[(wk * BM + rl) * LDR + cl] = acc[a][b][x];
        }
    __syncthreads();
    if (cvec) {
      // sum the partials in place (each thread owns the quads it will store), then the wide path
      constexpr int NC4 = BN / 4, RSTEP = 256 / NC4, NQ = BM / RSTEP;
      const int c4 = t % NC4, rl0 = t / NC4;
#pragma unroll
      for (int e = 0; e < NQ; ++e) {
        const int rl = rl0 + e * RSTEP;
        f32x4 sacc = *(const f32x4*)(red + rl * LDR + 4 * c4);
#pragma unroll
        for (int w = 1; w < WK; ++w) sacc += *(const f32x4*)(red + (w * BM + rl) * LDR + 4 * c4);
        *(f32x4*)(red + rl * LDR + 4 * c4) = sacc;
      }
      wide_out(red);
    } else {
      static_assert(BM * BN / 256 <= 16, "reduction epilogue handles up to 16 outputs per thread");
      // thread t owns column cl = t % BN of rows rl = t / BN + e * (256 / BN)
      const int cl = t % BN, j = j0 + cl;
      int iv[16];
      float val[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        iv[e] = M;   // inactive
        val[e] = 0.f;
        if (e < BM * BN / 256) {
          const int rl = t / BN + e * (256 / BN);
          float sacc = 0.f;
#pragma unroll
          for (int w = 0; w < WK; ++w) sacc += red[(w * BM + rl) * LDR + cl];
          iv[e] = i0 + rl;
          val[e] = sacc;
        }
      }
      if (j < N) emit16(iv, j, val, bias ? bias[j] : 0.f);
    }
  }
  I2_STAMP(4);
}


template <int PL, int QL, int PM, int QM, int CFG, int FLAGS>
__global__ __launch_bounds__(256) void igemm2_kernel(const IgemmProb* __restrict__ probs, const int4* __restrict__ tiles) {
  __shared__ __attribute__((aligned(16))) float lds[I2Lds<PL, QL, CFG>::value];
#ifdef GRL_TILE_TRACE
  const unsigned long long t0 = wall_clock64();
#endif
  // `probs` holds one descriptor copy per workgroup (add_launch): the tile entry and the descriptor are fetched side
  // by side instead of one after the other -- one dependent memory round trip less before the first operand load
  igemm2_tile<PL, QL, PM, QM, CFG, FLAGS>(probs + blockIdx.x, tiles[blockIdx.x], lds);
#ifdef GRL_TILE_TRACE
  i2_trace_record(probs + blockIdx.x, tiles[blockIdx.x], t0, CFG);
#endif
}

// Two kinds of tiles in one launch: blocks [0, n_a) run kind A -- the stage the launch exists for, on the critical
// path of the update -- and the remaining blocks kind B, independent work (weight gradients whose operands are
// already complete) that fills the SIMD time A's few tiles per CU leave idle and rides on A's launch ramp.
template <int PLa, int QLa, int PMa, int QMa, int CFGa, int FLa, int PLb, int QLb, int PMb, int QMb, int CFGb, int FLb>
__global__ __launch_bounds__(256) void igemm2_pair_kernel(const IgemmProb* __restrict__ pa, const int4* __restrict__ ta, int n_a,
                                                         const IgemmProb* __restrict__ pb, const int4* __restrict__ tb) {
  constexpr int LA = I2Lds<PLa, QLa, CFGa>::value, LB = I2Lds<PLb, QLb, CFGb>::value;
  __shared__ __attribute__((aligned(16))) float lds[LA > LB ? LA : LB];
  const int b = (int)blockIdx.x;
  const bool is_a = b < n_a;
  const int k = is_a ? b : b - n_a;
#ifdef GRL_TILE_TRACE
  const unsigned long long t0 = wall_clock64();
#endif
  if (is_a) igemm2_tile<PLa, QLa, PMa, QMa, CFGa, FLa>(pa + k, ta[k], lds);
  else igemm2_tile<PLb, QLb, PMb, QMb, CFGb, FLb>(pb + k, tb[k], lds);
#ifdef GRL_TILE_TRACE
  i2_trace_record(is_a ? pa + k : pb + k, is_a ? ta[k] : tb[k], t0, is_a ? CFGa : CFGb);
#endif
}

#endif  // GRL_HOSTEMU
#endif  // GRL_GEMM_TYPES_ONLY

}  // namespace grl
