// heads_mfma.h -- the actor / critic MLP heads of one SAC update, forward AND backward, as ONE launch on the
// f32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).
//
// What runs here (stable-baselines sac/policies.py make_actor / make_critics as wired by
// /root/reference/manipulation_main/training/sb_helper.py:87-96, losses of SAC.setup_model):
// everything between the layer-0 feature products (`u`, left by the heads_l0 GEMM) and the layer-0 gradients
// the feature-gradient / weight-gradient GEMMs consume.  heads_kernels.h does this with two launches of
// VALU chains (forward, then backward) that cost ~2.5 us per 64x64 layer stage; here
//
//   * a workgroup owns 16 batch rows -- the M of a 16x16x4 MFMA -- and a layer stage is 16 MFMAs per wave
//     (4 waves x 1 or 2 column blocks of 16: layer widths up to 64 or up to 128; wider heads stay on heads_kernels.h): weights go
//     global -> registers (B operand, no LDS staging), the activations of the 16 rows sit in LDS row-major with
//     a +4 pad (A operand: 16-byte reads, no conflicts);
//   * forward activations stay in REGISTERS in the MFMA result layout, which is exactly where the ReLU mask
//     of the backward stage needs them;
//   * every backward chain is made self-contained by recomputing the few forward heads it depends on
//     (row-local, ~0.3 us each), so forward and backward need no launch boundary between them:
//
//       type 0  pi -> sample -> qf1(s,pi) -> backward of qf1(s,pi) (d = -1/B) -> d a -> sample backward -> pi backward
//       type 1  pi -> sample -> qf1(s,pi), qf2(s,pi) -> vf -> d V = (V - (min q - alpha logp)) / B -> vf backward
//       type 2  target vf -> qf1(s,a) -> d qf1 = (qf1 - (r + (1-d) gamma V')) / B -> qf1 backward
//       type 3  target vf -> qf2(s,a) -> ... -> qf2 backward
//
//     each tensor the rest of the update reads is written by exactly one type.
//
// The k-order of a stage is the MFMA's (k = (W/4) q + s for lane quarter q, step s); the per-layer GEMM path
// (GRL_NO_FUSED_HEADS=1) and heads_kernels.h (GRL_NO_HEADS_MFMA=1) sum k sequentially -- all three are checked
// against the oracle with the same tolerances (tests/test_gpu_parity.py).
#pragma once
#include "heads_kernels.h"
#include "igemm2.h"

namespace grl {

struct HeadsFusedArgs {
  HtHead h[7];            // 0 pi, 1 vf, 2 qf1(a), 3 qf2(a), 4 target vf, 5 qf1(pi), 6 qf2(pi); g pointers on 0..3
  int B, A;
  const float* eps;       // [B, A]
  float* pi_a; float* logp; float* ent;
  const float* log_ent_coef;
  float* da_pi;           // [B, A]
  float* dmu; float* dls; // [B, A], row stride ld_dm
  int ld_dm;
  const float* rew; const float* done; float gamma;
  float* d_out[5];        // [B] (stride ld_d): 1 d_v, 2 d_qf1, 3 d_qf2, 4 d_qf1_pi
  int ld_d;
  unsigned long long* stamps;   // development aid (GRL_TUNE=heads_stamps=1): [4 types][32] wall-clock stamps of row block 0
  // multi-update calls that prefetch the next minibatch (engine.hip "prefetch"): this launch sits between the last
  // launch of the previous update (which read the Adam step size and drew the indices of THIS update with counter
  // rng_step + 1) and the last launch of this one, so one thread opens the update here instead of in the gather
  DevScalars* sc;
  int tick;          // fix the Adam step size of this update and advance the beta powers (adam_tick_device)
  int rng_advance;   // rng_step += 1: this update's minibatch was drawn a launch chain ago with rng_step + 1
};

#ifndef GRL_HEADS_TYPES_ONLY
#ifdef GRL_HOSTEMU
#include "heads_mfma_ref1.h"   // tests/hostemu: the emulation build only
#else  // ------------------------------------------------------------------------------------ device

typedef float hm_f4 __attribute__((ext_vector_type(4)));
// pointers read from the argument block are generic to the compiler: flat loads count on lgkmcnt too, so every LDS
// wait would drain the prefetched weights.  Typed as global (address space 1) they become global_load / global_store.
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter, i.e. it
// would wait at every stage for the operands requested a head ahead
#define HM_SYNC()                                                         \
  do {                                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");       \
    __builtin_amdgcn_s_barrier();                                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");       \
  } while (0)
#define HM_G(p) ((gcf32)(p))
#define HM_GW(p) ((gf32)(p))

// W: the padded layer width of the launch -- 64 (one 16-column block per wave) or 128 (two: columns 16 w + c and
// 64 + 16 w + c; trained_models/table_clearing/SAC_real_2m_buffer_128/config.yaml runs layers [128, 128]).  A stage over
// a hidden layer reduces over K = W: the lane of quarter q holds k = (W/4) q .. (W/4) q + W/4 - 1.  Stages whose
// reduction is an action vector or a head's outputs (K <= 64) keep the 64-wide layout whatever W is.
enum { HM_AW = 64, HM_KA = HM_AW / 4, HM_ALD = HM_AW + 4, HM_MAXA = 32 };
// NK: register steps of a stage with few outputs (a head's outputs, the action gradient: N <= 16 in the FAST shapes).
// At W = 128 such a stage is split over the four waves along k (wave w reduces k = 32 w .. 32 w + 31, 8 steps) and the
// partial tiles are added through LDS in wave order: a quarter of the registers and of the dependent MFMA chain.
template <int W, bool FAST> struct HmDim {
  static constexpr int NB = W / 64, K4 = W / 4, LD = W + 4;
  static constexpr bool KSPLIT = W == 128 && FAST;
  static constexpr int NK = KSPLIT ? K4 / 4 : K4;
};

// B operand of a stage: element (k, n) of a [K, N] matrix served from one or two weight tensors; zero outside.
struct HmB {
  const float* p0; const float* p1;
  int K, N;        // logical extents
  int sk, sn;      // strides of k and n inside a part
  int split;       // first index served by p1 -- along n (on_k == 0) or along k (on_k == 1); >= extent: one part
  int on_k;
  const float* bias0; const float* bias1;   // optional bias of the stage's outputs (split along n like p0 / p1)
};
// operands of every stage of one head, requested in one burst (static slots)
template <int W, bool FAST> struct HmFw {     // forward
  typedef HmDim<W, FAST> D;
  float a0[D::NB][HM_KA], hid[GRL_MAX_LAYERS][D::NB][D::K4], out[D::NK];
  float b0[D::NB], bh[GRL_MAX_LAYERS][D::NB], bo;
};
template <int W, bool FAST> struct HmBw {     // backward
  typedef HmDim<W, FAST> D;
  float out[D::NB][HM_KA], hid[GRL_MAX_LAYERS][D::NB][D::K4], da[D::NK];
  float ow[D::NB];
};

template <int W> struct HmLds {
  float z[2][HT_RB][W + 4];          // activations / gradients entering the next stage, row-major (A operand)
  float o[HT_RB][HM_ALD];            // head outputs (mu | log_std) / output gradients (dmu | dls) (A operand)
  float pi[HT_RB][HM_ALD];           // action part of a head's input: sampled action or minibatch action (A operand)
  float ls[HT_RB][HM_MAXA];          // raw log_std
  float mu[HT_RB][HM_MAXA];
  float da[HT_RB][HM_MAXA];          // d loss / d action
  float eps[HT_RB][HM_MAXA];         // policy noise of the rows
  float u[4][HT_RB][W + 4];          // layer-0 feature partial sums of the chain's heads (added up), fetched as coalesced quads
  float part[4][HT_RB][HT_RB + 1];   // partial tiles of a stage split over the waves along k (HmDim::KSPLIT)
  float sv[10][HT_RB];               // per-row scalars: 0 qf1_pi, 1 qf2_pi, 2 logp, 3 entropy, 4 v_tgt, 5 d, 6 reward, 7 done, 8 v / q
  float alpha;                       // exp(log_ent_coef)
  float pad_[3];
  HeadsFusedArgs args;               // the argument block, copied once: field reads are LDS reads, not scalar-cache misses
};

// Latency rules this kernel is built on (measured with the wall-clock stamps of GRL_TUNE=heads_stamps=1, MI355X):
// every dependent global access costs ~1.2 us here, an MFMA stage ~0.3 us.  So (1) the argument block is copied to
// LDS in one coalesced read, (2) the per-row inputs and the layer-0 partial sums of every head of the chain are
// requested in one burst of few, wide loads, (3) the register operands of ALL stages of a head are requested one
// head ahead of their use -- but never more than ~60 vector loads in flight: the counter the waits are expressed
// in has 6 bits, beyond it the wave stalls at issue and a wait for the oldest load drains younger ones too --
// and (4) no global store is issued before the end of the chain: gfx950 counts stores and loads on the same
// in-order counter, so a wait for a young load is also a wait for the acknowledgement of every older store.
// FAST: the reference's shapes -- two hidden layers of W units in every head (gripper_grasp.yaml:81 `layers: [64, 64]`,
// SAC_real_2m_buffer_128/config.yaml `layers: [128, 128]`) and a batch that is a multiple of 16.  The kernel is bound
// by instruction issue, and with the widths, the layer count and the row predicate known at compile time most of its
// address / predicate arithmetic folds away.
// TG: the chain types this copy of the body serves -- -1: all four, told apart at run time (the 64-wide kernels: one body);
// 0 / 1: that type; 2: types 2 and 3.  The 128-wide kernel is three copies behind a branch on blockIdx.y: as ONE body it
// needed more than the 512 registers a lane has (20 spilled), although no single chain needs more than ~340 -- the operand
// sets of all chains were live across the type branches.
template <int W, bool FAST, int TG>
__device__ __forceinline__ void heads_fused_body(const HeadsFusedArgs* __restrict__ ap, HmLds<W>& s) {
  static_assert(W == 64 || W == 128, "layer widths above 128 run on heads_kernels.h");
  typedef HmDim<W, FAST> D;
  constexpr int NB = D::NB, K4 = D::K4, LD = D::LD, KA = HM_KA, ALD = HM_ALD, NK = D::NK;
  constexpr bool KSPLIT = D::KSPLIT;
  const int t = threadIdx.x, w = t >> 6, l = t & 63, c = l & 15, q = l >> 4;
  // columns owned by this lane in a stage output (result layout: rows 4q .. 4q+3).  The NB column blocks of a wave are
  // INTERLEAVED (block b = columns NB c + b of the wave's 16 NB): the lane's columns are neighbours in memory, so a
  // row-major weight row serves all of them with one 8-byte load and the activations go to LDS as one 8-byte store
  int nb[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) nb[b] = 16 * NB * w + NB * c + b;
  const int n = 16 * w + c;       // column of a stage with few outputs (one block)
  // ---- round trip 1: the argument block (2.5 KB) in one coalesced read
  {
    typedef const GRL_GLOBAL uint32_t* gcu32;
    constexpr int NW = (int)(sizeof(HeadsFusedArgs) / 4), NE = (NW + 255) / 256;
    uint32_t tmp[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) tmp[e] = t + 256 * e < NW ? ((gcu32)ap)[t + 256 * e] : 0u;
    // every A operand is read over the padded width: start from finite (zero) LDS (everything in front of the arguments)
    for (int x = t; x < (int)(offsetof(HmLds<W>, args) / 4); x += 256) ((float*)&s)[x] = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e)
      if (t + 256 * e < NW) ((uint32_t*)&s.args)[t + 256 * e] = tmp[e];
  }
  HM_SYNC();
  const HeadsFusedArgs& a = s.args;
  if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0 && a.sc) {
    if (a.rng_advance) a.sc->rng_step += 1;
    if (a.tick) adam_tick_device(a.sc);
  }
  const int row0 = blockIdx.x * HT_RB, type = TG < 0 ? (int)blockIdx.y : TG < 2 ? TG : (int)(blockIdx.y | 2u), B = a.B, A = a.A;
  const float invB = 1.f / (float)B;
  auto LY = [](int v) { return FAST ? 2 : v; };          // hidden layers of a head
  auto WD = [](int v) { return FAST ? (int)W : v; };      // a hidden width
  auto ROW = [&](int r) { return FAST ? true : (row0 + r) < B; };   // row r of this block exists
  int stamp_k = 0;
  auto stamp = [&]() {
    if (a.stamps && blockIdx.x == 0 && t == 0 && stamp_k < 32) a.stamps[type * 32 + stamp_k] = wall_clock64();
    ++stamp_k;
  };
  stamp();

  // ---------------------------------------------------------------- primitives
  // Branch-free operand fetch: buffer loads through a descriptor on the part's base; an element outside the logical
  // matrix (or served by the other part) gets an out-of-range offset and comes back as zero.
  auto ldw = [&](__amdgpu_buffer_rsrc_t rs, int off_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off_bytes, 0, 0));
  };
  // The kernel is bound by instruction issue (a wave64 VALU instruction occupies its SIMD for 4 cycles; ~2000
  // address / predicate instructions were 3.6 us): the common shapes take paths with no per-element VALU work.
  // bw: the KS register steps of column nc -- the lane of quarter q holds k = KS q + step
  auto load_b = [&](auto& bw, const HmB& d, const int nc) {
    constexpr int KS = (int)(sizeof(bw) / sizeof(float));
    // one descriptor based on the lower of the two parts (both live in the parameter block): one load per element
    const float* base = (d.p1 && d.p1 < d.p0) ? d.p1 : d.p0;
    const __amdgpu_buffer_rsrc_t rs = i2_rsrc(base);
    const int off0 = (int)(d.p0 - base), off1 = d.p1 ? (int)(d.p1 - base) : 0;
    if (d.K >= 4 * KS && !d.on_k) {
      // full-depth stage: every k of the padded width exists; the lane's part / validity is fixed -> one lane
      // offset, the k step rides on the scalar offset (or on 16-byte loads when k is contiguous)
      const bool first = nc < d.split;
      const int voff = nc < d.N ? ((first ? off0 : off1) + KS * q * d.sk + (first ? nc : nc - d.split) * d.sn) * 4 : I2_OOB;
      if (d.sk == 1) {
#pragma unroll
        for (int j = 0; j < KS / 4; ++j) {
          const hm_f4 v = i2_ld(rs, voff + 16 * j);
          bw[4 * j] = v.x; bw[4 * j + 1] = v.y; bw[4 * j + 2] = v.z; bw[4 * j + 3] = v.w;
        }
      } else {
        const int stepb = d.sk * 4;
#pragma unroll
        for (int sI = 0; sI < KS; ++sI)
          bw[sI] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, sI * stepb, 0));
      }
      return;
    }
    const int steps = d.K < KS ? d.K : KS;     // K <= KS lives entirely in lane quarter 0: later steps are all zero
#pragma unroll
    for (int sI = 0; sI < KS; ++sI) {
      bw[sI] = 0.f;
      if (sI < steps) {
        const int k = KS * q + sI;
        const bool ok = k < d.K && nc < d.N;
        const bool first = d.on_k ? k < d.split : nc < d.split;
        const int kk = (d.on_k && !first) ? k - d.split : k, nn = (!d.on_k && !first) ? nc - d.split : nc;
        bw[sI] = ldw(rs, ok ? ((first ? off0 : off1) + kk * d.sk + nn * d.sn) * 4 : I2_OOB);
      }
    }
  };
  // the KS register steps of ALL the lane's columns at once.  Two neighbouring columns of a row-major one-part matrix
  // with an even row stride are one 8-byte load per step (the hidden layers of the W = 128 shape: 32 loads per lane
  // and stage instead of 64)
  auto load_b_cols = [&](auto& bw, const HmB& d) {
    constexpr int KS = (int)(sizeof(bw[0]) / sizeof(float));
    if (NB == 2 && !d.on_k && d.sn == 1 && d.split >= d.N && !(d.sk & 1) && !(d.N & 1) && !((uintptr_t)d.p0 & 7)) {
      typedef float hm_f2 __attribute__((ext_vector_type(2)));
      const __amdgpu_buffer_rsrc_t rs = i2_rsrc(d.p0);
      const int nc = nb[0];                      // even; nb[1] = nc + 1 exists exactly when nc does (N even)
      const int stepb = d.sk * 4;
      if (d.K >= 4 * KS) {
        const int voff = nc < d.N ? (KS * q * d.sk + nc) * 4 : I2_OOB;
#pragma unroll
        for (int sI = 0; sI < KS; ++sI) {
          const hm_f2 v = __builtin_bit_cast(hm_f2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, sI * stepb, 0));
          bw[0][sI] = v.x; bw[NB - 1][sI] = v.y;
        }
        return;
      }
      const int steps = d.K < KS ? d.K : KS;
#pragma unroll
      for (int sI = 0; sI < KS; ++sI) {
        bw[0][sI] = 0.f; bw[NB - 1][sI] = 0.f;
        if (sI < steps) {
          const int k = KS * q + sI;
          const hm_f2 v = __builtin_bit_cast(hm_f2, __builtin_amdgcn_raw_buffer_load_b64(rs, (k < d.K && nc < d.N) ? (k * d.sk + nc) * 4 : I2_OOB, 0, 0));
          bw[0][sI] = v.x; bw[NB - 1][sI] = v.y;
        }
      }
      return;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) load_b(bw[b], d, nb[b]);
  };
  auto load_bias = [&](const float* b0, const float* b1, int split, int N, int nc) {   // split along n like the B operand
    if (nc >= N) return 0.f;
    return nc < split ? HM_G(b0)[nc] : HM_G(b1)[nc - split];
  };
  // B operand of a stage with few outputs (column nc of this lane).  KSPLIT: wave w owns k = (W/4) w .. + W/4 - 1, the
  // lane of quarter q the NK steps k = (W/4) w + NK q + step; otherwise the K4 steps of load_b
  auto load_b_few = [&](float (&bw)[NK], const HmB& d, const int nc) {
    if (!KSPLIT) { load_b(bw, d, nc); return; }
    const float* base = (d.p1 && d.p1 < d.p0) ? d.p1 : d.p0;
    const __amdgpu_buffer_rsrc_t rs = i2_rsrc(base);
    const int off0 = (int)(d.p0 - base), off1 = d.p1 ? (int)(d.p1 - base) : 0;
    const bool first = nc < d.split;                 // (split along n; these stages never split along k)
    const int k0 = K4 * w + NK * q;
    const int voff = nc < d.N ? ((first ? off0 : off1) + k0 * d.sk + (first ? nc : nc - d.split) * d.sn) * 4 : I2_OOB;
    if (d.sk == 1) {
#pragma unroll
      for (int j = 0; j < NK / 4; ++j) {
        const hm_f4 v = i2_ld(rs, voff + 16 * j);
        bw[4 * j] = v.x; bw[4 * j + 1] = v.y; bw[4 * j + 2] = v.z; bw[4 * j + 3] = v.w;
      }
    } else {
      const int stepb = d.sk * 4;
#pragma unroll
      for (int sI = 0; sI < NK; ++sI)
        bw[sI] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, sI * stepb, 0));
    }
  };
  // A operand of a stage: lane (row c, quarter q) holds k = KS q .. KS q + KS - 1 of its row
  auto read_a = [&](auto& av, const float* zrow) {
    constexpr int KS = (int)(sizeof(av) / sizeof(float));
#pragma unroll
    for (int j = 0; j < KS / 4; ++j) {
      const hm_f4 v = *(const hm_f4*)(zrow + KS * q + 4 * j);
      av[4 * j] = v.x; av[4 * j + 1] = v.y; av[4 * j + 2] = v.z; av[4 * j + 3] = v.w;
    }
  };
  // acc[b] += zin[16 rows][W] . bw[b]      (every column block of the wave shares the A operand)
  auto mma = [&](hm_f4 (&acc)[NB], const float (*zin)[LD], const float (&bw)[NB][K4]) {
    float av[K4];
    read_a(av, &zin[c][0]);
#pragma unroll
    for (int sI = 0; sI < K4; ++sI)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sI], bw[b][sI], acc[b], 0, 0, 0);
  };
  // a stage with few outputs (head outputs, action gradients): out(row, col) = sum_k zin[row][k] bw(k, col) + bias, handed
  // to put(row, col, value) for the (row, col) pairs of this lane.  One column block per wave (N <= 64) -- or, KSPLIT,
  // N <= 16 with the reduction split over the waves and the four partial tiles added in wave order
  auto stage_few = [&](const float (*zin)[LD], const float (&bw)[NK], float bias, auto&& put) {
    hm_f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (!KSPLIT) {
      float av[NK];
      read_a(av, &zin[c][0]);
#pragma unroll
      for (int sI = 0; sI < NK; ++sI) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sI], bw[sI], acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) put(4 * q + i, n, acc[i] + bias);
      HM_SYNC();
      return;
    }
    float av[NK];
#pragma unroll
    for (int j = 0; j < NK / 4; ++j) {
      const hm_f4 v = *(const hm_f4*)(&zin[c][K4 * w + NK * q + 4 * j]);
      av[4 * j] = v.x; av[4 * j + 1] = v.y; av[4 * j + 2] = v.z; av[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int sI = 0; sI < NK; ++sI) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sI], bw[sI], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) s.part[w][4 * q + i][c] = acc[i];
    HM_SYNC();
    {
      const int r = t >> 4;                 // (t & 15 == c: the bias this lane fetched is the one of its column here)
      put(r, c, ((s.part[0][r][c] + s.part[1][r][c]) + s.part[2][r][c]) + s.part[3][r][c] + bias);
    }
    HM_SYNC();
  };
  // acc[b] += ain[16 rows][64] . bw[b]     (reduction over an action vector or a head's outputs: K <= 64)
  auto mma_a = [&](hm_f4 (&acc)[NB], const float (*ain)[ALD], const float (&bw)[NB][KA]) {
    float av[KA];
    read_a(av, &ain[c][0]);
#pragma unroll
    for (int sI = 0; sI < KA; ++sI)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sI], bw[b][sI], acc[b], 0, 0, 0);
  };

  // ---- operand prefetch of a whole head
  // (the head descriptors sit in LDS: every lambda first takes a local copy, so that the field reads are issued
  //  together and waited for once -- read where they are used, each one is a ds_read + wait in a dependent chain)
  auto prefetch_fwd = [&](const HtHead& h_, HmFw<W, FAST>& f) {
    const HtHead h = h_;
#pragma unroll
    for (int b = 0; b < NB; ++b) f.b0[b] = nb[b] < WD(h.H0) ? HM_G(h.b0)[nb[b]] : 0.f;
    if (h.n_xa > 0) load_b_cols(f.a0, HmB{h.w0a, nullptr, h.n_xa, WD(h.H0), WD(h.H0), 1, INT_MAX, 0, nullptr, nullptr});
#pragma unroll
    for (int ly = 1; ly < GRL_MAX_LAYERS; ++ly)
      if (ly < LY(h.L)) {
        load_b_cols(f.hid[ly], HmB{h.w[ly], nullptr, WD(h.hid[ly - 1]), WD(h.hid[ly]), WD(h.hid[ly]), 1, INT_MAX, 0, nullptr, nullptr});
#pragma unroll
        for (int b = 0; b < NB; ++b) f.bh[ly][b] = nb[b] < WD(h.hid[ly]) ? HM_G(h.b[ly])[nb[b]] : 0.f;
      }
    const int NO = h.n_out * h.out_dim;
    int HLf = WD(h.hid[0]);                       // width of the last hidden layer (static indices: no scratch copy)
#pragma unroll
    for (int ly = 1; ly < GRL_MAX_LAYERS; ++ly) HLf = (ly == LY(h.L) - 1) ? WD(h.hid[ly]) : HLf;
    const int ncf = KSPLIT ? c : n;               // the lane's column in a stage with few outputs
    load_b_few(f.out, HmB{h.ow[0], h.n_out > 1 ? h.ow[1] : nullptr, HLf, NO, h.out_dim, 1,
                          h.n_out > 1 ? h.out_dim : INT_MAX, 0, nullptr, nullptr}, ncf);
    f.bo = load_bias(h.ob[0], h.n_out > 1 ? h.ob[1] : h.ob[0], h.n_out > 1 ? h.out_dim : INT_MAX, NO, ncf);
  };
  auto prefetch_bwd = [&](const HtHead& h_, HmBw<W, FAST>& g, bool rank1, bool want_da) {
    const HtHead h = h_;
    const int L = LY(h.L);
    int HL = WD(h.hid[0]);
#pragma unroll
    for (int ly = 1; ly < GRL_MAX_LAYERS; ++ly) HL = (ly == L - 1) ? WD(h.hid[ly]) : HL;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      g.ow[b] = 0.f;
      if (rank1) g.ow[b] = nb[b] < HL ? HM_G(h.ow[0])[nb[b]] : 0.f;
      else   // element (k, n) = ow[k / out_dim][n * out_dim + k % out_dim]
        load_b(g.out[b], HmB{h.ow[0], h.n_out > 1 ? h.ow[1] : nullptr, h.n_out * h.out_dim, HL, 1, h.out_dim,
                             h.n_out > 1 ? h.out_dim : INT_MAX, 1, nullptr, nullptr}, nb[b]);
    }
#pragma unroll
    for (int ly = 1; ly < GRL_MAX_LAYERS; ++ly)
      if (ly < L) {  // g_{ly-1} = g_ly . W_ly^T: element (k, m) = w[ly][m * hid[ly] + k]
#pragma unroll
        for (int b = 0; b < NB; ++b)
          load_b(g.hid[ly][b], HmB{h.w[ly], nullptr, WD(h.hid[ly]), WD(h.hid[ly - 1]), 1, WD(h.hid[ly]), INT_MAX, 0, nullptr, nullptr}, nb[b]);
      }
    if (want_da) load_b_few(g.da, HmB{h.w0a, nullptr, WD(h.H0), h.n_xa, 1, WD(h.H0), INT_MAX, 0, nullptr, nullptr}, KSPLIT ? c : n);
  };

  int cur = 0;   // s.z[cur] holds the input of the next MFMA stage

  // forward of one head.  zsv[l][b][i]: activations kept in the result layout (row 4q+i, column nb[b]).  u0: layer-0
  // feature partial sums of the rows (already added up); the action part of the input is s.pi (an MFMA stage of its
  // own: K = n_xa).  Outputs land in s.o[row][k*out_dim+o].  No global stores (flush at the end of the kernel).
  auto fwd_head = [&](const HtHead& h_, const HmFw<W, FAST>& f, const float (&u0)[NB][4], float (&zsv)[GRL_MAX_LAYERS][NB][4]) {
    struct { int L, H0, n_xa, no, hid[GRL_MAX_LAYERS]; } h = {LY(h_.L), WD(h_.H0), h_.n_xa, h_.n_out * h_.out_dim, {WD(h_.hid[0]), WD(h_.hid[1]), WD(h_.hid[2]), WD(h_.hid[3])}};
    // ---- layer 0
    {
      hm_f4 acc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = hm_f4{u0[b][0], u0[b][1], u0[b][2], u0[b][3]};
      if (h.n_xa > 0) mma_a(acc, s.pi, f.a0);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * q + i;
          const float v = (nb[b] < WD(h.H0) && ROW(r)) ? fmaxf(acc[b][i] + f.b0[b], 0.f) : 0.f;
          zsv[0][b][i] = v;
          s.z[cur][r][nb[b]] = v;
        }
    }
    HM_SYNC();
    // ---- hidden layers
#pragma unroll
    for (int ly = 1; ly < GRL_MAX_LAYERS; ++ly) {
      if (ly < LY(h.L)) {
        hm_f4 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = hm_f4{0.f, 0.f, 0.f, 0.f};
        mma(acc, s.z[cur], f.hid[ly]);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * q + i;
            const float v = (nb[b] < WD(h.hid[ly]) && ROW(r)) ? fmaxf(acc[b][i] + f.bh[ly][b], 0.f) : 0.f;
            zsv[ly][b][i] = v;
            s.z[cur ^ 1][r][nb[b]] = v;
          }
        cur ^= 1;
        HM_SYNC();
      }
    }
    // ---- output layer(s): [mu | log_std] or one value
    stage_few(s.z[cur], f.out, f.bo, [&](int r, int col, float v) {
      if (col < h.no) s.o[r][col] = ROW(r) ? v : 0.f;
    });
  };

  // backward of one head.  rank1: the head has one scalar output whose gradient per row is s.sv[5][row]; otherwise
  // the output gradients are s.o[row][k*out_dim + o] (zero beyond).  gsv[l][b][i]: gradients w.r.t. the layer
  // pre-activations in the result layout (stored at the end); with want_da the gradient w.r.t. the action -> s.da.
  auto bwd_head = [&](const HtHead& h_, const HmBw<W, FAST>& g, const float (&zsv)[GRL_MAX_LAYERS][NB][4], float (&gsv)[GRL_MAX_LAYERS][NB][4],
                      bool rank1, bool want_da) {
    struct { int L, n_xa, hid[GRL_MAX_LAYERS]; } h = {LY(h_.L), h_.n_xa, {WD(h_.hid[0]), WD(h_.hid[1]), WD(h_.hid[2]), WD(h_.hid[3])}};
    const int L = LY(h.L);
    int HL = WD(h.hid[0]);
#pragma unroll
    for (int ly = 1; ly < GRL_MAX_LAYERS; ++ly) HL = (ly == L - 1) ? WD(h.hid[ly]) : HL;
    // ---- output layer(s) -> gradient of the last hidden pre-activation
    {
      hm_f4 acc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = hm_f4{0.f, 0.f, 0.f, 0.f};
      if (rank1) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[b][i] = s.sv[5][4 * q + i] * g.ow[b];
      } else {
        mma_a(acc, s.o, g.out);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * q + i;
          float zl = 0.f;                         // activation of the last hidden layer (static register indices)
#pragma unroll
          for (int ly = 0; ly < GRL_MAX_LAYERS; ++ly) zl = (ly == L - 1) ? zsv[ly][b][i] : zl;
          const float v = (nb[b] < HL && ROW(r) && zl > 0.f) ? acc[b][i] : 0.f;
          s.z[cur][r][nb[b]] = v;
#pragma unroll
          for (int ly = 0; ly < GRL_MAX_LAYERS; ++ly) gsv[ly][b][i] = (ly == L - 1) ? v : gsv[ly][b][i];
        }
    }
    HM_SYNC();
    // ---- hidden layers: g_{l-1} = mask * (g_l . W_l^T)
#pragma unroll
    for (int ly = GRL_MAX_LAYERS - 1; ly >= 1; --ly) {
      if (ly < L) {
        hm_f4 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = hm_f4{0.f, 0.f, 0.f, 0.f};
        mma(acc, s.z[cur], g.hid[ly]);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * q + i;
            const float v = (nb[b] < WD(h.hid[ly - 1]) && ROW(r) && zsv[ly - 1][b][i] > 0.f) ? acc[b][i] : 0.f;
            s.z[cur ^ 1][r][nb[b]] = v;
            gsv[ly - 1][b][i] = v;
          }
        cur ^= 1;
        HM_SYNC();
      }
    }
    // ---- d xa = g_0 . w0a^T
    if (want_da)
      stage_few(s.z[cur], g.da, 0.f, [&](int r, int col, float v) {
        if (col < h.n_xa) s.da[r][col] = v;
      });
  };

  // squashed-Gaussian sample of the rows from s.o = [mu | log_std]: one lane per (row, action dimension), 32 lanes
  // per row (2A <= 64), log-probability / entropy added up over a row's lanes by a fixed butterfly
  auto sample = [&]() {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int r = 8 * pass + (t >> 5), j = t & 31;
      float lp = 0.f, en = 0.f;
      if (j < A) {
        const float m = s.o[r][j], lr = s.o[r][A + j];
        s.pi[r][j] = ht_sample_elem(m, lr, s.eps[r][j], lp, en);
        s.mu[r][j] = m;
        s.ls[r][j] = lr;
      }
#pragma unroll
      for (int msk = 16; msk >= 1; msk >>= 1) { lp += __shfl_xor(lp, msk, 64); en += __shfl_xor(en, msk, 64); }
      if (j == 0) { s.sv[2][r] = lp; s.sv[3][r] = en; }
    }
    HM_SYNC();
  };

  // ---- end-of-kernel stores of a head's activations / gradients (result layout -> row-major tensors)
  auto store_z = [&](const HtHead& h_, const float (&zsv)[GRL_MAX_LAYERS][NB][4]) {
    const HtHead h = h_;
#pragma unroll
    for (int ly = 0; ly < GRL_MAX_LAYERS; ++ly) {
      float* zp = ly == 0 ? h.z0 : h.z[ly];
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (ly < LY(h.L) && zp && nb[b] < WD(h.hid[ly])) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (ROW(4 * q + i)) HM_GW(zp)[(long)(row0 + 4 * q + i) * WD(h.hid[ly]) + nb[b]] = zsv[ly][b][i];
        }
    }
  };
  auto store_g = [&](const HtHead& h_, const float (&gsv)[GRL_MAX_LAYERS][NB][4]) {
    const HtHead h = h_;
#pragma unroll
    for (int ly = 0; ly < GRL_MAX_LAYERS; ++ly) {
      float* gp = ly == 0 ? h.g0 : h.g[ly];
      const int ldg = ly == 0 ? h.ldg0 : WD(h.hid[ly]);
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (ly < LY(h.L) && gp && nb[b] < WD(h.hid[ly])) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (ROW(4 * q + i)) HM_GW(gp)[(long)(row0 + 4 * q + i) * ldg + nb[b]] = gsv[ly][b][i];
        }
    }
  };
  auto store_rows = [&](float* dst, int ld, const float* src_row0) {   // one scalar per row, lanes 0..15
    if (t < HT_RB && ROW(t)) HM_GW(dst)[(long)(row0 + t) * ld] = src_row0[t];
  };

  // ---------------------------------------------------------------- round trip 2, one burst
  // layer-0 partial sums of a head: the 16 x W tile of each split as 16-byte loads (quad e = t + 256 j of the tile:
  // row e / (W/4), columns 4 (e % (W/4)) ..), added in split order
  constexpr int QPR = W / 4;
  auto issue_u = [&](const HtHead& h_, hm_f4 (&v)[4][NB]) {
    struct { const float* u; int ldu, u_split, H0; long u_stride; } h = {h_.u, h_.ldu, h_.u_split, WD(h_.H0), h_.u_stride};
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int e = t + 256 * j, r = e / QPR, c4 = 4 * (e % QPR);
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        v[sp][j] = hm_f4{0.f, 0.f, 0.f, 0.f};
        if (sp < h.u_split && c4 < WD(h.H0) && ROW(r))
          v[sp][j] = *(const GRL_GLOBAL hm_f4*)(HM_G(h.u) + sp * h.u_stride + (long)(row0 + r) * h.ldu + c4);
      }
    }
  };
  auto land_u = [&](const HtHead& h, const hm_f4 (&v)[4][NB], int slot) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int e = t + 256 * j, r = e / QPR, c4 = 4 * (e % QPR);
      hm_f4 acc = v[0][j];
#pragma unroll
      for (int sp = 1; sp < 4; ++sp) acc += v[sp][j];
      for (int sp = 4; sp < h.u_split; ++sp)
        if (c4 < WD(h.H0) && ROW(r)) acc += *(const GRL_GLOBAL hm_f4*)(HM_G(h.u) + sp * h.u_stride + (long)(row0 + r) * h.ldu + c4);
      *(hm_f4*)&s.u[slot][r][c4] = acc;
    }
  };
  auto get_u = [&](int slot, float (&u0)[NB][4]) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) u0[b][i] = s.u[slot][4 * q + i][nb[b]];
  };
  float uA[NB][4], uB[NB][4], uC[NB][4], uD[NB][4];
  HmFw<W, FAST> fA, fB;
  HmBw<W, FAST> gA;
  const HtHead& hq = a.h[type <= 1 ? 5 : type];             // second head of the chain; its action input comes from s.pi
  const HtHead& h0 = a.h[type <= 1 ? 0 : 4];
  {
    // ---- issue ...
    hm_f4 vA[4][NB], vB[4][NB], vC[4][NB], vD[4][NB];
    issue_u(h0, vA);
    issue_u(hq, vB);
    if (type == 1) { issue_u(a.h[6], vC); issue_u(a.h[1], vD); }
    const float la = HM_G(a.log_ent_coef)[0];
    float in0[2], rw = 0.f, dn = 0.f;                        // per-row inputs: eps (types 0, 1) or the minibatch action
    const int nin = type <= 1 ? A : hq.n_xa;                 // <= 32: lane -> (row t / 32 + 8 e, dimension t % 32)
    const float* inp = type <= 1 ? a.eps : hq.xa;
    const int ldin = type <= 1 ? A : hq.ld_xa;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = (t >> 5) + 8 * e, j = t & 31;
      in0[e] = (j < nin && ROW(r)) ? HM_G(inp)[(long)(row0 + r) * ldin + j] : 0.f;
    }
    if (type >= 2 && t < HT_RB && ROW(t)) { rw = HM_G(a.rew)[row0 + t]; dn = HM_G(a.done)[row0 + t]; }
    prefetch_fwd(h0, fA);
    stamp();
    // ---- ... then consume
    if (t == 0) s.alpha = expf(la);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = (t >> 5) + 8 * e, j = t & 31;
      if (j < nin) { if (type <= 1) s.eps[r][j] = in0[e]; else s.pi[r][j] = in0[e]; }
    }
    if (type >= 2 && t < HT_RB) { s.sv[6][t] = rw; s.sv[7][t] = dn; }
    land_u(h0, vA, 0);
    land_u(hq, vB, 1);
    if (type == 1) { land_u(a.h[6], vC, 2); land_u(a.h[1], vD, 3); }
  }
  stamp();
  HM_SYNC();
  stamp();
  get_u(0, uA); get_u(1, uB);
  if (type == 1) { get_u(2, uC); get_u(3, uD); }
  prefetch_fwd(hq, fB);                       // lands while the first head runs

  // ---------------------------------------------------------------- the chains
  float zsA[GRL_MAX_LAYERS][NB][4], zsB[GRL_MAX_LAYERS][NB][4], gs[GRL_MAX_LAYERS][NB][4];
#pragma unroll
  for (int ly = 0; ly < GRL_MAX_LAYERS; ++ly)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) { zsA[ly][b][i] = 0.f; zsB[ly][b][i] = 0.f; gs[ly][b][i] = 0.f; }
  if (type == 0) {
    fwd_head(a.h[0], fA, uA, zsA); stamp();     // pi: s.o = [mu | log_std]
    prefetch_bwd(a.h[5], gA, true, true);       // (lands while the sample and qf1(s, pi) run)
    sample(); stamp();
    fwd_head(a.h[5], fB, uB, zsB); stamp();     // qf1(s, pi): s.o[r][0]
    if (t < HT_RB) { s.sv[0][t] = s.o[t][0]; s.sv[5][t] = ROW(t) ? -invB : 0.f; }
    HM_SYNC();
    // (gradients of qf1's own weights are not wanted here -- policy loss: they land in `gs`, which the pi backward overwrites)
    bwd_head(a.h[5], gA, zsB, gs, true, true); stamp();
    prefetch_bwd(a.h[0], gA, false, false);
    // sample backward; s.o becomes the A operand [dmu | dls] of the pi backward (zero beyond 2A)
    {
      const float alpha_over_b = s.alpha * invB;
      for (int e = t; e < HT_RB * HM_AW; e += 256) {
        const int r = e / HM_AW, k = e - r * HM_AW;
        float v = 0.f;
        if (k < 2 * A && ROW(r)) {
          const int j = k < A ? k : k - A;
          float m, d;
          ht_sample_bwd_elem(s.ls[r][j], s.eps[r][j], s.pi[r][j], s.da[r][j], alpha_over_b, m, d);
          v = k < A ? m : d;
        }
        s.o[r][k] = v;
      }
      HM_SYNC();
    }
    bwd_head(a.h[0], gA, zsA, gs, false, false); stamp();
    // ---- flush
    store_z(a.h[0], zsA);
    store_g(a.h[0], gs);
    for (int e = t; e < HT_RB * A; e += 256) {
      const int r = e / A, j = e - r * A, row = row0 + r;
      if (row < B) {
        HM_GW(a.h[0].out[0])[(long)row * A + j] = s.mu[r][j];
        HM_GW(a.h[0].out[1])[(long)row * A + j] = s.ls[r][j];
        HM_GW(a.pi_a)[(long)row * A + j] = s.pi[r][j];
        HM_GW(a.da_pi)[(long)row * A + j] = s.da[r][j];
        HM_GW(a.dmu)[(long)row * a.ld_dm + j] = s.o[r][j];
        HM_GW(a.dls)[(long)row * a.ld_dm + j] = s.o[r][A + j];
      }
    }
    store_rows(a.logp, 1, s.sv[2]);
    store_rows(a.ent, 1, s.sv[3]);
    store_rows(a.h[5].out[0], 1, s.sv[0]);
    store_rows(a.d_out[4], a.ld_d, s.sv[5]);
  } else if (type == 1) {
    fwd_head(a.h[0], fA, uA, zsA); stamp();     // pi (not stored: type 0 owns its tensors)
    prefetch_fwd(a.h[6], fA);
    sample(); stamp();
    fwd_head(a.h[5], fB, uB, zsB); stamp();     // qf1(s, pi)
    prefetch_fwd(a.h[1], fB);
    if (t < HT_RB) s.sv[0][t] = s.o[t][0];
    HM_SYNC();
    fwd_head(a.h[6], fA, uC, zsB); stamp();     // qf2(s, pi)
    prefetch_bwd(a.h[1], gA, true, false);
    if (t < HT_RB) s.sv[1][t] = s.o[t][0];
    HM_SYNC();
    fwd_head(a.h[1], fB, uD, zsA); stamp();     // vf
    if (t < HT_RB) {
      const float vb = fminf(s.sv[0][t], s.sv[1][t]) - s.alpha * s.sv[2][t];
      s.sv[8][t] = s.o[t][0];
      s.sv[5][t] = ROW(t) ? (s.o[t][0] - vb) * invB : 0.f;
    }
    HM_SYNC();
    bwd_head(a.h[1], gA, zsA, gs, true, false); stamp();
    store_z(a.h[1], zsA);
    store_g(a.h[1], gs);
    store_rows(a.h[6].out[0], 1, s.sv[1]);
    store_rows(a.h[1].out[0], 1, s.sv[8]);
    store_rows(a.d_out[1], a.ld_d, s.sv[5]);
  } else {
    const HtHead& h = a.h[type];
    fwd_head(a.h[4], fA, uA, zsA); stamp();     // target vf of next_obs
    prefetch_bwd(h, gA, true, false);
    if (t < HT_RB) s.sv[4][t] = s.o[t][0];
    HM_SYNC();
    fwd_head(h, fB, uB, zsB); stamp();          // qf(s, a) on the minibatch actions (s.pi)
    if (t < HT_RB) {
      const float qb = s.sv[6][t] + (1.f - s.sv[7][t]) * a.gamma * s.sv[4][t];
      s.sv[8][t] = s.o[t][0];
      s.sv[5][t] = ROW(t) ? (s.o[t][0] - qb) * invB : 0.f;
    }
    HM_SYNC();
    bwd_head(h, gA, zsB, gs, true, false); stamp();
    store_z(h, zsB);
    store_g(h, gs);
    if (type == 2) store_rows(a.h[4].out[0], 1, s.sv[4]);
    store_rows(h.out[0], 1, s.sv[8]);
    store_rows(a.d_out[type], a.ld_d, s.sv[5]);
  }
  stamp();
}

// gp / gx / n_ride: the image gather of the NEXT update riding on this launch (plan_sac "gather_ride"; gather_images_rider):
// n_ride workgroups behind the 4 rows of head workgroups, linearised over blockIdx.y >= 4.  The head chains leave three CUs in
// four idle for 16 us; the images land in the buffer this update does not read.
template <int W, bool FAST>
__global__ __launch_bounds__(256) void heads_fused_kernel(const HeadsFusedArgs* __restrict__ ap, const GatherArgs g, const int gx,
                                                         const int n_ride) {
  if (blockIdx.y >= 4) {
    const int r = ((int)blockIdx.y - 4) * (int)gridDim.x + (int)blockIdx.x;
    if (r < n_ride) gather_images_rider(g, gx, r);
    return;
  }
  __shared__ __attribute__((aligned(16))) HmLds<W> s;
  if constexpr (W == 128) {      // (the same split of the 64-wide fast kernel measured neutral: 5 780 / 5 758 against 5 776 / 5 759)
    if (blockIdx.y == 0) heads_fused_body<W, FAST, 0>(ap, s);
    else if (blockIdx.y == 1) heads_fused_body<W, FAST, 1>(ap, s);
    else heads_fused_body<W, FAST, 2>(ap, s);
  } else {
    heads_fused_body<W, FAST, -1>(ap, s);
  }
}

#endif  // GRL_HOSTEMU
#endif  // GRL_HEADS_TYPES_ONLY

}  // namespace grl
