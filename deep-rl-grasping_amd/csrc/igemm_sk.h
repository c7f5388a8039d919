// igemm_sk.h -- streaming implicit GEMM for SHORT reductions (K = 32 or 64) and narrow outputs (N <= 32):
// the first convolution of the extractors (custom_obs_policy.py:34, 8x8 stride 4 over one depth channel:
// M = B*225 output pixels, K = 64, N = 32).
//
// With K = 64 a 128-row tile is two slabs of MFMAs (0.85 us); in igemm2_kernel one workgroup per tile then
// spends its life in the dependent chain tile -> descriptor -> tables -> data -> LDS -> MFMA -> store
// (1350 workgroups, ~10 us each, three per CU).  Here a workgroup keeps the kernel matrix Q (K x 32, 8 KB)
// resident in LDS and STREAMS over consecutive row tiles of one problem: the gathered operand of tile t+1
// is loaded into registers (and the row offsets of tile t+2 fetched) while tile t is multiplied and stored,
// so the chain is paid once per workgroup instead of once per tile, and Q is loaded once instead of per tile.
//
// Arithmetic is igemm2_kernel's, element for element: same LDS k-permutation (MFMA step m of lane half h
// takes k = 2m + h), same slab order, same epilogue expression -- results are bit-identical, which keeps one
// summation order across the three GEMM kernels.
#pragma once
#include "igemm2.h"

namespace grl {

#ifndef GRL_GEMM_TYPES_ONLY
#ifdef GRL_HOSTEMU
#include "igemm_sk_ref1.h"   // tests/hostemu: the emulation build only
#else

template <int K>
__global__ __launch_bounds__(256) void igemm_sk_kernel(const IgemmProb* __restrict__ probs, const int4* __restrict__ work) {
  constexpr int BM = 128, BN = 32, NSLAB = K / 32;
  constexpr int LDPK = K + 4, LDQN = BN + 4, PSZ = BM * LDPK, QSZ = K * LDQN;
  constexpr int PNQ = K / 4, RP = 256 / PNQ, NVP = BM / RP;      // k-quads per row, rows per pass, vectors per thread
  constexpr int LDC_S = BN + 4;
  static_assert(K == 32 || K == 64, "short-K kernel: one or two slabs");
  static_assert(32 * LDC_S <= 32 * LDPK, "a wave stages its output inside its own operand rows");
  __shared__ __attribute__((aligned(16))) float lds[2 * PSZ + QSZ];
  float* Qs = lds + 2 * PSZ;

  const int4 wk = work[blockIdx.x];
  const IgemmProb* __restrict__ pb = probs + blockIdx.x;   // per-workgroup descriptor copy (add_launch)
  const int M = pb->M, N = pb->N;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  typedef const GRL_GLOBAL int32_t* gci32;
  typedef const GRL_GLOBAL float* gcf32;
  typedef GRL_GLOBAL float* gf32;
  const __amdgpu_buffer_rsrc_t rsP = i2_rsrc(pb->p_base[0]);
  const __amdgpu_buffer_rsrc_t rsQ = i2_rsrc(pb->q_base[0]);
  const gci32 pTi = (gci32)pb->p_tab_i;
  const gci32 pTr = (gci32)pb->p_tab_r;
  auto kpos = [](int k) { return (k & ~31) | ((k & 1) << 4) | ((k & 31) >> 1); };

  // ---- row offsets of the first two tiles, the (tile-independent) column offset, the kernel matrix
  const int p_q = t % PNQ, p_l = t / PNQ;
  int fixA[NVP], fixB[NVP];   // row offsets of alternate tiles (two named sets: no loop-carried copies)
  // raw table entries only: nothing touches a loaded value before load_tile consumes it one tile later
  // (a select right here would make the compiler wait for the load where it is issued)
  auto fetch_fix = [&](int tile, int (&f)[NVP]) {
#pragma unroll
    for (int e = 0; e < NVP; ++e) {
      const int i = tile * BM + p_l + e * RP;
      f[e] = pTi[(tile < wk.y + wk.z && i < M) ? i : 0];
    }
  };
  fetch_fix(wk.y, fixA);
  const int colterm = pTr[4 * p_q];
  {
    const int qLr = pb->q_ld_r[0];
    constexpr int NQQ = K * (BN / 4) / 256;      // quads of Q per thread
    const int q_q = t % (BN / 4), q_l = t / (BN / 4);
    f32x4 qv[NQQ];
#pragma unroll
    for (int e = 0; e < NQQ; ++e) {
      const int k = q_l + e * (256 / (BN / 4));
      qv[e] = i2_ld(rsQ, 4 * q_q < N ? (k * qLr + 4 * q_q) * 4 : I2_OOB);
    }
#pragma unroll
    for (int e = 0; e < NQQ; ++e) {
      const int k = q_l + e * (256 / (BN / 4));
      *(f32x4*)(Qs + kpos(k) * LDQN + 4 * q_q) = qv[e];
    }
  }
  f32x4 pv[NVP];
  auto load_tile = [&](const int (&f)[NVP], int tile) {
#pragma unroll
    for (int e = 0; e < NVP; ++e) {
      const int i = tile * BM + p_l + e * RP;
      pv[e] = i2_ld(rsP, (tile < wk.y + wk.z && i < M) ? (f[e] + colterm) * 4 : I2_OOB);
    }
  };
  load_tile(fixA, wk.y);
  __builtin_amdgcn_sched_barrier(0);
  fetch_fix(wk.y + 1, fixB);   // issued after the data loads, as in the loop: one static wait count serves both paths
  __builtin_amdgcn_sched_barrier(0);

  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto put_kquad = [&](float* rowp, int q, f32x4 v) {   // as igemm2_kernel: logical k = 4q..4q+3 of a K-contiguous row
    asm volatile("" : "+v"(v));
    float* d = rowp + (q >> 3) * 32 + 2 * (q & 7);
    *(f32x2*)d = f32x2{v.x, v.z};
    *(f32x2*)(d + 16) = f32x2{v.y, v.w};
  };

  // epilogue constants
  const __amdgpu_buffer_rsrc_t rsC = i2_rsrc(pb->c);   // masked lanes store to an out-of-range offset (dropped): no branches
  const int ldc = pb->ldc;
  const gcf32 bias = (gcf32)pb->bias;
  const int act = pb->act;
  const float alpha = pb->act_alpha;
  const float oscale = pb->out_scale != 0.f ? pb->out_scale : 1.f;
  const int c4 = lane & 7, rq = lane >> 3;         // wide store: lane -> (column quad, row group) inside the wave's 32 rows
  f32x4 bj = {0.f, 0.f, 0.f, 0.f};
  if (bias && 4 * c4 < N) bj = *(const GRL_GLOBAL f32x4*)(bias + 4 * c4);

  auto body = [&](int tt, const int (&fcur)[NVP], int (&fnext)[NVP]) {
    const int tile = wk.y + tt;
    float* buf = lds + (tt & 1) * PSZ;
    // ---- gathered operand of this tile: registers -> LDS
#pragma unroll
    for (int e = 0; e < NVP; ++e) put_kquad(buf + (p_l + e * RP) * LDPK, p_q, pv[e]);
    __syncthreads();
    // ---- next tile's data and the row offsets of the one after travel while this tile is multiplied
    load_tile(fcur, tile + 1);
    __builtin_amdgcn_sched_barrier(0);
    fetch_fix(tile + 2, fnext);
    __builtin_amdgcn_sched_barrier(0);
    // ---- MFMAs: wave w owns rows 32w .. 32w+31, all 32 columns
    f32x16 acc;
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[x] = 0.f;
    const float* Pw = buf + (wave * 32 + li) * LDPK + 16 * lh;
#pragma unroll
    for (int sb = 0; sb < NSLAB; ++sb) {
      float av[16], bv[16];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 v = *(const f32x4*)(Pw + sb * 32 + 4 * c);
        av[4 * c] = v.x; av[4 * c + 1] = v.y; av[4 * c + 2] = v.z; av[4 * c + 3] = v.w;
      }
#pragma unroll
      for (int m = 0; m < 16; ++m) bv[m] = Qs[(sb * 32 + 16 * lh + m) * LDQN + li];
#pragma unroll
      for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[m], acc, 0, 0, 0);
    }
    // ---- epilogue: the wave's 32x32 block goes through its own operand rows in LDS and leaves as 16-byte stores
    float* stg = buf + wave * 32 * LDPK;
#pragma unroll
    for (int x = 0; x < 16; ++x) stg[((x & 3) + 8 * (x >> 2) + 4 * lh) * LDC_S + li] = acc[x];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rl = rq + 8 * e;
      const int i = tile * BM + wave * 32 + rl;
      f32x4 v = *(const f32x4*)(stg + rl * LDC_S + 4 * c4);
      v = v * oscale + bj;
      if (act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      else if (act == ACT_LEAKY) {
        v.x = v.x > 0.f ? v.x : alpha * v.x; v.y = v.y > 0.f ? v.y : alpha * v.y;
        v.z = v.z > 0.f ? v.z : alpha * v.z; v.w = v.w > 0.f ? v.w : alpha * v.w;
      }
      const bool ok = i < M && 4 * c4 < N;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), rsC,
                                             ok ? (int)(((long)i * ldc + 4 * c4) * 4) : I2_OOB, 0, 0);
    }
  };
  // two named offset sets, alternating: a set is consumed by one body and refilled by the next
  int tt = 0;
  for (; tt + 1 < wk.z; tt += 2) {
    body(tt, fixB, fixA);
    body(tt + 1, fixA, fixB);
  }
  if (tt < wk.z) body(tt, fixB, fixA);
}
#endif
#endif  // GRL_GEMM_TYPES_ONLY

}  // namespace grl
