// igemm.h -- table-driven implicit GEMM on the gfx950 f32 matrix cores.
//
// One kernel covers every GEMM-shaped piece of the update (conv forward, conv backward-data by
// stride-parity classes, conv/dense weight gradients, dense layers of the CNN head and the
// actor/critic MLPs):
//
//     C[i, j] = epilogue( sum_r P(i, r) * Q(r, j) )
//
// P and Q are *addressing functions* over tensors that stay in their natural HBM layout (NHWC
// activations, TF HWIO / [in,out] weights): offset = rowterm(i) + colterm(r), each term either an
// affine expression or a small int32 table built once on the host (static shapes).  This is what
// lets a strided VALID convolution, its transposed gather and its weight gradient share one
// MFMA main loop without im2col buffers.
//
// Tile: 64x64 outputs per 256-thread workgroup (4 wave64, one 32x32 accumulator each) on
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  K is consumed in slabs of 32 staged
// through LDS as Ps[r][i], Qs[r][j] with a +1 pad so that both the staging writes and the
// per-MFMA fragment reads (lane l: row l&31, k-slice l>>5) are bank-conflict free.
#pragma once
#ifdef GRL_HOSTEMU
#include "hostemu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace grl {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2 };

struct IgemmProb {
  int32_t M, N, K;        // output rows, output cols, reduction length
  int32_t split;          // number of reduction splits (>= 1); split s writes slab s
  int32_t k_chunk;        // reduction elements per split (multiple of 32)
  // ---- P operand: P(i, r) = p_base[part(r)][ rowterm(i) + colterm(r) ]
  const float* p_base[3];
  int32_t p_ld_i[3];      // rowterm = i * p_ld_i[part]           (when p_tab_i == nullptr)
  int32_t p_ld_r[3];      // colterm = (r - start[part]) * p_ld_r (when p_tab_r == nullptr)
  int32_t p_k0, p_k1;     // part boundaries along r: [0,k0) [k0,k1) [k1,K)
  const int32_t* p_tab_i; // optional offset tables
  const int32_t* p_tab_r;
  const uint64_t* p_vmask_i;  // optional validity: bit p_tap_r[r] of p_vmask_i[i] (padded convs,
  const uint8_t* p_tap_r;     //   transposed-conv gathers); invalid taps read as 0
  int32_t p_ones_i;       // row whose P value is the constant 1 (bias-gradient row), -1 = none
  // ---- Q operand: Q(r, j) = q_base[part(r)][ rowterm(r) + j * q_ld_j[part] ]
  const float* q_base[3];
  int32_t q_ld_r[3];
  int32_t q_ld_j[3];
  const int32_t* q_tab_r;
  // ---- epilogue
  float* c;
  int32_t ldc;
  const int32_t* c_tab_i; // optional output row offsets; negative = row not stored
  int64_t slab_stride;    // floats between split slabs
  const float* bias;      // + bias[j]
  const float* relu_mask; // value kept only where relu_mask[same offset as c] > 0
  int32_t act;
  float act_alpha;
  int32_t accumulate;     // += existing c
};

#ifdef GRL_HOSTEMU
// TEST-ONLY reference of the descriptor semantics (see hostemu.h); one call computes one tile.
template <bool P_CONTIG_R, bool Q_CONTIG_J>
void igemm_kernel(const IgemmProb* probs, const int4* tiles) {
  if (threadIdx.x != 0) return;
  const int4 tl = tiles[blockIdx.x];
  const IgemmProb& pb = probs[tl.x];
  const int r_begin = tl.y * pb.k_chunk, r_end = std::min(pb.K, r_begin + pb.k_chunk);
  float* cbase = pb.c + (long)tl.y * pb.slab_stride;
  for (int i = tl.z * 64; i < std::min(pb.M, tl.z * 64 + 64); ++i)
    for (int j = tl.w * 64; j < std::min(pb.N, tl.w * 64 + 64); ++j) {
      float acc = 0.f;
      for (int r = r_begin; r < r_end; ++r) {
        const int part = (r >= pb.p_k0) + (r >= pb.p_k1);
        const int rstart = part == 0 ? 0 : (part == 1 ? pb.p_k0 : pb.p_k1);
        float pv = 0.f;
        bool ok = true;
        if (pb.p_vmask_i) ok = (pb.p_vmask_i[i] >> pb.p_tap_r[r]) & 1ull;
        if (i == pb.p_ones_i) pv = 1.f;
        else if (ok) {
          const long rowterm = pb.p_tab_i ? pb.p_tab_i[i] : (long)i * pb.p_ld_i[part];
          const long colterm = pb.p_tab_r ? pb.p_tab_r[r] : (long)(r - rstart) * pb.p_ld_r[part];
          pv = pb.p_base[part][rowterm + colterm];
        }
        const long qrow = pb.q_tab_r ? pb.q_tab_r[r] : (long)(r - rstart) * pb.q_ld_r[part];
        const float qv = pb.q_base[part][qrow + (long)j * pb.q_ld_j[part]];
        acc = fmaf(pv, qv, acc);
      }
      long off;
      if (pb.c_tab_i) {
        if (pb.c_tab_i[i] < 0) continue;
        off = (long)pb.c_tab_i[i] + j;
      } else off = (long)i * pb.ldc + j;
      float v = acc + (pb.bias ? pb.bias[j] : 0.f);
      if (pb.accumulate) v += cbase[off];
      if (pb.act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (pb.act == ACT_LEAKY) v = v > 0.f ? v : pb.act_alpha * v;
      if (pb.relu_mask) v = pb.relu_mask[off] > 0.f ? v : 0.f;
      cbase[off] = v;
    }
}
#else
template <bool P_CONTIG_R, bool Q_CONTIG_J>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmProb* __restrict__ probs,
                                                   const int4* __restrict__ tiles) {
  constexpr int BI = 64, BJ = 64, BR = 32, LDP = BI + 1, LDQ = BJ + 1;
  __shared__ float lds[BR * LDP + BR * LDQ];
  float* Ps = lds;
  float* Qs = lds + BR * LDP;

  // flat tile list: {problem, split, i-tile, j-tile}; heavy problems first (built on the host)
  const int4 tl = tiles[blockIdx.x];
  const int2 zm = make_int2(tl.x, tl.y);
  const IgemmProb* __restrict__ pb = probs + zm.x;
  const int M = pb->M, N = pb->N, K = pb->K;
  const int i0 = tl.z * BI, j0 = tl.w * BJ;

  const int r_begin = zm.y * pb->k_chunk;
  const int r_end = min(K, r_begin + pb->k_chunk);

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wi = wave >> 1, wj = wave & 1;

  // ---- uniform operand descriptors
  const float* pB0 = pb->p_base[0];
  const float* pB1 = pb->p_base[1];
  const float* pB2 = pb->p_base[2];
  const int pLi0 = pb->p_ld_i[0], pLi1 = pb->p_ld_i[1], pLi2 = pb->p_ld_i[2];
  const int pLr0 = pb->p_ld_r[0], pLr1 = pb->p_ld_r[1], pLr2 = pb->p_ld_r[2];
  const int pk0 = pb->p_k0, pk1 = pb->p_k1;
  const int32_t* pTi = pb->p_tab_i;
  const int32_t* pTr = pb->p_tab_r;
  const uint64_t* pVm = pb->p_vmask_i;
  const uint8_t* pTap = pb->p_tap_r;
  const int ones_i = pb->p_ones_i;
  const float* qB0 = pb->q_base[0];
  const float* qB1 = pb->q_base[1];
  const float* qB2 = pb->q_base[2];
  const int qLr0 = pb->q_ld_r[0], qLr1 = pb->q_ld_r[1], qLr2 = pb->q_ld_r[2];
  const int qLj0 = pb->q_ld_j[0], qLj1 = pb->q_ld_j[1], qLj2 = pb->q_ld_j[2];
  const int32_t* qTr = pb->q_tab_r;

  // ---- per-thread staging coordinates (8 P elements + 8 Q elements per 32-deep slab)
  // P: lanes run along r (P_CONTIG_R) or along i.
  const int p_rl = P_CONTIG_R ? (t & 31) : (t >> 6);   // + 4e when !P_CONTIG_R
  const int p_il = P_CONTIG_R ? (t >> 5) : (t & 63);   // + 8e when  P_CONTIG_R
  // Q: lanes run along j (Q_CONTIG_J) or along r.
  const int q_rl = Q_CONTIG_J ? (t >> 6) : (t & 31);
  const int q_jl = Q_CONTIG_J ? (t & 63) : (t >> 5);

  // rows of P this thread stages are the same for every slab: hoist their row terms
  int p_rowtab[8];
  uint64_t p_vm[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int i = i0 + p_il + (P_CONTIG_R ? 8 * e : 0);
    p_rowtab[e] = (pTi && i < M) ? pTi[i] : 0;
    p_vm[e] = (pVm && i < M) ? pVm[i] : ~0ull;
    if (!P_CONTIG_R) break;   // one row per thread
  }

  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 acc;
#pragma unroll
  for (int x = 0; x < 16; ++x) acc[x] = 0.f;

  for (int r0 = r_begin; r0 < r_end; r0 += BR) {
    float pv[8], qv[8];
    // ---------------- stage P
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int rl = p_rl + (P_CONTIG_R ? 0 : 4 * e);
      const int il = p_il + (P_CONTIG_R ? 8 * e : 0);
      const int r = r0 + rl, i = i0 + il;
      const int part = (r >= pk0) + (r >= pk1);
      const float* base = part == 0 ? pB0 : (part == 1 ? pB1 : pB2);
      const int ldi = part == 0 ? pLi0 : (part == 1 ? pLi1 : pLi2);
      const int ldr = part == 0 ? pLr0 : (part == 1 ? pLr1 : pLr2);
      const int rstart = part == 0 ? 0 : (part == 1 ? pk0 : pk1);
      bool ok = (r < r_end) && (i < M);
      const int rc = ok ? r : r_begin;   // keep table reads in range
      const int colterm = pTr ? pTr[rc] : (rc - rstart) * ldr;
      const int rowterm = pTi ? p_rowtab[P_CONTIG_R ? e : 0] : i * ldi;
      if (pVm) ok = ok && ((p_vm[P_CONTIG_R ? e : 0] >> pTap[rc]) & 1ull);
      float v = 0.f;
      if (ok && i != ones_i) v = base[(long)rowterm + colterm];
      if (i == ones_i && r < r_end) v = 1.f;
      pv[e] = v;
    }
    // ---------------- stage Q
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int rl = q_rl + (Q_CONTIG_J ? 4 * e : 0);
      const int jl = q_jl + (Q_CONTIG_J ? 0 : 8 * e);
      const int r = r0 + rl, j = j0 + jl;
      const int part = (r >= pk0) + (r >= pk1);
      const float* base = part == 0 ? qB0 : (part == 1 ? qB1 : qB2);
      const int ldr = part == 0 ? qLr0 : (part == 1 ? qLr1 : qLr2);
      const int ldj = part == 0 ? qLj0 : (part == 1 ? qLj1 : qLj2);
      const int rstart = part == 0 ? 0 : (part == 1 ? pk0 : pk1);
      const bool ok = (r < r_end) && (j < N);
      const int rc = ok ? r : r_begin;
      const int rowterm = qTr ? qTr[rc] : (rc - rstart) * ldr;
      float v = 0.f;
      if (ok) v = base[(long)rowterm + (long)j * ldj];
      qv[e] = v;
    }
    __syncthreads();   // previous slab fully consumed
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int prl = p_rl + (P_CONTIG_R ? 0 : 4 * e);
      const int pil = p_il + (P_CONTIG_R ? 8 * e : 0);
      Ps[prl * LDP + pil] = pv[e];
      const int qrl = q_rl + (Q_CONTIG_J ? 4 * e : 0);
      const int qjl = q_jl + (Q_CONTIG_J ? 0 : 8 * e);
      Qs[qrl * LDQ + qjl] = qv[e];
    }
    __syncthreads();
    // ---------------- 16 x v_mfma_f32_32x32x2_f32: lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]
    const float* pa = Ps + (lane >> 5) * LDP + wi * 32 + (lane & 31);
    const float* qa = Qs + (lane >> 5) * LDQ + wj * 32 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < BR / 2; ++kk) {
      const float a = pa[2 * kk * LDP];
      const float b = qa[2 * kk * LDQ];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }

  // ---------------- epilogue: D[row][col], col = lane&31, row = (x&3) + 8*(x>>2) + 4*(lane>>5)
  float* cbase = pb->c + (long)zm.y * pb->slab_stride;
  const int ldc = pb->ldc;
  const int32_t* cT = pb->c_tab_i;
  const float* bias = pb->bias;
  const float* rmask = pb->relu_mask;
  const int act = pb->act;
  const float alpha = pb->act_alpha;
  const int accumulate = pb->accumulate;
  const int j = j0 + wj * 32 + (lane & 31);
  if (j < N) {
    const float bj = bias ? bias[j] : 0.f;
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const int i = i0 + wi * 32 + (x & 3) + 8 * (x >> 2) + 4 * (lane >> 5);
      if (i >= M) continue;
      long off;
      if (cT) {
        const int o = cT[i];
        if (o < 0) continue;
        off = (long)o + j;
      } else {
        off = (long)i * ldc + j;
      }
      float v = acc[x] + bj;
      if (accumulate) v += cbase[off];
      if (act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (act == ACT_LEAKY) v = v > 0.f ? v : alpha * v;
      if (rmask) v = rmask[off] > 0.f ? v : 0.f;
      cbase[off] = v;
    }
  }
}

#endif  // GRL_HOSTEMU

}  // namespace grl
