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
  int32_t p_mask_swap;    // 1: validity is bit p_tap_r[i] of p_vmask_i[r] (weight gradients of padded convs)
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
  const float* relu_mask; // value kept where relu_mask[same offset as c] > 0, scaled by act_alpha elsewhere (0: ReLU)
  const int32_t* m_tab_i; // optional row offsets of relu_mask (mask element of (i, j) = relu_mask[m_tab_i[i] + j]): the mask lives
                          // in another layout than c (auto-encoder: activations kept only in their zero-bordered form)
  int32_t act;
  float act_alpha;
  int32_t accumulate;     // += existing c
  float out_scale;        // accumulator scale before bias / accumulate (0 means 1)
  // ---- planning hints: addressing tables checked for 16-byte runs -- bit 0: P tables, bit 1: Q table
  // (igemm2.h eligibility, host only); bit 2: outputs (and ReLU mask / accumulate operands) can be moved
  // 16 bytes at a time: base, row offsets and N are multiples of 4 floats (igemm2 epilogue)
  uint32_t vflags;
  void* dbg_t;            // I2_TIMING builds (scripts/igemm_bench.hip): s_memtime stamps of three workgroups
  float alg_frac;         // host only (profiling): share of the M*N*K MACs that are algorithmic -- the masked parity form of
                          // the conv backward-data multiplies zeros for taps outside the output; 0 means 1
};
enum { VF_P_TABS = 1u, VF_Q_TAB = 2u, VF_C_VEC = 4u, VF_CT4 = 8u };   // VF_CT4 (host): scatter-table offsets are multiples of 4   // VF_C_VEC is also read by the igemm2 epilogue

// Addressing modes are compile-time so the staging code has no branch around any load: every
// load of a slab is issued back to back (masked lanes read offset 0 and select 0 afterwards) and the
// compiler needs a single wait per slab.
#ifndef GRL_HOSTEMU
// Pointers fetched from a descriptor in memory are generic ("flat") to the compiler; flat loads also
// count on lgkmcnt, so every LDS wait in the MFMA loop would drain the prefetched slab.  Typing them
// as global (address space 1) yields global_load / global_store.
#define GRL_GLOBAL __attribute__((address_space(1)))
typedef const GRL_GLOBAL float* gcf32;
typedef GRL_GLOBAL float* gf32;
typedef const GRL_GLOBAL int32_t* gci32;
typedef const GRL_GLOBAL uint64_t* gcu64;
typedef const GRL_GLOBAL uint8_t* gcu8;
#endif
enum { PM_AFFINE = 0, PM_TABLE = 1, PM_TABLE_MASK = 2 };
enum { QM_AFFINE = 0, QM_TABLE = 1 };

#ifndef GRL_GEMM_TYPES_ONLY     // (engine.hip needs the descriptors only: the kernels live in gemm_*.hip)
#ifdef GRL_HOSTEMU
#include "igemm_ref1.h"   // tests/hostemu: the emulation build only
#else
template <int PM, int QM, bool P_CONTIG_R, bool Q_CONTIG_J, int NP>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmProb* __restrict__ probs,
                                                   const int4* __restrict__ tiles) {
  constexpr int BI = 64, BJ = 64, BR = 32, LDP = BI + 1, LDQ = BJ + 1;
  __shared__ float lds[BR * LDP + BR * LDQ];
  float* Ps = lds;
  float* Qs = lds + BR * LDP;

  // flat tile list: {problem, split, i-tile, j-tile}; heavy problems first (built on the host)
  const int4 tl = tiles[blockIdx.x];
  const IgemmProb* __restrict__ pb = probs + blockIdx.x;   // one descriptor COPY per tile (see add_launch): no tile -> descriptor hop
  const int M = pb->M, N = pb->N, K = pb->K;
  const int i0 = tl.z * BI, j0 = tl.w * BJ;
  const int r_begin = tl.y * pb->k_chunk;
  const int r_end = min(K, r_begin + pb->k_chunk);

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wi = wave >> 1, wj = wave & 1;

  // ---- uniform operand descriptors (NP == 1: single-part problems, no per-element part selects)
  const gcf32 pB0 = (gcf32)pb->p_base[0];
  const gcf32 pB1 = (gcf32)pb->p_base[1];
  const gcf32 pB2 = (gcf32)pb->p_base[2];
  const int pLi0 = pb->p_ld_i[0], pLi1 = pb->p_ld_i[1], pLi2 = pb->p_ld_i[2];
  const int pLr0 = pb->p_ld_r[0], pLr1 = pb->p_ld_r[1], pLr2 = pb->p_ld_r[2];
  const int pk0 = NP == 1 ? 0x7fffffff : pb->p_k0, pk1 = NP == 1 ? 0x7fffffff : pb->p_k1;
  const gci32 pTi = (gci32)pb->p_tab_i;
  const gci32 pTr = (gci32)pb->p_tab_r;
  const gcu64 pVm = (gcu64)pb->p_vmask_i;
  const gcu8 pTap = (gcu8)pb->p_tap_r;
  const int ones_i = pb->p_ones_i;
  const gcf32 qB0 = (gcf32)pb->q_base[0];
  const gcf32 qB1 = (gcf32)pb->q_base[1];
  const gcf32 qB2 = (gcf32)pb->q_base[2];
  const int qLr0 = pb->q_ld_r[0], qLr1 = pb->q_ld_r[1], qLr2 = pb->q_ld_r[2];
  const int qLj0 = pb->q_ld_j[0], qLj1 = pb->q_ld_j[1], qLj2 = pb->q_ld_j[2];
  const gci32 qTr = (gci32)pb->q_tab_r;

  // ---- per-thread staging coordinates: 8 P + 8 Q elements per 32-deep slab.
  // P: lanes run along r (P_CONTIG_R: one r, rows p_il + 8e) or along i (one row, r = p_rl + 4e).
  const int p_rl = P_CONTIG_R ? (t & 31) : (t >> 6);
  const int p_il = P_CONTIG_R ? (t >> 5) : (t & 63);
  // Q: lanes run along j (Q_CONTIG_J: one column, r = q_rl + 4e) or along r (one r, cols q_jl + 8e).
  const int q_rl = Q_CONTIG_J ? (t >> 6) : (t & 31);
  const int q_jl = Q_CONTIG_J ? (t & 63) : (t >> 5);
  constexpr int PROWS = P_CONTIG_R ? 8 : 1;

  // row terms of the rows this thread stages (fixed for the whole reduction)
  const int p_tap_i = (PM == PM_TABLE_MASK && !P_CONTIG_R) ? (int)pTap[min(i0 + p_il, M - 1)] : 0;
  int p_row[PROWS];
  uint64_t p_vm[PROWS];
  bool p_iok[PROWS];
#pragma unroll
  for (int e = 0; e < PROWS; ++e) {
    const int i = i0 + p_il + 8 * e;
    p_iok[e] = (i < M) && (i != ones_i);
    const int ic = i < M ? i : i0;
    p_row[e] = PM != PM_AFFINE ? pTi[ic] : i;
    p_vm[e] = (PM == PM_TABLE_MASK && P_CONTIG_R) ? pVm[ic] : ~0ull;
  }

  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 acc;
#pragma unroll
  for (int x = 0; x < 16; ++x) acc[x] = 0.f;

  // The loads of slab k+1 are issued before the MFMAs of slab k and only waited for at the next LDS
  // write, so L2/HBM latency hides behind the matrix pipe.  The loop is rotated (first trip only
  // loads) so the staging code exists once.
  // Loaded values are kept RAW in registers; validity masks are applied only when the slab is
  // written to LDS on the next trip, so nothing touches a load result before the MFMAs are issued.
  float pv[8], qv[8];
  unsigned p_okm = 0, p_onem = 0, q_okm = 0;
  // table entries are fetched one slab ahead so their latency is never exposed
  constexpr int PCOLS = P_CONTIG_R ? 1 : 8;
  constexpr int QROWS = Q_CONTIG_J ? 8 : 1;
  int pcol_nx[PCOLS], ptap_nx = 0, qrow_nx[QROWS];
#pragma unroll
  for (int e = 0; e < PCOLS; ++e) {
    const int r = r_begin + p_rl + 4 * e;
    pcol_nx[e] = PM != PM_AFFINE ? pTr[r < r_end ? r : r_begin] : 0;
    if (PM == PM_TABLE_MASK && P_CONTIG_R) ptap_nx = pTap[r < r_end ? r : r_begin];
  }
#pragma unroll
  for (int e = 0; e < QROWS; ++e) {
    const int r = r_begin + q_rl + 4 * e;
    qrow_nx[e] = QM == QM_TABLE ? qTr[r < r_end ? r : r_begin] : 0;
  }
  for (int rr0 = r_begin - BR; rr0 < r_end; rr0 += BR) {
    if (rr0 >= r_begin) {
      __syncthreads();   // previous slab fully consumed
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int prl = p_rl + (P_CONTIG_R ? 0 : 4 * e);
        const int pil = p_il + (P_CONTIG_R ? 8 * e : 0);
        const float pvv = ((p_okm >> e) & 1u) ? pv[e] : 0.f;
        Ps[prl * LDP + pil] = ((p_onem >> e) & 1u) ? 1.f : pvv;
        const int qrl = q_rl + (Q_CONTIG_J ? 4 * e : 0);
        const int qjl = q_jl + (Q_CONTIG_J ? 0 : 8 * e);
        Qs[qrl * LDQ + qjl] = ((q_okm >> e) & 1u) ? qv[e] : 0.f;
      }
      __syncthreads();
    }
    if (rr0 + BR < r_end) {
      const int r0 = rr0 + BR;
      p_okm = 0; p_onem = 0; q_okm = 0;
      // ================= P
      if (P_CONTIG_R) {
        const int r = r0 + p_rl;
        const bool rok = r < r_end;
        const int rc = rok ? r : r_begin;
        const int part = NP == 1 ? 0 : (r >= pk0) + (r >= pk1);
        const gcf32 base = PM != PM_AFFINE ? pB0 : (part == 0 ? pB0 : (part == 1 ? pB1 : pB2));
        const int ldi = part == 0 ? pLi0 : (part == 1 ? pLi1 : pLi2);
        const int ldr = part == 0 ? pLr0 : (part == 1 ? pLr1 : pLr2);
        const int rstart = part == 0 ? 0 : (part == 1 ? pk0 : pk1);
        const int colterm = PM != PM_AFFINE ? pcol_nx[0] : (rc - rstart) * ldr;
        const int tap = PM == PM_TABLE_MASK ? ptap_nx : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          bool ok = rok && p_iok[e];
          if (PM == PM_TABLE_MASK) ok = ok && ((p_vm[e] >> tap) & 1ull);
          const long rowterm = PM != PM_AFFINE ? (long)p_row[e] : (long)p_row[e] * ldi;
          pv[e] = base[ok ? rowterm + colterm : 0l];
          p_okm |= ok ? (1u << e) : 0u;
          p_onem |= (i0 + p_il + 8 * e == ones_i && rok) ? (1u << e) : 0u;
        }
        if (PM != PM_AFFINE) {   // next slab's table entries (issued behind the data loads)
          const int rn = r + BR, rnc = rn < r_end ? rn : r_begin;
          pcol_nx[0] = pTr[rnc];
          if (PM == PM_TABLE_MASK) ptap_nx = pTap[rnc];
        }
      } else {
        int colterm[8];
        gcf32 base[8];
        int ldi[8];
        bool rok[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {   // table phase: every lookup issued before any data load
          const int r = r0 + p_rl + 4 * e;
          rok[e] = r < r_end;
          const int rc = rok[e] ? r : r_begin;
          const int part = NP == 1 ? 0 : (r >= pk0) + (r >= pk1);
          base[e] = PM != PM_AFFINE ? pB0 : (part == 0 ? pB0 : (part == 1 ? pB1 : pB2));
          ldi[e] = part == 0 ? pLi0 : (part == 1 ? pLi1 : pLi2);
          const int ldr = part == 0 ? pLr0 : (part == 1 ? pLr1 : pLr2);
          const int rstart = part == 0 ? 0 : (part == 1 ? pk0 : pk1);
          colterm[e] = PM != PM_AFFINE ? pcol_nx[P_CONTIG_R ? 0 : e] : (rc - rstart) * ldr;
          // P along i is only masked for weight gradients of padded convs: vmask by r, tap by i (p_mask_swap)
          if (PM == PM_TABLE_MASK) rok[e] = rok[e] && ((pVm[rc] >> p_tap_i) & 1ull);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool ok = rok[e] && p_iok[0];
          const long rowterm = PM != PM_AFFINE ? (long)p_row[0] : (long)p_row[0] * ldi[e];
          pv[e] = base[e][ok ? rowterm + colterm[e] : 0l];
          p_okm |= ok ? (1u << e) : 0u;
          p_onem |= (i0 + p_il == ones_i && (r0 + p_rl + 4 * e) < r_end) ? (1u << e) : 0u;
        }
        if (PM != PM_AFFINE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int rn = r0 + BR + p_rl + 4 * e;
            pcol_nx[P_CONTIG_R ? 0 : e] = pTr[rn < r_end ? rn : r_begin];
          }
        }
      }
      // ================= Q
      if (Q_CONTIG_J) {
        const int j = j0 + q_jl;
        const bool jok = j < N;
        int rowterm[8];
        gcf32 base[8];
        int ldj[8];
        bool rok[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int r = r0 + q_rl + 4 * e;
          rok[e] = r < r_end;
          const int rc = rok[e] ? r : r_begin;
          const int part = NP == 1 ? 0 : (r >= pk0) + (r >= pk1);
          base[e] = part == 0 ? qB0 : (part == 1 ? qB1 : qB2);
          const int ldr = part == 0 ? qLr0 : (part == 1 ? qLr1 : qLr2);
          ldj[e] = part == 0 ? qLj0 : (part == 1 ? qLj1 : qLj2);
          const int rstart = part == 0 ? 0 : (part == 1 ? pk0 : pk1);
          rowterm[e] = QM == QM_TABLE ? qrow_nx[Q_CONTIG_J ? e : 0] : (rc - rstart) * ldr;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool ok = rok[e] && jok;
          qv[e] = base[e][ok ? (long)rowterm[e] + (long)j * ldj[e] : 0l];
          q_okm |= ok ? (1u << e) : 0u;
        }
        if (QM == QM_TABLE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int rn = r0 + BR + q_rl + 4 * e;
            qrow_nx[Q_CONTIG_J ? e : 0] = qTr[rn < r_end ? rn : r_begin];
          }
        }
      } else {
        const int r = r0 + q_rl;
        const bool rok = r < r_end;
        const int rc = rok ? r : r_begin;
        const int part = NP == 1 ? 0 : (r >= pk0) + (r >= pk1);
        const gcf32 base = part == 0 ? qB0 : (part == 1 ? qB1 : qB2);
        const int ldr = part == 0 ? qLr0 : (part == 1 ? qLr1 : qLr2);
        const int ldj = part == 0 ? qLj0 : (part == 1 ? qLj1 : qLj2);
        const int rstart = part == 0 ? 0 : (part == 1 ? pk0 : pk1);
        const int rowterm = QM == QM_TABLE ? qrow_nx[0] : (rc - rstart) * ldr;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int j = j0 + q_jl + 8 * e;
          const bool ok = rok && (j < N);
          qv[e] = base[ok ? (long)rowterm + (long)j * ldj : 0l];
          q_okm |= ok ? (1u << e) : 0u;
        }
        if (QM == QM_TABLE) {
          const int rn = r + BR;
          qrow_nx[0] = qTr[rn < r_end ? rn : r_begin];
        }
      }
    }
    if (rr0 < r_begin) continue;
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
      float v = acc[x] * oscale + bj;
      if (accumulate) v += cbase[off];
      if (act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (act == ACT_LEAKY) v = v > 0.f ? v : alpha * v;
      if (rmask) v = rmask[mT ? (long)mT[i] + j : off] > 0.f ? v : alpha * v;
      cbase[off] = v;
    }
  }
}

#endif  // GRL_HOSTEMU
#endif  // GRL_GEMM_TYPES_ONLY

}  // namespace grl
