// q_chain.h -- the backward half of a DQN / BDQ update with the loss and the weight gradients INSIDE the row-local chains
// (stable-baselines `deepq` build_train as driven by /root/reference/manipulation_main/training/sb_helper.py:159-165, and the
// branching fork, :210-224; BASELINE configs[2]: gripper_grasp.yaml:104-118, batch 64).
//
// At batch 32 - 64 an update is a handful of launches at the launch floor (round 4: seven launches of 6 - 11 us for
// 0.01 GFLOP).  Two of them only existed because their work had been given a launch of its own:
//   * the loss (q_loss_kernel, 6.8 us): row-local -- a row's TD target needs the towers of all three networks for THAT row.
//     Every tower chain of q_bwd_towers forms the loss of its own 16 rows itself (one 16-lane group per row, four bins per
//     lane, the branches a loop over LDS operands; the 64-lane sums of q_loss_kernel are rebuilt from the same four DPP row
//     sums in the same order: same results bit for bit) and takes its own tower's output gradients from it; the value tower's workgroups also write what the rest of the update reads (TD errors,
//     priorities, the rows' partial sums of the metrics);
//   * the weight gradients (q_wgrad, 6.0 us): dW = x^T g over the batch rows is 4 k-steps of a 16x16x4 MFMA per 16 rows and
//     16x16 tile -- the chain that has just produced g for its 16 rows, with the layer's input rows still at hand, forms the
//     partial product of ITS rows and stores it as one slab per row block; the launch that applies the update sums the
//     B / 16 slabs in row-block order (ReduceDesc.splits = B / 16), exactly as it sums split-K slabs of the GEMM launches.
// Five launches per update instead of seven: sampler / gather, forward, tower chains, trunk chain, reduce + clip + Adam.
// With a trunk (BDQ) the towers' weight gradients leave the critical path as well: the tower chains store their gradient
// rows as before and (D + 1) extra workgroups per row block of the TRUNK launch form the slabs while the trunk chain runs.
//
// Summation order of a weight gradient: rows 4 s + q of a row block in MFMA order (quarter q, step s), row blocks in order --
// the GEMM launch it replaces sums the batch in its own tile order; both are checked against the oracle with the same
// tolerances (tests/test_gpu_q_parity.py; GRL_TUNE q_chain=0 keeps the seven launches).
#pragma once
#include "q_mfma.h"
#include "elem_kernels.h"
#include "per_kernels.h"

namespace grl {

struct QcLayer {          // weight gradient of one layer as row-block slabs
  float* dw; float* db;   // [n_rb][K, N], [n_rb][N]
  int K, N;
};
struct QcHead {
  QcLayer lay[GRL_MAX_LAYERS + 1];   // [li]: hidden layer li of the chain; [L]: its output layer (towers)
  const float* xin; int ld_xin;      // input rows of layer 0: the trunk's output (towers behind a trunk) or the observations
};
struct QChainArgs {
  QFusedArgs f;
  QLossArgs l;
  const QcHead* tw;       // [D+1]
  const QcHead* tr;       // trunk, or nullptr
  int late;               // 1 (networks with a trunk): the towers' weight gradients are formed by workgroups of their own in the
                          // TRUNK launch, in the shadow of the trunk chain, from the gradient rows the tower chains stored
  // Prioritised replay, multi-update calls (per_wb > 0: the minibatch size): the TRUNK launch also carries the priority
  // write-back of this minibatch (one workgroup) and the refresh of the block sums it touches (one workgroup per sample) --
  // both need only the TD errors the tower chains left behind -- so that the launch that ENDS the update can carry the sampler
  // of the next one (q_apply_kernels.h): four launches per update instead of five
  PerArgs per; const int64_t* per_idx; int per_wb;
};
// grid rows of the trunk launch: 1 (trunk chain) + D + 1 (late tower slabs) [+ 1 write-back + ceil(B / row blocks) refresh rows]
static inline int qc_trunk_rows(const QChainArgs& a) {
  const int nrb = (a.f.B + HT_RB - 1) / HT_RB;
  return (a.late ? a.f.D + 2 : 1) + (a.per_wb > 0 ? 1 + (a.per_wb + nrb - 1) / nrb : 0);
}
enum { QC_XW = 2 * QM_W, QC_XLD = QC_XW + 4 };

// host: the chains stage D x bins <= 256 advantages per row for the loss (bins <= 64: four per lane) and layer-0 inputs up to 128 wide
static inline bool qc_shape_ok(int n_bins, int D, int obs_dim) { return n_bins <= 64 && D * n_bins <= 256 && obs_dim <= QC_XW; }

#ifndef GRL_HEADS_TYPES_ONLY
#ifdef GRL_HOSTEMU
#include "q_chain_ref1.h"   // tests/hostemu: the emulation build only
#else

// Code size is a cost here: the first cut, with the loss unrolled over rows and branches and a weight-gradient call per stage,
// was 70 KB of straight-line code and took 38 us for the two launches that had taken 13.  Hence real loops over the branches
// and over the layers, with their operands in LDS.
enum { QC_DN = 256 };                      // D * bins of a row the loss staging holds
struct __attribute__((aligned(16))) QcLds {
  QmLds m;
  float x[GRL_MAX_LAYERS][HT_RB][QM_LD];   // x[li]: forward activations z_li of the rows = input of layer li + 1 / of the output layer
  float g[GRL_MAX_LAYERS][HT_RB][QM_LD];   // g[li]: gradient rows of layer li, kept for the weight gradients at the end of the chain
  float xin[HT_RB][QC_XLD];                // input of layer 0
  QcHead y;
};
struct __attribute__((aligned(16))) QcLossLds {
  float q[3][HT_RB * QC_DN];               // the rows' advantages [r][D * n]: selector net (arg max), target net, online net
  float rs[HT_RB][16];                     // per row: V'(target), V(online), reward, done, importance weight, -, -, -, stored bins [D]
  float qsel[HT_RB][8];                    // per row and branch: the online net's Q of the stored bin
};

// dW partial of one layer over the 16 rows of this workgroup: X[r][k] the layer's input rows, G[r][n] the gradient w.r.t. its
// pre-activations (both in LDS, zero beyond the batch / the widths).  Wave w owns output columns 16 w .. 16 w + 15 and walks
// the input features 16 at a time: A = X^T (m = feature, k = row), B = G (k = row, n = column), 4 steps of 4 rows.
__device__ __forceinline__ void qc_wgrad(float* dw, float* db, int K, int N, int rb, const float* X, int ldx, const float* G) {
  const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63, c = l & 15, q = l >> 4;
  // the 16-column blocks go round the waves; with one or two of them the waves left over share the feature tiles (P phases)
  const int nNt = (N + 15) >> 4;
  const int P = nNt == 1 ? 4 : nNt == 2 ? 2 : 1;
  const int nt = nNt >= 3 ? w : (w & (nNt - 1)), phase = nNt >= 3 ? 0 : (w >> (nNt - 1));
  if (nt < nNt) {
    const int n = 16 * nt + c;
    float bv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) bv[s] = G[(4 * s + q) * QM_LD + n];
    const __amdgpu_buffer_rsrc_t rd = i2_rsrc(dw + (long)rb * K * N);     // (branch-free stores: out-of-range offsets are dropped)
    // four feature tiles per pass, independent accumulators: their LDS reads are one batch, their MFMAs issue back to back
    // (one tile at a time every pass was a read -> wait -> 4 dependent MFMAs -> store chain of ~600 cycles)
    for (int m0 = 16 * phase; m0 < K; m0 += 64 * P) {
      float av[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int mu = m0 + 16 * P * u;
        const int col = mu < K ? mu + c : c;               // (tiles beyond K: any in-range address, result not stored)
#pragma unroll
        for (int s = 0; s < 4; ++s) av[u][s] = X[(4 * s + q) * ldx + col];
      }
      qm_f4 acc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = qm_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][s], bv[s], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int mu = m0 + 16 * P * u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = mu + 4 * q + i;
          // (pinned in a vector register first: stored straight from the accumulation registers, this compiler wrote element 0
          //  of the tile four times)
          float v = acc[u][i];
          asm volatile("" : "+v"(v));
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rd,
                                                (k < K && n < N) ? (k * N + n) * 4 : I2_OOB, 0, 0);
        }
      }
    }
  }
  if (t < N && t < QM_W) {      // bias: the column sums, rows in order
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < HT_RB; ++r) sum += G[r * QM_LD + t];
    QM_GW(db)[(long)rb * N + t] = sum;
  }
}
// every weight gradient of the chain, at its end: layers 0 .. L - 1 from the parked gradient rows, then (towers) the output layer
__device__ __forceinline__ void qc_wgrad_all(QcLds& s, int rb, int L, bool out_layer) {
  for (int li = 0; li < L + (out_layer ? 1 : 0); ++li) {
    const QcLayer& yl = s.y.lay[li];
    const int K = __builtin_amdgcn_readfirstlane(yl.K), N = __builtin_amdgcn_readfirstlane(yl.N);
    const float* X = li == 0 ? &s.xin[0][0] : &s.x[li - 1][0][0];
    const float* G = li < L ? &s.g[li][0][0] : &s.m.o[0][0];
    qc_wgrad(yl.dw, yl.db, K, N, rb, X, li == 0 ? QC_XLD : QM_LD, G);
  }
}

// input rows of layer 0 (K <= 128 columns of `xin`), 16 threads per row: requested with the chain's operands, parked in LDS later
__device__ __forceinline__ void qc_xin_load(const QcHead& y, int row0, int B, float (&xv)[8]) {
  const int t = threadIdx.x, row = row0 + (t >> 4), K = y.lay[0].K;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int col = (t & 15) + 16 * j;
    xv[j] = (row < B && col < K) ? QM_G(y.xin)[(long)row * y.ld_xin + col] : 0.f;
  }
}
__device__ __forceinline__ void qc_xin_store(QcLds& s, const float (&xv)[8]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 8; ++j) s.xin[t >> 4][(t & 15) + 16 * j] = xv[j];
}
// the forward activations of the rows (result layout, registers of the backward chain) -> LDS rows
__device__ __forceinline__ void qc_park_z(QcLds& s, const float (&zm)[GRL_MAX_LAYERS][4], int L) {
  const int t = threadIdx.x, w = t >> 6, l = t & 63, c = l & 15, q = l >> 4, n = 16 * w + c;
#pragma unroll
  for (int li = 0; li < GRL_MAX_LAYERS; ++li)
    if (li < L) {
#pragma unroll
      for (int i = 0; i < 4; ++i) s.x[li][4 * q + i][n] = zm[li][i];
    }
}
// g_li (just arrived in s.m.z[li & 1]) -> its own rows: the ping-pong buffer is overwritten two stages later
__device__ __forceinline__ void qc_park_g(QcLds& s, int li) {
  const int t = threadIdx.x, r = t >> 4, c4 = 4 * (t & 15);
  *(qm_f4*)&s.g[li][r][c4] = *(const qm_f4*)&s.m.z[li & 1][r][c4];
}
__device__ __forceinline__ void qc_head_to_lds(QcLds& s, const QcHead* y) {
  const int t = threadIdx.x;
  if (t < (int)(sizeof(QcHead) / 4)) ((float*)&s.y)[t] = QM_G((const float*)y)[t];     // (bit copy)
}

// sum over the 16 lanes of a DPP row, left in all of them: the first four steps of q_wave_sum_dpp -- the same partners, the
// same order, so the sum of bins 16 j .. 16 j + 15 is bit for bit the row value that kernel forms
__device__ __forceinline__ float qc_sum16(float v) {
  v += q_dpp<Q_DPP_XOR1>(v);
  v += q_dpp<Q_DPP_XOR2>(v);
  v += q_dpp<Q_DPP_HALF_MIRROR>(v);
  v += q_dpp<Q_DPP_MIRROR>(v);
  return v;
}
// The TD loss of the 16 rows of a row block, one 16-lane group per row (lane i holds bins i, 16 + i, 32 + i, 48 + i), the
// branches a loop over LDS operands.  Arithmetic of q_row_loss / q_loss_kernel expression for expression; the 64-lane sums of
// that kernel are (r0 + r1) + (r2 + r3) over its four DPP rows, formed here from the same four row sums.  g_own / ai_own
// receive the loss gradient scale and the stored bin of branch `own`; emit_td(d, td) and emit_row(dv, priority, weighted
// loss, mean selected Q) are called by every lane with the values of its row.
template <class EmitTd, class EmitRow>
__device__ __forceinline__ void qc_rows_loss(const QLossArgs& a, QcLossLds& ql, int own, float& g_own, int& ai_own, EmitTd&& emit_td,
                                             EmitRow&& emit_row) {
  const int t = threadIdx.x, r = t >> 4, i = t & 15;
  const int D = a.D, n = a.n, DN = D * n;
  const float invB = 1.f / (float)a.B, invD = 1.f / (float)D, invn = 1.f / (float)n;
  const float v2b = ql.rs[r][0], v0b = ql.rs[r][1], rewb = ql.rs[r][2], doneb = ql.rs[r][3], w = ql.rs[r][4];
  bool on[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) on[j] = 16 * j + i < n;
  // ---- one pass over the branches: target part (arg max of the selector net, target value there, mean) and the online
  // net's selected Q.  Every LDS read is unconditional (reads beyond a row's bins stay inside the staging block and are
  // discarded by a select) so that the twelve of a branch are ONE batch; the arg max is branch-free.
  float qbest = 0.f;
  for (int d = 0; d < D; ++d) {
    const int o = r * DN + d * n + i;
    float xs[4], xt[4], x0[4], rt[4], r0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { xs[j] = ql.q[0][o + 16 * j]; xt[j] = ql.q[1][o + 16 * j]; x0[j] = ql.q[2][o + 16 * j]; }
    const int ai = (int)ql.rs[r][8 + d];
    float sv = -INFINITY;
    int si = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xs[j] = on[j] ? xs[j] : -INFINITY; xt[j] = on[j] ? xt[j] : 0.f; x0[j] = on[j] ? x0[j] : 0.f;
      const bool up = xs[j] > sv;                // (ascending bins: the first maximum stays)
      sv = up ? xs[j] : sv; si = up ? 16 * j + i : si;
      rt[j] = qc_sum16(xt[j]);
      r0[j] = qc_sum16(x0[j]);
    }
#define QC_AM_STEP(CTRL)                                                    \
    {                                                                       \
      const float ov = q_dpp<CTRL>(sv);                                     \
      const int oi = q_dpp_i<CTRL>(si);                                     \
      const bool take = (ov > sv) | ((ov == sv) & (oi < si));               \
      sv = take ? ov : sv; si = take ? oi : si;                             \
    }
    QC_AM_STEP(Q_DPP_XOR1) QC_AM_STEP(Q_DPP_XOR2) QC_AM_STEP(Q_DPP_HALF_MIRROR) QC_AM_STEP(Q_DPP_MIRROR)
#undef QC_AM_STEP
    const float mean2 = (rt[0] + rt[1]) + (rt[2] + rt[3]);
    const float mean0 = ((r0[0] + r0[1]) + (r0[2] + r0[3])) * invn;
    float pt = 0.f, p0 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                // (one lane, one j: the sums below are exact)
      pt += (16 * j + i == si) ? xt[j] : 0.f;
      p0 += (16 * j + i == ai) ? x0[j] : 0.f;
    }
    pt = qc_sum16(pt);
    p0 = qc_sum16(p0);
    qbest += v2b + pt - mean2 * invn;
    if (i == 0) ql.qsel[r][d] = v0b + p0 - mean0;
  }
  qbest *= invD;
  const float y = rewb + a.gamma * (1.f - doneb) * qbest;
  float dv = 0.f, prio = 0.f, lossb = 0.f, qs = 0.f;
  for (int d = 0; d < D; ++d) {                  // (same wave wrote them: LDS operations of a wave complete in order)
    const float q_sel = ql.qsel[r][d];
    const int ai = (int)ql.rs[r][8 + d];
    const float tdv = q_sel - y;
    prio += fabsf(tdv);
    qs += q_sel;
    float err, dfd;
    if (a.huber) {
      const float at = fabsf(tdv);
      err = at < 1.f ? 0.5f * tdv * tdv : at - 0.5f;
      dfd = fminf(fmaxf(tdv, -1.f), 1.f);
    } else {
      err = tdv * tdv;
      dfd = 2.f * tdv;
    }
    lossb += err;
    const float g = w * invB * (a.loss_sum ? 1.f : invD) * dfd;
    dv += g;
    g_own = d == own ? g : g_own;
    ai_own = d == own ? ai : ai_own;
    emit_td(d, tdv);
  }
  emit_row(dv, prio, w * lossb * (a.loss_sum ? 1.f : invD), qs * invD);
}

// grid (B/16, D+1): loss of the rows -> tower output + hidden layers of the online net -> gradient w.r.t. the trunk output
// (one partial per tower) -> weight-gradient slabs of the tower
__global__ __launch_bounds__(256) void q_bwd_towers_chain_kernel(QChainArgs ca) {
  __shared__ QcLds s;
  __shared__ QcLossLds ql;
  const QFusedArgs& a = ca.f;
  const QLossArgs& la = ca.l;
  const int rb = blockIdx.x, row0 = rb * HT_RB, tw = blockIdx.y, t = threadIdx.x;
  const HtHead& h = a.bwd_tw[tw];
  const int D = a.D, nbins = la.n, DN = D * nbins;
  // ---- everything this workgroup reads travels with the chain's operands: the rows' advantages of the three networks (a
  // contiguous block each), one per-row scalar per thread, the input rows of layer 0
  constexpr int NLQ = HT_RB * QC_DN / 256;
  float lq[3][NLQ];
  {
    const int cnt = (a.B - row0 < HT_RB ? a.B - row0 : HT_RB) * DN;
    const float* src[3] = {la.double_q ? la.adv1 : la.adv2, la.adv2, la.adv0};
#pragma unroll
    for (int k3 = 0; k3 < 3; ++k3) {
      const __amdgpu_buffer_rsrc_t rq = i2_rsrc(src[k3] + (long)row0 * DN);
#pragma unroll
      for (int k = 0; k < NLQ; ++k) {
        const int e = t + 256 * k;
        lq[k3][k] = 256 * k < HT_RB * DN
                        ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rq, e < cnt ? e * 4 : I2_OOB, 0, 0))
                        : 0.f;
      }
    }
  }
  float rsv = 0.f;
  {
    const int r = t >> 4, k = t & 15, b = row0 + r;
    const float* p = k == 0 ? la.v2 + b : k == 1 ? la.v0 + (long)b : k == 2 ? la.rew + b : k == 3 ? la.done + b : k == 4 ? la.weights + b
                     : (k >= 8 && k - 8 < D) ? la.act + (long)b * D + (k - 8) : nullptr;
    if (b < a.B && p) rsv = *QM_G(p);
  }
  float xv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (!ca.late) {         // (the rows the weight gradients need: formed elsewhere in the late form)
    qc_xin_load(ca.tw[tw], row0, a.B, xv);
    qc_head_to_lds(s, ca.tw + tw);
  }
  qm_bwd_head(
      h, row0, a.B, s.m, h.n_xa ? a.dh_part + (long)tw * a.B * a.Ht : nullptr, a.Ht, nullptr, 0, 0, 1.f,
      [&](const auto& zm) {
        qm_zero(s.m);
        if (!ca.late) { qc_park_z(s, zm, h.L); qc_xin_store(s, xv); }
#pragma unroll
        for (int k3 = 0; k3 < 3; ++k3)
#pragma unroll
          for (int k = 0; k < NLQ; ++k)
            if (256 * k < HT_RB * DN) ql.q[k3][t + 256 * k] = lq[k3][k];
        ql.rs[t >> 4][t & 15] = rsv;
        __syncthreads();
        const int r = t >> 4, i = t & 15, b = row0 + r;
        const float invD = 1.f / (float)D, invn = 1.f / (float)nbins;
        const bool live = b < a.B;           // (rows beyond the batch: operands are zeros, nothing is stored, s.m.o keeps its zeros)
        float g_own = 0.f;
        int ai_own = 0;
        qc_rows_loss(
              la, ql, tw, g_own, ai_own,
              [&](int d, float tdv) {
                if (tw == D && live && i == 0) QM_GW(la.td)[b * D + d] = tdv;
              },
              [&](float dv, float prio, float wloss, float qsel) {
                if (tw == D && live && i == 0) {
                  QM_GW(la.priority)[b] = prio;
                  QM_GW(la.row_part)[3 * b] = wloss;
                  QM_GW(la.row_part)[3 * b + 1] = qsel;
                  QM_GW(la.row_part)[3 * b + 2] = prio * invD;
                  s.m.o[r][0] = dv;
                  if (ca.late) QM_GW(la.d_v0)[(long)b * la.ld_dv] = dv;
                }
              });
        // the tower's own output gradients (to memory as well when the weight gradients are formed in the trunk launch)
        if (tw < D && live) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = 16 * j + i;
            const float ga = g_own * ((k == ai_own ? 1.f : 0.f) - invn);
            if (k < nbins) {
              s.m.o[r][k] = ga;
              if (ca.late) QM_GW(la.d_adv0)[((long)b * D + tw) * la.nbp + k] = ga;
            }
          }
        }
        __syncthreads();
      },
      [&](int li) { if (!ca.late) qc_park_g(s, li); });
  if (ca.late) return;
  __syncthreads();
  qc_wgrad_all(s, rb, h.L, true);
}

// grid (B/16): trunk -- partials added in tower order, scaled, masked, propagated down to layer 0; weight-gradient slabs
__global__ __launch_bounds__(256) void q_bwd_trunk_chain_kernel(QChainArgs ca) {
  __shared__ QcLds s;
  const QFusedArgs& a = ca.f;
  const int rb = blockIdx.x, row0 = rb * HT_RB;
  const int y_chain = ca.late ? a.D + 2 : 1;       // grid rows of the chain work; behind them the prioritised-replay rows
  if (ca.per_wb > 0 && (int)blockIdx.y >= y_chain) {
    static_assert(sizeof(QcLds) >= (2 * PER_BLK + 256) * sizeof(double) + 1024 * sizeof(int64_t), "the chain's LDS doubles as the trees' scratch");
    double* tr = (double*)&s;
    if ((int)blockIdx.y == y_chain) {              // priority write-back of this minibatch (per_update_kernel's work)
      if (rb == 0) {
        per_update_body(ca.per, ca.per_idx, (int64_t*)tr, (float*)(tr + 1024));
        // the minibatch is consumed (the tower chains formed its loss): the Philox counter moves on HERE, a launch before the
        // next update's sampler reads it on the apply launch (whose own sums then leave it alone: finish bit 1)
        if (threadIdx.x == 0) q_rng_tick(ca.per.sc);
      }
      return;
    }
    const int k = ((int)blockIdx.y - y_chain - 1) * (int)gridDim.x + rb;      // block sums the next sampler reads
    if (k < ca.per_wb) per_refresh_body(ca.per, ca.per_idx, k, tr, tr + 2 * PER_BLK, (int64_t*)(tr + 2 * PER_BLK + 256));
    return;
  }
  if (blockIdx.y > 0) {
    // ---- grid row 1 + tw: the weight gradients of tower tw for this row block, from what the tower chains left in memory
    // (activations, gradient rows, output gradients): one batch of loads, the rows parked in LDS, qc_wgrad_all
    const int tw = blockIdx.y - 1, t = threadIdx.x, w = t >> 6, l = t & 63, c = l & 15, q = l >> 4, n = 16 * w + c;
    const HtHead& h = a.bwd_tw[tw];
    const int L = h.L;
    float xv[8], zm[GRL_MAX_LAYERS][4], gm[GRL_MAX_LAYERS][4], dv[4] = {0.f, 0.f, 0.f, 0.f};
    qc_xin_load(ca.tw[tw], row0, a.B, xv);
    qc_head_to_lds(s, ca.tw + tw);
#pragma unroll
    for (int li = 0; li < GRL_MAX_LAYERS; ++li)
      if (li < L) {
        const int H = h.hid[li];
        const float* zp = li == 0 ? h.z0 : h.z[li];
        const float* gp = li == 0 ? h.g0 : h.g[li];
        const int ldg = li == 0 ? h.ldg0 : H;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = row0 + 4 * q + i;
          const bool ok = row < a.B && n < H;
          zm[li][i] = ok ? QM_G(zp)[(long)row * H + n] : 0.f;
          gm[li][i] = ok ? QM_G(gp)[(long)row * ldg + n] : 0.f;
        }
      }
    const int r = t >> 4, row = row0 + r, o0 = t & 15;
    if (tw < a.D) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = o0 + 16 * j;
        if (row < a.B && o < a.nb) dv[j] = QM_G(a.d_adv)[((long)row * a.D + tw) * a.nbp + o];
      }
    } else if (o0 == 0 && row < a.B) {
      dv[0] = QM_G(a.d_v)[(long)row * a.ld_dv];
    }
    qc_park_z(s, zm, L);
#pragma unroll
    for (int li = 0; li < GRL_MAX_LAYERS; ++li)
      if (li < L) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s.g[li][4 * q + i][n] = gm[li][i];
      }
    qc_xin_store(s, xv);
#pragma unroll
    for (int j = 0; j < 4; ++j) s.m.o[r][o0 + 16 * j] = dv[j];
    __syncthreads();
    qc_wgrad_all(s, rb, L, true);
    return;
  }
  const HtHead& h = *a.bwd_tr;
  float xv[8];
  qc_xin_load(*ca.tr, row0, a.B, xv);
  qc_head_to_lds(s, ca.tr);
  qm_bwd_head(
      h, row0, a.B, s.m, nullptr, 0, a.dh_part, a.D + 1, (long)a.B * a.Ht, a.trunk_scale,
      [&](const auto& zm) {
        qm_zero(s.m);
        qc_park_z(s, zm, h.L);
        qc_xin_store(s, xv);
        __syncthreads();
      },
      [&](int li) { qc_park_g(s, li); });
  __syncthreads();
  qc_wgrad_all(s, rb, h.L, false);
}

#endif  // GRL_HOSTEMU
#endif  // GRL_HEADS_TYPES_ONLY

}  // namespace grl
