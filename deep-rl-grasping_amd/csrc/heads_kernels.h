// heads_kernels.h -- row-local kernels for the actor / critic MLP heads of the SAC update.
//
// The heads (stable-baselines sac/policies.py FeedForwardPolicy.make_actor / make_critics as wired by
// /root/reference/manipulation_main/training/sb_helper.py:87-96) are tiny: after the
// first layer (K = 513 features, done by the implicit GEMM as a bias-free partial pre-activation `u`)
// every remaining layer is a 64x64-ish product per batch row.  Launching them as GEMMs costs ~20 us of
// latency per layer; here one workgroup owns 16 batch rows and walks a whole chain of heads with the
// activations kept transposed in LDS:
//
//   forward  chain 0: pi head -> squashed-Gaussian sample -> qf1(s, pi)   (chain 5: the same with qf2(s, pi);
//            the pi head is cheap to recompute and the critical path loses a whole head)
//            chains 1..4: vf, qf1(s, a), qf2(s, a), target vf
//   backward chain 0: qf1(s, pi) backward -> d a_pi -> sample backward -> pi head backward
//            chains 1..3: vf, qf1, qf2 (gradients w.r.t. every pre-activation; the layer-0 ones feed the
//            feature-gradient GEMM and the weight-gradient GEMMs)
//
// Arithmetic order per output element is the k-sequential fmaf chain the MFMA kernels produce
// (partial sum, then the remaining inputs in order, bias last), so moving a head between this path
// and the GEMM path does not change its bits.
#pragma once
#include "elem_kernels.h"
#include "../../include/grl.h"

namespace grl {

enum { HT_RB = 16, HT_MAXW = 128, HT_MAXA = 64 };

struct HtHead {
  // layer 0: z0 = relu(u + xa . w0a + b0)
  const float* u; int ldu;                 // [B, ldu] feature part of the pre-activation (no bias) ...
  int u_split; long u_stride;              // ... as u_split partial sums, u_stride floats apart (0 / 1: a single one), added in order
  const float* xa; int ld_xa; int n_xa;    // second input part (the action), n_xa == 0: none
  const float* w0a;                        // [n_xa, H0]: rows of the layer-0 kernel after the feature rows
  const float* b0;
  float* z0;                               // [B, H0]
  float* g0; int ldg0;                     // backward: gradient w.r.t. the layer-0 pre-activation
  int H0, L;
  // layers 1 .. L-1 (index l): z_l = relu(z_{l-1} . w[l] + b[l]),  w[l] is [hid[l-1], hid[l]]
  const float* w[GRL_MAX_LAYERS]; const float* b[GRL_MAX_LAYERS];
  float* z[GRL_MAX_LAYERS]; float* g[GRL_MAX_LAYERS];
  int hid[GRL_MAX_LAYERS];                 // hid[0] == H0
  // output layers (1 for vf / qf, 2 for pi: mu and log_std), each [hid[L-1], out_dim]
  int n_out, out_dim;
  const float* ow[2]; const float* ob[2]; float* out[2];
  int ld_out;                              // row stride of out[k] (0: out_dim); activations z0 / z[l] may be nullptr (not stored)
  const float* dout[2]; int ld_dout;       // backward: gradient w.r.t. the outputs [B, out_dim], row stride ld_dout
  float* da; int ld_da;                    // backward: gradient w.r.t. xa (nullptr: not needed)
};

struct HeadsFwdArgs {
  HtHead h[7];            // 0 pi, 1 vf, 2 qf1(a), 3 qf2(a), 4 target vf, 5 qf1(pi), 6 qf2(pi)
  int B, A;
  const float* eps;       // [B, A]
  float* pi_a; float* logp; float* ent;
};

struct HeadsBwdArgs {
  HtHead h[5];            // 0 pi, 1 vf, 2 qf1(a), 3 qf2(a), 4 qf1(pi)
  int B, A;
  const float* mu; const float* ls_raw; const float* eps; const float* pi_a;
  const float* log_ent_coef;
  float* da_pi;           // [B, A] (debug tap / split API)
  float* dmu; float* dls; // [B, A], row stride ld_dm
  int ld_dm;
  // loss inputs: the output gradients of the critics are row-local (A.4), so they are formed here
  //   d qf_i = (qf_i - (r + (1-d) gamma V_tgt)) / B,  d V = (V - (min(qf1_pi, qf2_pi) - alpha logp)) / B,  d qf1_pi = -1/B
  const float* rew; const float* done; const float* v_tgt; const float* qf1; const float* qf2; const float* v;
  const float* qf1_pi; const float* qf2_pi; const float* logp;
  float gamma;
  float* d_out[5];        // [B] (stride ld_d) per chain head index: 1 d_v, 2 d_qf1, 3 d_qf2, 4 d_qf1_pi
  int ld_d;
};

__device__ __forceinline__ float ht_dout(const HeadsBwdArgs& a, int head, int row, float alpha) {
  const float invB = 1.f / (float)a.B;
  if (head == 4) return -invB;
  if (head == 1) {
    const float vb = fminf(a.qf1_pi[row], a.qf2_pi[row]) - alpha * a.logp[row];
    return (a.v[row] - vb) * invB;
  }
  const float qb = a.rew[row] + (1.f - a.done[row]) * a.gamma * a.v_tgt[row];
  return ((head == 2 ? a.qf1[row] : a.qf2[row]) - qb) * invB;
}

// ------------------------------------------------------------------------------------------------
// per-element pieces of the squashed-Gaussian sample and its backward (same arithmetic as
// sample_kernel / sample_bwd_kernel in elem_kernels.h)
__device__ __forceinline__ float ht_sample_elem(float m, float ls_raw, float ep, float& logp, float& ent) {
  const float ls = fminf(fmaxf(ls_raw, GRL_LOG_STD_MIN), GRL_LOG_STD_MAX);
  const float sd = expf(ls);
  const float u = m + ep * sd;
  const float z = (u - m) / (sd + GRL_EPS);
  logp += -0.5f * (z * z + 2.f * ls + 1.8378770664093453f);
  ent += ls + 1.4189385332046727f;
  const float t = tanhf(u);
  logp -= logf(1.f - t * t + GRL_EPS);
  return t;
}
__device__ __forceinline__ void ht_sample_bwd_elem(float lr, float ep, float t, float da, float alpha_over_b,
                                                   float& dmu, float& dls) {
  const float ls = fminf(fmaxf(lr, GRL_LOG_STD_MIN), GRL_LOG_STD_MAX);
  const float sd = expf(ls);
  const float omt = 1.f - t * t;
  const float du = da * omt + alpha_over_b * (2.f * t * omt / (omt + GRL_EPS));
  dmu = du;
  const float den = sd + GRL_EPS;
  const float z = sd * ep / den;
  const float dz_dls = ep * sd * GRL_EPS / (den * den);
  float d = du * sd * ep + alpha_over_b * (-(z * dz_dls + 1.f));
  if (lr < GRL_LOG_STD_MIN || lr > GRL_LOG_STD_MAX) d = 0.f;
  dls = d;
}

// per-row forms (hostemu reference)
__device__ __forceinline__ void ht_sample_row(const float* mu, const float* ls_raw, const float* eps, int A,
                                              float* pi, float* logp_out, float* ent_out) {
  float logp = 0.f, ent = 0.f;
  for (int j = 0; j < A; ++j) {
    const float m = mu[j];
    const float ls = fminf(fmaxf(ls_raw[j], GRL_LOG_STD_MIN), GRL_LOG_STD_MAX);
    const float sd = expf(ls);
    const float u = m + eps[j] * sd;
    const float z = (u - m) / (sd + GRL_EPS);
    logp += -0.5f * (z * z + 2.f * ls + 1.8378770664093453f);
    ent += ls + 1.4189385332046727f;
    const float t = tanhf(u);
    logp -= logf(1.f - t * t + GRL_EPS);
    pi[j] = t;
  }
  *logp_out = logp;
  *ent_out = ent;
}

__device__ __forceinline__ void ht_sample_bwd_row(const float* ls_raw, const float* eps, const float* pi,
                                                  const float* da, int A, float alpha_over_b, float* dmu,
                                                  float* dls) {
  for (int j = 0; j < A; ++j) {
    const float lr = ls_raw[j];
    const float ls = fminf(fmaxf(lr, GRL_LOG_STD_MIN), GRL_LOG_STD_MAX);
    const float sd = expf(ls);
    const float ep = eps[j];
    const float t = pi[j];
    const float omt = 1.f - t * t;
    const float du = da[j] * omt + alpha_over_b * (2.f * t * omt / (omt + GRL_EPS));
    dmu[j] = du;
    const float den = sd + GRL_EPS;
    const float z = sd * ep / den;
    const float dz_dls = ep * sd * GRL_EPS / (den * den);
    float d = du * sd * ep + alpha_over_b * (-(z * dz_dls + 1.f));
    if (lr < GRL_LOG_STD_MIN || lr > GRL_LOG_STD_MAX) d = 0.f;
    dls[j] = d;
  }
}

#ifndef GRL_HEADS_TYPES_ONLY    // (engine.hip needs the argument blocks only: the kernels live in heads.hip)
#ifdef GRL_HOSTEMU
#include "heads_kernels_ref1.h"   // tests/hostemu: the emulation build only
#else  // ------------------------------------------------------------------------------------ device

typedef float ht_f4 __attribute__((ext_vector_type(4)));

enum { HT_LDW = HT_MAXW + 1 };   // odd row stride: column reads (forward) and row reads (backward) are both conflict-free
struct __attribute__((aligned(16))) HtLds {
  float W[HT_MAXW * HT_LDW + 128]; // the layer kernel(s), staged once per (head, layer); two [128, 64+1] output kernels fit
  float zT[2][HT_MAXW][HT_RB];      // activations / gradients of the current and the next layer, [col][row]
  float xaT[HT_MAXA][HT_RB];        // second input part (action) of the rows, [a][row]
  float oT[2 * HT_MAXA][HT_RB];     // outputs (forward: mu | log_std) / output gradients (backward), [k*out_dim + o][row]
};

// kernel [Hin, Hout] (row-major) -> LDS W[k * (Hout+1) + n]; all loads are issued before the first write
__device__ __forceinline__ void ht_stage_w_at(const float* w, int Hin, int Hout, float* W) {
  const int t = threadIdx.x;
  const int ld = Hout + 1;
  __syncthreads();   // the previous stage may still be reading W
  if ((Hout & 3) == 0) {
    const int nq = Hin * Hout / 4;            // <= 4096 quads -> <= 16 per thread
    ht_f4 r[16];
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (t + 256 * e < nq) r[e] = *(const ht_f4*)(w + 4 * (t + 256 * e));
    // element 4*(t + 256 e) -> (row k, column c): one division per thread, then incremental
    int k = (4 * t) / Hout, c = 4 * t - k * Hout;
    const int dk = 1024 / Hout, dc = 1024 - dk * Hout;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (t + 256 * e < nq) {
        float* d = W + k * ld + c;
        d[0] = r[e].x; d[1] = r[e].y; d[2] = r[e].z; d[3] = r[e].w;
      }
      k += dk; c += dc;
      if (c >= Hout) { c -= Hout; ++k; }
    }
  } else {
    int k = t / Hout, c = t - k * Hout;
    const int dk = 256 / Hout, dc = 256 - dk * Hout;
    for (int idx = t; idx < Hin * Hout; idx += 256) {
      W[k * ld + c] = w[idx];
      k += dk; c += dc;
      if (c >= Hout) { c -= Hout; ++k; }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void ht_stage_w(const float* w, int Hin, int Hout, HtLds& s) { ht_stage_w_at(w, Hin, Hout, s.W); }

// The kernel of the NEXT stage travels global -> registers while the current stage computes (kernels of up
// to 4096 floats, i.e. 64x64; larger or odd-sized ones are copied at commit time without overlap).
struct HtW { ht_f4 r[4]; };
__device__ __forceinline__ bool ht_w_fast(int n) { return (n & 3) == 0 && n <= 4096; }
__device__ __forceinline__ void ht_w_fetch(HtW& p, const float* w, int n) {
  if (!ht_w_fast(n)) return;
  const int t = threadIdx.x, nq = n >> 2;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (256 * e < nq) p.r[e] = *(const ht_f4*)(w + 4 * min(t + 256 * e, nq - 1));
}
__device__ __forceinline__ void ht_w_commit(const HtW& p, const float* w, int Hin, int Hout, float* W) {
  const int n = Hin * Hout;
  if (!ht_w_fast(n)) { ht_stage_w_at(w, Hin, Hout, W); return; }
  const int t = threadIdx.x, ld = Hout + 1, nq = n >> 2;
  __syncthreads();   // the previous stage may still be reading W
  int k = (4 * t) / Hout, c = 4 * t - k * Hout;
  const int dk = 1024 / Hout, dc = 1024 - dk * Hout;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (t + 256 * e < nq) {
      const float v[4] = {p.r[e].x, p.r[e].y, p.r[e].z, p.r[e].w};
      int kk = k, cc = c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        W[kk * ld + cc] = v[i];
        if (++cc == Hout) { cc = 0; ++kk; }
      }
    }
    k += dk; c += dc;
    if (c >= Hout) { c -= Hout; ++k; }
  }
  __syncthreads();
}

// acc[r] += sum_k x[k][r] * w[k * wstride]  (k ascending: the fmaf chain order of the GEMM path).
// x points at zT[..][0][4*rg]: 4 consecutive rows of column k are one 16-byte LDS read (row stride HT_RB).
// Operands of 8 k-steps are fetched before the first fma: otherwise every step exposes the LDS latency.
__device__ __forceinline__ void ht_dot(float (&acc)[4], const float* w, int wstride, const float* x, int K) {
  int k = 0;
  for (; k + 8 <= K; k += 8) {
    float wv[8];
    ht_f4 xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      wv[u] = w[(k + u) * wstride];
      xv[u] = *(const ht_f4*)(x + (k + u) * HT_RB);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0] = fmaf(xv[u].x, wv[u], acc[0]); acc[1] = fmaf(xv[u].y, wv[u], acc[1]);
      acc[2] = fmaf(xv[u].z, wv[u], acc[2]); acc[3] = fmaf(xv[u].w, wv[u], acc[3]);
    }
  }
  for (; k < K; ++k) {
    const float wv = w[k * wstride];
    const ht_f4 xv = *(const ht_f4*)(x + k * HT_RB);
    acc[0] = fmaf(xv.x, wv, acc[0]); acc[1] = fmaf(xv.y, wv, acc[1]);
    acc[2] = fmaf(xv.z, wv, acc[2]); acc[3] = fmaf(xv.w, wv, acc[3]);
  }
}

// sum over the 64 lanes of a wave, fixed butterfly order (deterministic)
__device__ __forceinline__ float ht_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// forward of one head for the 16 rows of this workgroup; xaT must hold the head's action part.
// `store`: write activations / outputs to global (false for the duplicate pi head of chain 5).
// A single 1-wide output layer (vf / qf heads) is folded into the last hidden stage: each lane holds
// z[4 rows][its column], multiplies by ow[column] and the wave (= one row group) adds the 64 lanes.
__device__ __forceinline__ void ht_fwd_head(const HtHead& h, int row0, int B, HtLds& s, bool keep_out, bool store = true) {
  const int t = threadIdx.x, cl = t & 63, rg = t >> 6;
  const int HL = h.hid[h.L - 1];
  const bool fuse_out = h.n_out == 1 && h.out_dim == 1;
  float opart[4] = {0.f, 0.f, 0.f, 0.f};   // fused output layer: per-lane partial of sum_n z[r][n] * ow[n]
  HtW pw;
  // kernel of the first staged stage (hidden layer 1, or the first output layer)
  if (h.L > 1) ht_w_fetch(pw, h.w[1], h.hid[0] * h.hid[1]);
  else if (!fuse_out) ht_w_fetch(pw, h.ow[0], HL * h.out_dim);
  // ---- layer 0
  for (int c0 = 0; c0 < h.H0; c0 += 64) {
    const int n = c0 + cl;
    if (n < h.H0) {
      float acc[4];
      const float bn = h.b0[n];
      const float own = (fuse_out && h.L == 1) ? h.ow[0][n] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + 4 * rg + i;
        acc[i] = row < B ? h.u[(long)row * h.ldu + n] : 0.f;
      }
      for (int sp = 1; sp < h.u_split; ++sp)     // split reduction of the layer-0 GEMM: partial sums, in order
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = row0 + 4 * rg + i;
          acc[i] += row < B ? h.u[sp * h.u_stride + (long)row * h.ldu + n] : 0.f;
        }
      for (int a = 0; a < h.n_xa; ++a) {
        const float w = h.w0a[a * h.H0 + n];
        const ht_f4 x = *(const ht_f4*)&s.xaT[a][4 * rg];
        acc[0] = fmaf(x.x, w, acc[0]); acc[1] = fmaf(x.y, w, acc[1]);
        acc[2] = fmaf(x.z, w, acc[2]); acc[3] = fmaf(x.w, w, acc[3]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + 4 * rg + i;
        const float v = fmaxf(acc[i] + bn, 0.f);
        s.zT[0][n][4 * rg + i] = v;
        if (row < B && store && h.z0) h.z0[(long)row * h.H0 + n] = v;
        opart[i] = fmaf(v, own, opart[i]);
      }
    }
  }
  // ---- hidden layers (ht_w_commit's barriers also publish zT of the previous layer)
  for (int l = 1; l < h.L; ++l) {
    const int Hin = h.hid[l - 1], Hout = h.hid[l];
    float(*src)[HT_RB] = s.zT[(l - 1) & 1];
    float(*dst)[HT_RB] = s.zT[l & 1];
    ht_w_commit(pw, h.w[l], Hin, Hout, s.W);
    if (l + 1 < h.L) ht_w_fetch(pw, h.w[l + 1], Hout * h.hid[l + 1]);
    else if (!fuse_out) ht_w_fetch(pw, h.ow[0], HL * h.out_dim);
    for (int c0 = 0; c0 < Hout; c0 += 64) {
      const int n = c0 + cl;
      if (n < Hout) {
        const float bn = h.b[l][n];   // requested before the dot product, consumed after it
        const float own = (fuse_out && l + 1 == h.L) ? h.ow[0][n] : 0.f;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        ht_dot(acc, s.W + n, Hout + 1, &src[0][4 * rg], Hin);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = row0 + 4 * rg + i;
          const float v = fmaxf(acc[i] + bn, 0.f);
          dst[n][4 * rg + i] = v;
          if (row < B && store && h.z[l]) h.z[l][(long)row * Hout + n] = v;
          opart[i] = fmaf(v, own, opart[i]);
        }
      }
    }
  }
  // ---- output layers (out_dim <= 64)
  if (fuse_out) {
    const float ob = h.ob[0][0];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float q = ht_wave_sum(opart[i]) + ob;
      const int row = row0 + 4 * rg + i;
      if (cl == 0 && row < B && store) h.out[0][row] = q;
    }
  } else {
    float(*src)[HT_RB] = s.zT[(h.L - 1) & 1];
    for (int k = 0; k < h.n_out; ++k) {
      ht_w_commit(pw, h.ow[k], HL, h.out_dim, s.W);
      if (k + 1 < h.n_out) ht_w_fetch(pw, h.ow[k + 1], HL * h.out_dim);
      if (cl < h.out_dim) {
        const float bo = h.ob[k][cl];
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        ht_dot(acc, s.W + cl, h.out_dim + 1, &src[0][4 * rg], HL);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = row0 + 4 * rg + i;
          const float v = acc[i] + bo;
          if (keep_out) s.oT[k * h.out_dim + cl][4 * rg + i] = v;
          if (row < B && store) h.out[k][(long)row * (h.ld_out ? h.ld_out : h.out_dim) + cl] = v;
        }
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void ht_load_xa(const HtHead& h, int row0, int B, HtLds& s) {
  const int t = threadIdx.x;
  const int r = t & (HT_RB - 1), row = row0 + r;
  for (int a = t / HT_RB; a < h.n_xa; a += 256 / HT_RB) s.xaT[a][r] = row < B ? h.xa[(long)row * h.ld_xa + a] : 0.f;
  __syncthreads();
}

__global__ __launch_bounds__(256) void heads_fwd_kernel(HeadsFwdArgs a) {
  __shared__ HtLds s;
  const int row0 = blockIdx.x * HT_RB, chain = blockIdx.y, t = threadIdx.x;
  if (chain >= 1 && chain <= 4) {
    const HtHead& h = a.h[chain];
    if (h.n_xa) ht_load_xa(h, row0, a.B, s);
    ht_fwd_head(h, row0, a.B, s, false);
    return;
  }
  // chains 0 and 5: pi head -> sample -> qf1(s, pi) resp. qf2(s, pi).  Both compute the (cheap) pi head;
  // chain 0 owns its global outputs.
  const bool own = chain == 0;
  ht_fwd_head(a.h[0], row0, a.B, s, true, own);
  if (t < HT_RB) {
    const int row = row0 + t;
    float lp = 0.f, en = 0.f;
    for (int j = 0; j < a.A; ++j) {
      const float ep = row < a.B ? a.eps[(long)row * a.A + j] : 0.f;
      const float pj = ht_sample_elem(s.oT[j][t], s.oT[a.A + j][t], ep, lp, en);
      s.xaT[j][t] = pj;
      if (row < a.B && own) a.pi_a[(long)row * a.A + j] = pj;
    }
    if (row < a.B && own) { a.logp[row] = lp; a.ent[row] = en; }
  }
  __syncthreads();
  if (own) ht_fwd_head(a.h[5], row0, a.B, s, false);
  else ht_fwd_head(a.h[6], row0, a.B, s, false);
}

// backward of one head for the 16 rows of this workgroup.  oT must hold the output gradients
// [k*out_dim + o][row].  Writes every g[l]; optionally d xa (to global and to xaT).
// dz_parts != nullptr: instead of output layers, the gradient w.r.t. the last hidden ACTIVATION is given as
// n_parts partial sums [n_parts][B, HL] (added in order, then scaled): a trunk whose consumers (towers) were
// differentiated by other workgroups.
__device__ __forceinline__ void ht_bwd_head(const HtHead& h, int row0, int B, HtLds& s, float* da, int ld_da,
                                            const float* dz_parts = nullptr, int n_parts = 0, long part_stride = 0,
                                            float dz_scale = 1.f) {
  const int t = threadIdx.x, cl = t & 63, rg = t >> 6;
  const int L = h.L;
  HtW pw;   // kernel of the stage after the output layers: last hidden layer, or the action rows (d xa)
  if (L > 1) ht_w_fetch(pw, h.w[L - 1], h.hid[L - 2] * h.hid[L - 1]);
  else if (da) ht_w_fetch(pw, h.w0a, h.n_xa * h.H0);
  // ---- output layers -> g_{L-1}
  {
    const int HL = h.hid[L - 1];
    const float* zl = L == 1 ? h.z0 : h.z[L - 1];
    float* gl = L == 1 ? nullptr : h.g[L - 1];
    const bool one = h.n_out == 1 && h.out_dim == 1;   // vf / qf heads: a rank-1 product, straight from global
    if (!one && !dz_parts)
      for (int k = 0; k < h.n_out; ++k) ht_stage_w_at(h.ow[k], HL, h.out_dim, s.W + k * HL * (h.out_dim + 1));
    for (int c0 = 0; c0 < HL; c0 += 64) {
      const int n = c0 + cl;
      if (n < HL) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (dz_parts) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = row0 + 4 * rg + i;
            float sum = 0.f;
            if (row < B)
              for (int p = 0; p < n_parts; ++p) sum += dz_parts[p * part_stride + (long)row * HL + n];
            acc[i] = sum * dz_scale;
          }
        } else if (one) {
          const float w = h.ow[0][n];
          const ht_f4 d = *(const ht_f4*)&s.oT[0][4 * rg];
          acc[0] = fmaf(d.x, w, 0.f); acc[1] = fmaf(d.y, w, 0.f); acc[2] = fmaf(d.z, w, 0.f); acc[3] = fmaf(d.w, w, 0.f);
        } else
        for (int k = 0; k < h.n_out; ++k) {
          const float* wp = s.W + k * HL * (h.out_dim + 1) + n * (h.out_dim + 1);
          for (int o = 0; o < h.out_dim; ++o) {
            const float w = wp[o];
            const ht_f4 d = *(const ht_f4*)&s.oT[k * h.out_dim + o][4 * rg];
            acc[0] = fmaf(d.x, w, acc[0]); acc[1] = fmaf(d.y, w, acc[1]);
            acc[2] = fmaf(d.z, w, acc[2]); acc[3] = fmaf(d.w, w, acc[3]);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = row0 + 4 * rg + i;
          float v = 0.f;
          if (row < B) {
            v = zl[(long)row * HL + n] > 0.f ? acc[i] : 0.f;
            if (gl) gl[(long)row * HL + n] = v;
            else h.g0[(long)row * h.ldg0 + n] = v;
          }
          s.zT[(L - 1) & 1][n][4 * rg + i] = v;
        }
      }
    }
  }
  __syncthreads();
  // ---- hidden layers: g_{l-1} = mask * (g_l . W_l^T)
  for (int l = L - 1; l >= 1; --l) {
    const int Hout = h.hid[l], Hin = h.hid[l - 1];
    float(*src)[HT_RB] = s.zT[l & 1];
    float(*dst)[HT_RB] = s.zT[(l - 1) & 1];
    const float* zp = l == 1 ? h.z0 : h.z[l - 1];
    ht_w_commit(pw, h.w[l], Hin, Hout, s.W);
    if (l > 1) ht_w_fetch(pw, h.w[l - 1], h.hid[l - 2] * Hin);
    else if (da) ht_w_fetch(pw, h.w0a, h.n_xa * h.H0);
    for (int c0 = 0; c0 < Hin; c0 += 64) {
      const int m = c0 + cl;
      if (m < Hin) {
        float zm[4];   // ReLU mask operands: requested before the dot product
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = row0 + 4 * rg + i;
          zm[i] = row < B ? zp[(long)row * Hin + m] : 0.f;
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        ht_dot(acc, s.W + m * (Hout + 1), 1, &src[0][4 * rg], Hout);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = row0 + 4 * rg + i;
          float v = 0.f;
          if (row < B) {
            v = zm[i] > 0.f ? acc[i] : 0.f;
            if (l == 1) h.g0[(long)row * h.ldg0 + m] = v;
            else h.g[l - 1][(long)row * Hin + m] = v;
          }
          dst[m][4 * rg + i] = v;
        }
      }
    }
    __syncthreads();
  }
  // ---- d xa = g_0 . w0a^T
  if (da) {
    ht_w_commit(pw, h.w0a, h.n_xa, h.H0, s.W);
    if (cl < h.n_xa) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      ht_dot(acc, s.W + cl * (h.H0 + 1), 1, &s.zT[0][0][4 * rg], h.H0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + 4 * rg + i;
        s.xaT[cl][4 * rg + i] = acc[i];
        if (row < B) da[(long)row * ld_da + cl] = acc[i];
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void heads_bwd_kernel(HeadsBwdArgs a) {
  __shared__ HtLds s;
  const int row0 = blockIdx.x * HT_RB, chain = blockIdx.y, t = threadIdx.x;
  if (chain != 0) {
    const HtHead& h = a.h[chain];
    if (t < HT_RB) {
      const int row = row0 + t;
      float d = 0.f;
      if (row < a.B) {
        d = ht_dout(a, chain, row, expf(a.log_ent_coef[0]));
        a.d_out[chain][(long)row * a.ld_d] = d;
      }
      s.oT[0][t] = d;
    }
    __syncthreads();
    ht_bwd_head(h, row0, a.B, s, nullptr, 0);
    return;
  }
  // chain 0: qf1(s, pi) backward -> d a_pi -> sample backward -> pi head backward
  const HtHead& q = a.h[4];
  if (t < HT_RB) {
    const int row = row0 + t;
    float d = 0.f;
    if (row < a.B) {
      d = ht_dout(a, 4, row, 0.f);
      a.d_out[4][(long)row * a.ld_d] = d;
    }
    s.oT[0][t] = d;
  }
  __syncthreads();
  ht_bwd_head(q, row0, a.B, s, a.da_pi, a.A);
  if (t < HT_RB) {
    const int row = row0 + t;
    const float alpha_over_b = expf(a.log_ent_coef[0]) / (float)a.B;
    const int rc = row < a.B ? row : 0;
    for (int j = 0; j < a.A; ++j) {
      float m, d;
      ht_sample_bwd_elem(a.ls_raw[(long)rc * a.A + j], a.eps[(long)rc * a.A + j], a.pi_a[(long)rc * a.A + j],
                         s.xaT[j][t], alpha_over_b, m, d);
      if (row >= a.B) { m = 0.f; d = 0.f; }
      s.oT[j][t] = m;
      s.oT[a.A + j][t] = d;
      if (row < a.B) { a.dmu[(long)row * a.ld_dm + j] = m; a.dls[(long)row * a.ld_dm + j] = d; }
    }
  }
  __syncthreads();
  ht_bwd_head(a.h[0], row0, a.B, s, nullptr, 0);
}

#endif  // GRL_HOSTEMU
#endif  // GRL_HEADS_TYPES_ONLY

}  // namespace grl
