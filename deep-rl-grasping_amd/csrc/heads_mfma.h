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
//     (4 waves = 4 column blocks of 16): weights go global -> registers (B operand, no LDS staging), the
//     activations of the 16 rows sit in LDS row-major with a +4 pad (A operand: 16-byte reads, no conflicts);
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
};

#ifdef GRL_HOSTEMU
// ------------------------------------------------------------------------------------------------
// TEST-ONLY sequential form (see hostemu.h): one "thread" walks the rows of a block for one type.
// zs: [L][HT_MAXW] activations of this row, outs: [n_out * out_dim]
inline void hm_ref_fwd(const HtHead& h, int row, const float* xa_row, bool store, float* zs, float* outs) {
  for (int n = 0; n < h.H0; ++n) {
    float acc = h.u[(long)row * h.ldu + n];
    for (int sp = 1; sp < h.u_split; ++sp) acc += h.u[sp * h.u_stride + (long)row * h.ldu + n];
    for (int a = 0; a < h.n_xa; ++a) acc = fmaf(xa_row[a], h.w0a[a * h.H0 + n], acc);
    zs[n] = fmaxf(acc + h.b0[n], 0.f);
    if (store && h.z0) h.z0[(long)row * h.H0 + n] = zs[n];
  }
  for (int l = 1; l < h.L; ++l) {
    float* zi = zs + (l - 1) * HT_MAXW;
    float* zo = zs + l * HT_MAXW;
    for (int n = 0; n < h.hid[l]; ++n) {
      float acc = 0.f;
      for (int k = 0; k < h.hid[l - 1]; ++k) acc = fmaf(zi[k], h.w[l][k * h.hid[l] + n], acc);
      zo[n] = fmaxf(acc + h.b[l][n], 0.f);
      if (store && h.z[l]) h.z[l][(long)row * h.hid[l] + n] = zo[n];
    }
  }
  const int HL = h.hid[h.L - 1];
  const float* zl = zs + (h.L - 1) * HT_MAXW;
  for (int k = 0; k < h.n_out; ++k)
    for (int o = 0; o < h.out_dim; ++o) {
      float acc = 0.f;
      for (int n = 0; n < HL; ++n) acc = fmaf(zl[n], h.ow[k][n * h.out_dim + o], acc);
      outs[k * h.out_dim + o] = acc + h.ob[k][o];
      if (store) h.out[k][(long)row * (h.ld_out ? h.ld_out : h.out_dim) + o] = outs[k * h.out_dim + o];
    }
}
// dvals: n_out * out_dim output gradients; zs as left by hm_ref_fwd; writes g (when the head has g pointers), da_row
inline void hm_ref_bwd(const HtHead& h, int row, const float* zs, const float* dvals, float* da_row) {
  float gin[HT_MAXW], gout[HT_MAXW];
  const int HL = h.hid[h.L - 1];
  const float* zl = zs + (h.L - 1) * HT_MAXW;
  for (int n = 0; n < HL; ++n) {
    float acc = 0.f;
    for (int k = 0; k < h.n_out; ++k)
      for (int o = 0; o < h.out_dim; ++o) acc = fmaf(dvals[k * h.out_dim + o], h.ow[k][n * h.out_dim + o], acc);
    gin[n] = zl[n] > 0.f ? acc : 0.f;
  }
  for (int l = h.L - 1; l >= 1; --l) {
    if (h.g[l]) for (int n = 0; n < h.hid[l]; ++n) h.g[l][(long)row * h.hid[l] + n] = gin[n];
    const float* zp = zs + (l - 1) * HT_MAXW;
    for (int m = 0; m < h.hid[l - 1]; ++m) {
      float acc = 0.f;
      for (int n = 0; n < h.hid[l]; ++n) acc = fmaf(gin[n], h.w[l][m * h.hid[l] + n], acc);
      gout[m] = zp[m] > 0.f ? acc : 0.f;
    }
    for (int m = 0; m < h.hid[l - 1]; ++m) gin[m] = gout[m];
  }
  if (h.g0) for (int n = 0; n < h.H0; ++n) h.g0[(long)row * h.ldg0 + n] = gin[n];
  if (da_row)
    for (int a = 0; a < h.n_xa; ++a) {
      float acc = 0.f;
      for (int n = 0; n < h.H0; ++n) acc = fmaf(gin[n], h.w0a[a * h.H0 + n], acc);
      da_row[a] = acc;
    }
}

template <int W>
void heads_fused_kernel(const HeadsFusedArgs* ap) {
  if (threadIdx.x != 0) return;
  const HeadsFusedArgs& a = *ap;
  const int type = blockIdx.y;
  const float alpha = expf(a.log_ent_coef[0]);
  const float invB = 1.f / (float)a.B;
  static thread_local float zs_a[GRL_MAX_LAYERS * HT_MAXW], zs_b[GRL_MAX_LAYERS * HT_MAXW];
  for (int row = blockIdx.x * HT_RB; row < std::min(a.B, (int)blockIdx.x * HT_RB + HT_RB); ++row) {
    float outs[2 * HT_MAXA], o1[2], o2[2], dv[2 * HT_MAXA];
    if (type <= 1) {
      const bool own = type == 0;
      hm_ref_fwd(a.h[0], row, nullptr, own, zs_a, outs);
      float pi[HT_MAXA], lp, en;
      ht_sample_row(outs, outs + a.A, a.eps + (long)row * a.A, a.A, pi, &lp, &en);
      if (own) {
        for (int j = 0; j < a.A; ++j) a.pi_a[(long)row * a.A + j] = pi[j];
        a.logp[row] = lp; a.ent[row] = en;
      }
      hm_ref_fwd(a.h[5], row, pi, false, zs_b, o1);
      if (own) {
        a.h[5].out[0][row] = o1[0];
        dv[0] = -invB;
        a.d_out[4][(long)row * a.ld_d] = dv[0];
        float da[HT_MAXA];
        HtHead q = a.h[5];
        q.g0 = nullptr;
        for (int l = 0; l < GRL_MAX_LAYERS; ++l) q.g[l] = nullptr;
        hm_ref_bwd(q, row, zs_b, dv, da);
        for (int j = 0; j < a.A; ++j) a.da_pi[(long)row * a.A + j] = da[j];
        ht_sample_bwd_row(outs + a.A, a.eps + (long)row * a.A, pi, da, a.A, alpha * invB, a.dmu + (long)row * a.ld_dm,
                          a.dls + (long)row * a.ld_dm);
        for (int j = 0; j < a.A; ++j) { dv[j] = a.dmu[(long)row * a.ld_dm + j]; dv[a.A + j] = a.dls[(long)row * a.ld_dm + j]; }
        hm_ref_bwd(a.h[0], row, zs_a, dv, nullptr);
      } else {
        hm_ref_fwd(a.h[6], row, pi, false, zs_b, o2);
        a.h[6].out[0][row] = o2[0];
        float v[2];
        hm_ref_fwd(a.h[1], row, nullptr, true, zs_a, v);
        dv[0] = (v[0] - (fminf(o1[0], o2[0]) - alpha * lp)) * invB;
        a.d_out[1][(long)row * a.ld_d] = dv[0];
        hm_ref_bwd(a.h[1], row, zs_a, dv, nullptr);
      }
    } else {
      float vt[2], q[2];
      hm_ref_fwd(a.h[4], row, nullptr, type == 2, zs_a, vt);
      const HtHead& h = a.h[type];
      hm_ref_fwd(h, row, h.xa + (long)row * h.ld_xa, true, zs_b, q);
      const float qb = a.rew[row] + (1.f - a.done[row]) * a.gamma * vt[0];
      dv[0] = (q[0] - qb) * invB;
      a.d_out[type][(long)row * a.ld_d] = dv[0];
      hm_ref_bwd(h, row, zs_b, dv, nullptr);
    }
  }
}

#else  // ------------------------------------------------------------------------------------ device

typedef float hm_f4 __attribute__((ext_vector_type(4)));

// B operand of a stage: element (k, n) of a [K, N] matrix served from one or two weight tensors; zero outside.
struct HmB {
  const float* p0; const float* p1;
  int K, N;        // logical extents
  int sk, sn;      // strides of k and n inside a part
  int split;       // first index served by p1 -- along n (on_k == 0) or along k (on_k == 1); >= extent: one part
  int on_k;
};

template <int W>
struct HmLds {
  float z[2][HT_RB][W + 4];       // activations / gradients entering the next stage, row-major
  float o[HT_RB][W + 4];          // head outputs (mu | log_std), output gradients, feature of a rank-K product
  float pi[HT_RB][HT_MAXA];       // sampled action of the rows
  float ls[HT_RB][HT_MAXA];       // raw log_std
  float da[HT_RB][HT_MAXA];       // d loss / d action
  float sv[8][HT_RB];             // per-row scalars: 0 qf1_pi, 1 qf2_pi, 2 logp, 3 v / q, 4 v_tgt, 5 d
};

// The argument block lives in device memory (descriptor upload of the plan): a by-value struct indexed with the
// run-time chain type would be copied to scratch by the compiler.
template <int W>
__global__ __launch_bounds__(256) void heads_fused_kernel(const HeadsFusedArgs* __restrict__ ap) {
  constexpr int K4 = W / 4, NB = W / 64, LD = W + 4;
  const HeadsFusedArgs& a = *ap;
  __shared__ __attribute__((aligned(16))) HmLds<W> s;
  const int t = threadIdx.x, w = t >> 6, l = t & 63, c = l & 15, q = l >> 4;
  const int row0 = blockIdx.x * HT_RB, type = blockIdx.y, B = a.B, A = a.A;
  const float invB = 1.f / (float)B;

  // ---------------------------------------------------------------- primitives
  // column owned by this lane in column block b of a stage output (MFMA result layout: rows 4q .. 4q+3)
  auto col_of = [&](int b) { return 16 * (w + 4 * b) + c; };
  auto load_b = [&](float (&bw)[NB][K4], const HmB& d) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int n = col_of(b);
#pragma unroll
      for (int sI = 0; sI < K4; ++sI) {
        const int k = K4 * q + sI;
        const float* p = d.p0;
        int kk = k, nn = n;
        if (d.on_k) { if (k >= d.split) { p = d.p1; kk = k - d.split; } }
        else { if (n >= d.split) { p = d.p1; nn = n - d.split; } }
        const bool ok = k < d.K && n < d.N;
        bw[b][sI] = ok ? p[(long)kk * d.sk + (long)nn * d.sn] : 0.f;
      }
    }
  };
  // acc[b] += zin[16 rows][W] . bw   (A operand: lane (row c, quarter q) holds k = K4 q .. K4 q + K4 - 1)
  auto mma = [&](hm_f4 (&acc)[NB], const float (*zin)[LD], const float (&bw)[NB][K4]) {
    float av[K4];
#pragma unroll
    for (int j = 0; j < K4 / 4; ++j) {
      const hm_f4 v = *(const hm_f4*)&zin[c][K4 * q + 4 * j];
      av[4 * j] = v.x; av[4 * j + 1] = v.y; av[4 * j + 2] = v.z; av[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int sI = 0; sI < K4; ++sI)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sI], bw[b][sI], acc[b], 0, 0, 0);
  };
  auto zero_acc = [&](hm_f4 (&acc)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = hm_f4{0.f, 0.f, 0.f, 0.f};
  };

  int cur = 0;   // s.z[cur] holds the input of the next MFMA stage

  // forward of one head.  zsv[l][b][i]: activations kept in the result layout (row 4q+i, column col_of(b)).
  // xa: the action part comes from s.pi (use_pi) or from global h.xa.  Outputs land in s.o[row][k*out_dim+o].
  auto fwd_head = [&](const HtHead& h, bool use_pi, bool store, float (&zsv)[GRL_MAX_LAYERS][NB][4]) {
    float bw[NB][K4];
    if (h.L > 1) load_b(bw, HmB{h.w[1], nullptr, h.hid[0], h.hid[1], h.hid[1], 1, INT_MAX, 0});
    else load_b(bw, HmB{h.ow[0], h.n_out > 1 ? h.ow[1] : nullptr, h.hid[0], h.n_out * h.out_dim, h.out_dim, 1,
                        h.n_out > 1 ? h.out_dim : INT_MAX, 0});
    // ---- layer 0 (element-wise on the partial sums of the heads_l0 GEMM)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int n = col_of(b);
      const bool nok = n < h.H0;
      const float bn = nok ? h.b0[n] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * q + i, row = row0 + r;
        float v = 0.f;
        if (nok && row < B) {
          float acc = h.u[(long)row * h.ldu + n];
          for (int sp = 1; sp < h.u_split; ++sp) acc += h.u[sp * h.u_stride + (long)row * h.ldu + n];
          for (int x = 0; x < h.n_xa; ++x)
            acc = fmaf(use_pi ? s.pi[r][x] : h.xa[(long)row * h.ld_xa + x], h.w0a[x * h.H0 + n], acc);
          v = fmaxf(acc + bn, 0.f);
          if (store && h.z0) h.z0[(long)row * h.H0 + n] = v;
        }
        zsv[0][b][i] = v;
        s.z[cur][r][n] = v;
      }
    }
    __syncthreads();
    // ---- hidden layers
#pragma unroll
    for (int ly = 1; ly < GRL_MAX_LAYERS; ++ly) {
      if (ly < h.L) {
        const int Hout = h.hid[ly];
        hm_f4 acc[NB];
        zero_acc(acc);
        float bias[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) bias[b] = col_of(b) < Hout ? h.b[ly][col_of(b)] : 0.f;
        mma(acc, s.z[cur], bw);
        if (ly + 1 < h.L) load_b(bw, HmB{h.w[ly + 1], nullptr, Hout, h.hid[ly + 1], h.hid[ly + 1], 1, INT_MAX, 0});
        else load_b(bw, HmB{h.ow[0], h.n_out > 1 ? h.ow[1] : nullptr, Hout, h.n_out * h.out_dim, h.out_dim, 1,
                            h.n_out > 1 ? h.out_dim : INT_MAX, 0});
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int n = col_of(b);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * q + i, row = row0 + r;
            const float v = (n < Hout && row < B) ? fmaxf(acc[b][i] + bias[b], 0.f) : 0.f;
            zsv[ly][b][i] = v;
            s.z[cur ^ 1][r][n] = v;
            if (store && h.z[ly] && n < Hout && row < B) h.z[ly][(long)row * Hout + n] = v;
          }
        }
        cur ^= 1;
        __syncthreads();
      }
    }
    // ---- output layer(s): [mu | log_std] or one value
    {
      const int NO = h.n_out * h.out_dim;
      hm_f4 acc[NB];
      zero_acc(acc);
      float bias[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int n = col_of(b);
        bias[b] = n < NO ? (n < h.out_dim ? h.ob[0][n] : h.ob[1][n - h.out_dim]) : 0.f;
      }
      mma(acc, s.z[cur], bw);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int n = col_of(b);
        if (n < NO) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * q + i, row = row0 + r;
            const float v = row < B ? acc[b][i] + bias[b] : 0.f;
            s.o[r][n] = v;
            if (store && row < B) {
              const int k = n < h.out_dim ? 0 : 1, o = n - k * h.out_dim;
              h.out[k][(long)row * (h.ld_out ? h.ld_out : h.out_dim) + o] = v;
            }
          }
        }
      }
    }
    __syncthreads();
  };

  // backward of one head.  rank1: the head has one scalar output whose gradient per row is s.sv[5][row];
  // otherwise the output gradients are s.o[row][k*out_dim + o] (zero beyond).  Writes g (heads with g pointers),
  // and with want_da the gradient w.r.t. the action part into s.da (+ global da when given).
  auto bwd_head = [&](const HtHead& h, const float (&zsv)[GRL_MAX_LAYERS][NB][4], bool rank1, bool want_da, float* da_glob,
                      bool store_g) {
    const int L = h.L;
    const int HL = h.hid[L - 1];
    float bw[NB][K4];
    // ---- output layer(s) -> gradient of the last hidden pre-activation
    {
      hm_f4 acc[NB];
      zero_acc(acc);
      if (rank1) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int n = col_of(b);
          const float wn = n < HL ? h.ow[0][n] : 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[b][i] = s.sv[5][4 * q + i] * wn;
        }
      } else {
        // element (k, n) = ow[k / out_dim][n * out_dim + k % out_dim]
        load_b(bw, HmB{h.ow[0], h.n_out > 1 ? h.ow[1] : nullptr, h.n_out * h.out_dim, HL, 1, h.out_dim,
                       h.n_out > 1 ? h.out_dim : INT_MAX, 1});
        mma(acc, s.o, bw);
      }
      if (L > 1) load_b(bw, HmB{h.w[L - 1], nullptr, HL, h.hid[L - 2], 1, HL, INT_MAX, 0});
      else if (want_da) load_b(bw, HmB{h.w0a, nullptr, h.H0, h.n_xa, 1, h.H0, INT_MAX, 0});
      float* gl = !store_g ? nullptr : (L == 1 ? h.g0 : h.g[L - 1]);
      const int ldgl = L == 1 ? h.ldg0 : HL;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int n = col_of(b);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * q + i, row = row0 + r;
          float zl = 0.f;                         // activation of the last hidden layer (static register indices)
#pragma unroll
          for (int ly = 0; ly < GRL_MAX_LAYERS; ++ly) zl = (ly == L - 1) ? zsv[ly][b][i] : zl;
          const float v = (n < HL && row < B && zl > 0.f) ? acc[b][i] : 0.f;
          s.z[cur][r][n] = v;
          if (gl && n < HL && row < B) gl[(long)row * ldgl + n] = v;
        }
      }
    }
    __syncthreads();
    // ---- hidden layers: g_{l-1} = mask * (g_l . W_l^T)
#pragma unroll
    for (int ly = GRL_MAX_LAYERS - 1; ly >= 1; --ly) {
      if (ly < L) {
        const int Hin = h.hid[ly - 1];
        hm_f4 acc[NB];
        zero_acc(acc);
        mma(acc, s.z[cur], bw);
        if (ly > 1) load_b(bw, HmB{h.w[ly - 1], nullptr, Hin, h.hid[ly - 2], 1, Hin, INT_MAX, 0});
        else if (want_da) load_b(bw, HmB{h.w0a, nullptr, h.H0, h.n_xa, 1, h.H0, INT_MAX, 0});
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int m = col_of(b);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * q + i, row = row0 + r;
            const float v = (m < Hin && row < B && zsv[ly - 1][b][i] > 0.f) ? acc[b][i] : 0.f;
            s.z[cur ^ 1][r][m] = v;
            if (store_g && m < Hin && row < B) {
              if (ly == 1) h.g0[(long)row * h.ldg0 + m] = v;
              else h.g[ly - 1][(long)row * Hin + m] = v;
            }
          }
        }
        cur ^= 1;
        __syncthreads();
      }
    }
    // ---- d xa = g_0 . w0a^T
    if (want_da) {
      hm_f4 acc[NB];
      zero_acc(acc);
      mma(acc, s.z[cur], bw);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int x = col_of(b);
        if (x < h.n_xa) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * q + i, row = row0 + r;
            s.da[r][x] = acc[b][i];
            if (da_glob && row < B) da_glob[(long)row * h.n_xa + x] = acc[b][i];
          }
        }
      }
      __syncthreads();
    }
  };

  // squashed-Gaussian sample of the rows from s.o = [mu | log_std]; one lane per row (A is small)
  auto sample = [&](bool store) {
    if (t < HT_RB) {
      const int row = row0 + t;
      float lp = 0.f, en = 0.f;
      for (int j = 0; j < A; ++j) {
        const float ep = row < B ? a.eps[(long)row * A + j] : 0.f;
        const float lr = s.o[t][A + j];
        const float pj = ht_sample_elem(s.o[t][j], lr, ep, lp, en);
        s.pi[t][j] = pj;
        s.ls[t][j] = lr;
        if (row < B && store) a.pi_a[(long)row * A + j] = pj;
      }
      s.sv[2][t] = lp;
      if (row < B && store) { a.logp[row] = lp; a.ent[row] = en; }
    }
    __syncthreads();
  };

  // every A operand is read over the padded width: start from finite (zero) LDS
  for (int x = t; x < (int)(sizeof(s) / 4); x += 256) ((float*)&s)[x] = 0.f;
  __syncthreads();

  float zsA[GRL_MAX_LAYERS][NB][4], zsB[GRL_MAX_LAYERS][NB][4];
  if (type <= 1) {
    const bool own = type == 0;
    fwd_head(a.h[0], false, own, zsA);          // pi: s.o = [mu | log_std]
    sample(own);
    fwd_head(a.h[5], true, false, zsB);         // qf1(s, pi): s.o[r][0]
    if (t < HT_RB) {
      s.sv[0][t] = s.o[t][0];
      if (own && row0 + t < B) a.h[5].out[0][row0 + t] = s.o[t][0];
    }
    __syncthreads();
    if (own) {
      if (t < HT_RB) {
        s.sv[5][t] = row0 + t < B ? -invB : 0.f;
        if (row0 + t < B) a.d_out[4][(long)(row0 + t) * a.ld_d] = -invB;
      }
      __syncthreads();
      bwd_head(a.h[5], zsB, true, true, a.da_pi, false);   // gradients of qf1's own weights are not wanted here (policy loss)
      // sample backward: one lane per (row, j) pair
      {
        const float alpha_over_b = expf(a.log_ent_coef[0]) * invB;
        for (int e = t; e < HT_RB * W; e += 256) {     // refresh s.o as the next A operand: zero beyond 2A
          const int r = e / W, k = e - r * W;
          const int row = row0 + r;
          float v = 0.f;
          if (k < 2 * A && row < B) {
            const int j = k < A ? k : k - A;
            float m, d;
            ht_sample_bwd_elem(s.ls[r][j], a.eps[(long)row * A + j], s.pi[r][j], s.da[r][j], alpha_over_b, m, d);
            v = k < A ? m : d;
            (k < A ? a.dmu : a.dls)[(long)row * a.ld_dm + j] = v;
          }
          s.o[r][k] = v;
        }
        __syncthreads();
      }
      bwd_head(a.h[0], zsA, false, false, nullptr, true);
    } else {
      fwd_head(a.h[6], true, false, zsB);       // qf2(s, pi)
      if (t < HT_RB) {
        s.sv[1][t] = s.o[t][0];
        if (row0 + t < B) a.h[6].out[0][row0 + t] = s.o[t][0];
      }
      __syncthreads();
      fwd_head(a.h[1], false, true, zsA);       // vf
      if (t < HT_RB) {
        const int row = row0 + t;
        const float alpha = expf(a.log_ent_coef[0]);
        const float vb = fminf(s.sv[0][t], s.sv[1][t]) - alpha * s.sv[2][t];
        const float d = row < B ? (s.o[t][0] - vb) * invB : 0.f;
        s.sv[5][t] = d;
        if (row < B) a.d_out[1][(long)row * a.ld_d] = d;
      }
      __syncthreads();
      bwd_head(a.h[1], zsA, true, false, nullptr, true);
    }
  } else {
    fwd_head(a.h[4], false, type == 2, zsA);    // target vf of next_obs
    if (t < HT_RB) s.sv[4][t] = s.o[t][0];
    __syncthreads();
    const HtHead& h = a.h[type];
    fwd_head(h, false, true, zsB);              // qf(s, a) on the minibatch actions
    if (t < HT_RB) {
      const int row = row0 + t;
      float d = 0.f;
      if (row < B) {
        const float qb = a.rew[row] + (1.f - a.done[row]) * a.gamma * s.sv[4][t];
        d = (s.o[t][0] - qb) * invB;
        a.d_out[type][(long)row * a.ld_d] = d;
      }
      s.sv[5][t] = d;
    }
    __syncthreads();
    bwd_head(h, zsB, true, false, nullptr, true);
  }
}

#endif  // GRL_HOSTEMU

}  // namespace grl
