// act_mfma.h -- the policy head of the ACT path (grl_act: model.predict, /root/reference/manipulation_main/utils.py:71,
// training/base_callbacks.py:84-88) on the f32 matrix cores, with the stage primitives of q_mfma.h.
//
// A kernel trace of grl_act for 16 environments (scripts/act_trace.sh) put 65 of its 77 us on the GPU, 33 of them in two
// launches: the dense layer of the extractor (M = 16 rows, K = 1024: ONE tile walking 32 slabs, 13.6 us) and the VALU head
// kernel (K = 517 per thread from L2, 19.1 us).  Here
//   * the dense layer runs as a split-K GEMM (plan_sac.inl: partial sums [split][rows, 512], 4 x shorter chains on 4 x the
//     tiles) and THIS kernel adds the partial sums, the bias and the ReLU while it stages its input rows into LDS;
//   * layer 0 (K = 512 + n_direct) is nine 64-deep MFMA stages whose B operands are all requested up front, the hidden and
//     output layers follow with one more round trip; tanh (+ sampling) closes the kernel.
// One workgroup per 16 observations.  Shapes outside (hidden widths > 64, K0 > 576, more than 16 action dimensions) keep
// act_heads_kernel (elem_kernels.h).  The k-order of a stage is the MFMA's: actions agree with the oracle / the golden
// actions to float32 rounding, not bit for bit with the VALU kernel.
#pragma once
#include "q_mfma.h"

namespace grl {

enum { AM_MAXK = 576, AM_CH = AM_MAXK / QM_W, AM_LDX = AM_MAXK + 4,
       AM_MAXP = 4,                 // partial sums of the dense layer this kernel adds up
       AM_MAXSUM = 512, AM_SE = HT_RB * (AM_MAXSUM / 4) / 256 };   // summed columns; 16-byte pieces of them per thread

static inline bool am_shape_ok(int K0, int L, const int* hid, int A) {
  bool ok = K0 >= 1 && K0 <= AM_MAXK && L >= 1 && L <= GRL_MAX_LAYERS && A >= 1 && A <= 16;   // (a split dense layer: <= AM_MAXP parts, <= AM_MAXSUM columns)
  for (int l = 0; l < L; ++l) ok = ok && hid[l] >= 1 && hid[l] <= QM_W;
  return ok;
}

#ifndef GRL_HEADS_TYPES_ONLY
#ifndef GRL_HOSTEMU

struct __attribute__((aligned(16))) AmLds {
  float x[HT_RB][AM_LDX];      // the rows' layer-0 input, zero beyond K0
  float z[2][HT_RB][QM_LD];    // hidden activations
};

// acc + sum_k row[k] * B(k, col) over one 64-deep chunk; `xrow` points at the lane's row, first column of the chunk
__device__ __forceinline__ qm_f4 am_mma(const float* xrow, const float (&bw)[QM_KS], int q, qm_f4 acc) {
  float av[QM_KS];
#pragma unroll
  for (int j = 0; j < QM_KS / 4; ++j) {
    const qm_f4 v = *(const qm_f4*)(xrow + QM_KS * q + 4 * j);
    av[4 * j] = v.x; av[4 * j + 1] = v.y; av[4 * j + 2] = v.z; av[4 * j + 3] = v.w;
  }
#pragma unroll
  for (int s = 0; s < QM_KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw[s], acc, 0, 0, 0);
  return acc;
}

__global__ __launch_bounds__(256) void act_heads_mfma_kernel(ActHeadsArgs a) {
  __shared__ AmLds s;
  const int t = threadIdx.x, w = t >> 6, l = t & 63, c = l & 15, q = l >> 4;
  const int n = 16 * w + c, row0 = blockIdx.x * HT_RB, rows = a.rows, K0 = a.K0, L = a.L, H0 = a.hid[0];
  // ---- the rows' input.  Columns [0, n_sum): ReLU(sum of the dense layer's partial sums + bias) -- every 16-byte piece of
  //      every partial sum requested before the first is used (a loop would wait for each); columns [n_sum, K0): as the
  //      ingest launch left them in `x`
  const int n_sum = a.n_parts > 0 ? a.n_sum : 0, qpr = n_sum >> 2;      // (n_sum: a multiple of 4, <= AM_MAXSUM)
  qm_f4 pv[AM_SE][AM_MAXP], pb[AM_SE];
#pragma unroll
  for (int e = 0; e < AM_SE; ++e) {
    const int qd = t + 256 * e, r = qpr ? qd / qpr : 0, k = 4 * (qd - r * qpr), row = row0 + r;
    const bool ok = qd < HT_RB * qpr && row < rows;
    pb[e] = ok ? *(const qm_f4*)(a.x_bias + k) : qm_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < AM_MAXP; ++p)
      pv[e][p] = (ok && p < a.n_parts) ? *(const qm_f4*)(a.x_parts + p * a.part_stride + (long)row * a.ld_parts + k) : qm_f4{0.f, 0.f, 0.f, 0.f};
  }
  for (int i = t; i < (int)(sizeof(AmLds) / 4); i += 256) ((float*)&s)[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < AM_SE; ++e) {
    const int qd = t + 256 * e, r = qpr ? qd / qpr : 0, k = 4 * (qd - r * qpr);
    if (qd < HT_RB * qpr) {
      qm_f4 v = pv[e][0];
#pragma unroll
      for (int p = 1; p < AM_MAXP; ++p) v += pv[e][p];          // (zeros beyond n_parts)
      v += pb[e];
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      *(qm_f4*)&s.x[r][k] = v;
    }
  }
  for (int i = t; i < HT_RB * (K0 - n_sum); i += 256) {
    const int r = i / (K0 - n_sum), k = n_sum + (i - r * (K0 - n_sum)), row = row0 + r;
    s.x[r][k] = row < rows ? a.x[(long)row * a.ldx + k] : 0.f;
  }
  // ---- layer-0 operands: every 64-deep chunk of the kernel [K0, H0], requested at once (the staging registers are free)
  float bw0[AM_CH][QM_KS];
#pragma unroll
  for (int ch = 0; ch < AM_CH; ++ch)
    if (QM_W * ch < K0) qm_load_b(bw0[ch], a.w[0] + (long)ch * QM_W * H0, K0 - QM_W * ch, H0, H0, 1, n, q);
  const float b0 = n < H0 ? a.b[0][n] : 0.f;
  // ---- the later layers' operands travel with them
  float bw[GRL_MAX_LAYERS][QM_KS], bias[GRL_MAX_LAYERS], bo[2][QM_KS], ob[2];
#pragma unroll
  for (int li = 1; li < GRL_MAX_LAYERS; ++li)
    if (li < L) {
      qm_load_b(bw[li], a.w[li], a.hid[li - 1], a.hid[li], a.hid[li], 1, n, q);
      bias[li] = n < a.hid[li] ? a.b[li][n] : 0.f;
    }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    qm_load_b(bo[k], a.ow[k], a.hid[L - 1], a.A, a.A, 1, n, q);
    ob[k] = n < a.A ? a.ob[k][n] : 0.f;
  }
  __syncthreads();
  // ---- layer 0: three accumulators in turn (a chain of 144 dependent MFMAs would wait for each result), added in order
  qm_f4 acc3[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int ch = 0; ch < AM_CH; ++ch)
    if (QM_W * ch < K0) acc3[ch % 3] = am_mma(&s.x[c][QM_W * ch], bw0[ch], q, acc3[ch % 3]);
  const qm_f4 acc = (acc3[0] + acc3[1]) + acc3[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) s.z[0][4 * q + i][n] = n < H0 ? fmaxf(acc[i] + b0, 0.f) : 0.f;
  __syncthreads();
#pragma unroll
  for (int li = 1; li < GRL_MAX_LAYERS; ++li)
    if (li < L) {
      const qm_f4 h = qm_mma(s.z[(li - 1) & 1], bw[li], c, q);
#pragma unroll
      for (int i = 0; i < 4; ++i) s.z[li & 1][4 * q + i][n] = n < a.hid[li] ? fmaxf(h[i] + bias[li], 0.f) : 0.f;
      __syncthreads();
    }
  // ---- mu, log_std, the action
  const qm_f4 mu4 = qm_mma(s.z[(L - 1) & 1], bo[0], c, q), ls4 = qm_mma(s.z[(L - 1) & 1], bo[1], c, q);
  if (n < a.A) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + 4 * q + i;
      if (row < rows) {
        const int j = row * a.A + n;
        float u = mu4[i] + ob[0];
        const float lsr = ls4[i] + ob[1];
        a.mu[j] = u;
        a.ls[j] = lsr;
        if (!a.deterministic) u += expf(fminf(fmaxf(lsr, GRL_LOG_STD_MIN), GRL_LOG_STD_MAX)) * a.eps[j];
        a.out[j] = tanhf(u);
      }
    }
  }
  // ---- tell the host: the actions sit in page-locked host memory already, so the call can return as soon as every thread's
  // stores are out -- a system-scope release per thread, a barrier, ONE increment per workgroup -- instead of waiting for the
  // stream's completion signal behind the kernel's end (measured same-box: grl_act on 16 observed observations 53.9 -> 44.6 us)
  if (a.done) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

#endif  // GRL_HOSTEMU
#endif  // GRL_HEADS_TYPES_ONLY

}  // namespace grl
