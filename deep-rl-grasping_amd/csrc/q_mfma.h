// q_mfma.h -- the DQN / BDQ networks' row-local chains on the f32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32):
// the kernels of q_kernels.h (stable-baselines `deepq` dueling towers, /root/reference/manipulation_main/training/
// sb_helper.py:159-165, and the branching fork, :210-224) with every layer stage as 16 MFMAs
// per wave instead of a VALU dot-product chain over LDS (~2.5 us per stage there; the chains are 4-5 stages long at
// batch 32-64, where nothing but latency counts).
//
// Same launches, same descriptors (QFusedArgs / HtHead), same tensors written: only the stage arithmetic moves.
//   * a workgroup owns 16 batch rows -- the M of a 16x16x4 MFMA; wave w produces output columns 16 w .. 16 w + 15, so a
//     stage covers widths up to 64 (wider networks stay on q_kernels.h: plan_q.inl decides);
//   * A operand: the 16 rows' activations / gradients in LDS, row-major, zero-padded to 64 (+4 floats: 16-byte reads
//     without bank conflicts); lane (row c, quarter q) holds k = 16 q .. 16 q + 15;
//   * B operand: the layer kernel straight from global memory into registers (transposed for the backward stages by the
//     strides alone), out-of-range elements served as zeros by the buffer descriptor -- and the operands of EVERY stage
//     of the chain are requested before the first stage runs: one memory round trip per workgroup, not one per stage.
// The k-order of a stage is the MFMA's; q_kernels.h and the per-layer GEMM path (GRL_TUNE fused_q=0) sum k sequentially.
// All three are checked against the oracle with the same tolerances (tests/test_gpu_q_parity.py).
#pragma once
#include "q_kernels.h"
#include "igemm2.h"

namespace grl {

enum { QM_W = 64, QM_LD = QM_W + 4, QM_KS = QM_W / 4, QM_MAXP = 8 };   // QM_MAXP: towers whose trunk gradients one launch adds up

// host: can this chain run here?  (layer widths and outputs within one 64-wide stage; one output layer)
// (forward heads may carry the rows' INPUT as their action part: layer 0 is then computed in the chain, K = n_xa <= 128)
static inline bool qm_head_ok(const HtHead& h, bool has_out = true, bool fwd = false) {
  bool ok = h.L >= 1 && h.L <= GRL_MAX_LAYERS && h.n_xa <= (fwd ? 2 * QM_W : QM_W) &&
            (!has_out || (h.n_out == 1 && h.out_dim >= 1 && h.out_dim <= QM_W));
  for (int l = 0; l < h.L; ++l) ok = ok && h.hid[l] >= 1 && h.hid[l] <= QM_W;
  return ok;
}

#ifndef GRL_HEADS_TYPES_ONLY
#ifndef GRL_HOSTEMU

typedef float qm_f4 __attribute__((ext_vector_type(4)));
// Pointers read from a descriptor are generic to the compiler and become FLAT accesses, which count on the LDS counter as
// well: the wait in front of every stage barrier then also waited for the acknowledgement of the activations / gradients
// the stage before had just stored (a memory round trip per stage).  Typed as global (address space 1) they are
// global_load / global_store and the barriers order LDS traffic only.
typedef __attribute__((address_space(1))) float* qm_gp;
typedef __attribute__((address_space(1))) const float* qm_gcp;
#define QM_GW(p) ((qm_gp)(p))
#define QM_G(p) ((qm_gcp)(p))
struct __attribute__((aligned(16))) QmLds {
  float z[2][HT_RB][QM_LD];   // activations (forward) / gradients (backward) entering the next stage
  float o[HT_RB][QM_LD];      // backward: output gradients of the rows
};

// column `col` of the stage's [K, N] B operand, element (k, n) at W[k * sk + n * sn]: the lane of quarter q gets its 16
// reduction steps k = 16 q + s; k >= K or col >= N read as zero (out-of-range offset of the buffer descriptor)
__device__ __forceinline__ void qm_load_b(float (&bw)[QM_KS], const float* W, int K, int N, int sk, int sn, int col, int q) {
  const __amdgpu_buffer_rsrc_t rs = i2_rsrc(W);
  const int base = col * sn + QM_KS * q * sk;
#pragma unroll
  for (int s = 0; s < QM_KS; ++s) {
    const bool ok = QM_KS * q + s < K && col < N;
    bw[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (base + s * sk) * 4 : I2_OOB, 0, 0));
  }
}
// out(rows 4q .. 4q+3, column of this lane) = acc + sum_k zin[row][k] * B(k, col)
__device__ __forceinline__ qm_f4 qm_mma(const float (*zin)[QM_LD], const float (&bw)[QM_KS], int c, int q,
                                        qm_f4 acc = qm_f4{0.f, 0.f, 0.f, 0.f}) {
  float av[QM_KS];
#pragma unroll
  for (int j = 0; j < QM_KS / 4; ++j) {
    const qm_f4 v = *(const qm_f4*)(&zin[c][QM_KS * q + 4 * j]);
    av[4 * j] = v.x; av[4 * j + 1] = v.y; av[4 * j + 2] = v.z; av[4 * j + 3] = v.w;
  }
#pragma unroll
  for (int s = 0; s < QM_KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw[s], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ void qm_zero(QmLds& s) {
  for (int x = threadIdx.x; x < (int)(sizeof(QmLds) / 4); x += 256) ((float*)&s)[x] = 0.f;
}

// ---------------------------------------------------------------------------------------------- forward
// grid (B/16, 3 nets, D+1 towers): shared trunk (recomputed per tower, stored by tower 0) -> tower -> advantage / value
__global__ __launch_bounds__(256) void q_fwd_mfma_kernel(QFusedArgs a) {
  __shared__ QmLds s;
  if (a.tick_sc && (blockIdx.x | blockIdx.y | blockIdx.z) == 0 && threadIdx.x == 0) { if (a.tick_rng) a.tick_sc->rng_step += 1; adam_tick_device(a.tick_sc); }
  const HtHead& h = a.fwd[blockIdx.y * (a.D + 1) + blockIdx.z];
  const int t = threadIdx.x, w = t >> 6, l = t & 63, c = l & 15, q = l >> 4;
  const int n = 16 * w + c, row0 = blockIdx.x * HT_RB, B = a.B, L = h.L;
  // ---- every operand of the chain, requested at once
  float bw[GRL_MAX_LAYERS][QM_KS], bias[GRL_MAX_LAYERS];     // [0]: the output layer; [li]: hidden layer li
#pragma unroll
  for (int li = 1; li < GRL_MAX_LAYERS; ++li)
    if (li < L) {
      qm_load_b(bw[li], h.w[li], h.hid[li - 1], h.hid[li], h.hid[li], 1, n, q);
      bias[li] = n < h.hid[li] ? QM_G(h.b[li])[n] : 0.f;
    }
  qm_load_b(bw[0], h.ow[0], h.hid[L - 1], h.out_dim, h.out_dim, 1, n, q);
  bias[0] = n < h.out_dim ? QM_G(h.ob[0])[n] : 0.f;
  // layer 0 inside the chain (n_xa > 0: `xa` holds the rows' inputs, `w0a` the layer-0 kernel [n_xa, H0], n_xa <= 128):
  // two 64-deep stages over the inputs staged in z[0] | z[1] -- no GEMM launch in front of the chain
  const int nx = h.n_xa;
  float bx[2][QM_KS], xv[8], bx0 = 0.f;
  if (nx > 0) {
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) qm_load_b(bx[kc], h.w0a + (long)kc * QM_W * h.H0, nx - kc * QM_W, h.H0, h.H0, 1, n, q);
    bx0 = n < h.H0 ? QM_G(h.b0)[n] : 0.f;
    const int r = t >> 4, row = row0 + r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = (t & 15) + 16 * j;
      xv[j] = (row < B && col < nx) ? QM_G(h.xa)[(long)row * h.ld_xa + col] : 0.f;
    }
  }
  // layer 0 from the GEMM launch: z0 = relu(u (+ further partial sums, in order) + b0); thread -> (column t & 63, rows 4 (t >> 6) ..)
  const int n0 = t & 63, rg = t >> 6;
  float u0[4] = {0.f, 0.f, 0.f, 0.f};
  float b0 = 0.f;
  if (nx == 0 && n0 < h.H0) {
    b0 = QM_G(h.b0)[n0];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + 4 * rg + i;
      u0[i] = row < B ? QM_G(h.u)[(long)row * h.ldu + n0] : 0.f;
    }
    for (int sp = 1; sp < h.u_split; ++sp)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + 4 * rg + i;
        u0[i] += row < B ? QM_G(h.u)[sp * h.u_stride + (long)row * h.ldu + n0] : 0.f;
      }
  }
  qm_zero(s);
  __syncthreads();
  if (nx > 0) {
    const int r = t >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) s.z[j >> 2][r][(t & 15) + 16 * (j & 3)] = xv[j];
    __syncthreads();
    const qm_f4 acc = qm_mma(s.z[1], bx[1], c, q, qm_mma(s.z[0], bx[0], c, q));
    __syncthreads();                      // every wave has read the inputs: z[0] now receives the layer's output
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r2 = 4 * q + i, row = row0 + r2;
      const float v = n < h.H0 ? fmaxf(acc[i] + bx0, 0.f) : 0.f;
      s.z[0][r2][n] = v;
      if (row < B && n < h.H0 && h.z0) QM_GW(h.z0)[(long)row * h.H0 + n] = v;
    }
  } else if (n0 < h.H0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + 4 * rg + i;
      const float v = fmaxf(u0[i] + b0, 0.f);
      s.z[0][4 * rg + i][n0] = v;
      if (row < B && h.z0) QM_GW(h.z0)[(long)row * h.H0 + n0] = v;
    }
  }
  __syncthreads();
  // ---- hidden layers
#pragma unroll
  for (int li = 1; li < GRL_MAX_LAYERS; ++li)
    if (li < L) {
      const int Hout = h.hid[li];
      const qm_f4 acc = qm_mma(s.z[(li - 1) & 1], bw[li], c, q);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * q + i, row = row0 + r;
        const float v = n < Hout ? fmaxf(acc[i] + bias[li], 0.f) : 0.f;
        s.z[li & 1][r][n] = v;
        if (row < B && n < Hout && h.z[li]) QM_GW(h.z[li])[(long)row * Hout + n] = v;
      }
      __syncthreads();
    }
  // ---- output layer
  {
    const qm_f4 acc = qm_mma(s.z[(L - 1) & 1], bw[0], c, q);
    const int ld = h.ld_out ? h.ld_out : h.out_dim;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + 4 * q + i;
      if (row < B && n < h.out_dim) QM_GW(h.out[0])[(long)row * ld + n] = acc[i] + bias[0];
    }
  }
}

// ---------------------------------------------------------------------------------------------- backward
// one head (tower or trunk) for the 16 rows of this workgroup.
//   dz_parts == nullptr: s.o holds the output gradients [row][o]; g_{L-1} = mask * (d_out . ow^T)
//   dz_parts != nullptr: g_{L-1} = mask * (sum of n_parts partial gradients w.r.t. the last hidden activation) * scale
// then g_{l-1} = mask * (g_l . w_l^T) down to layer 0, and optionally d xa = g_0 . w0a^T.  Writes every g.
// `stage_in(zm)` runs after every operand of the chain has been requested: it clears the LDS block, brings the output
// gradients in (its own loads travel with the operands') and ends with a barrier; zm[l][i] are the forward activations of
// this lane's (rows 4q + i, column n) per layer.  `after(li)` runs behind the barrier that follows g_li's arrival in
// s.z[li & 1] (q_chain.h forms the layer's weight gradient there; the kernels of this file pass a no-op).
struct QmNoAfter { __device__ __forceinline__ void operator()(int) const {} };
template <class StageIn, class After = QmNoAfter>
__device__ __forceinline__ void qm_bwd_head(const HtHead& h, int row0, int B, QmLds& s, float* da, int ld_da,
                                            const float* dz_parts, int n_parts, long part_stride, float dz_scale, StageIn&& stage_in,
                                            After&& after = After()) {
  const int t = threadIdx.x, w = t >> 6, l = t & 63, c = l & 15, q = l >> 4;
  const int n = 16 * w + c, L = h.L;
  // ---- operands of every stage: [0] output layer (transposed), [l] hidden layer l (transposed), bwa: layer-0 action rows
  float bw[GRL_MAX_LAYERS][QM_KS], bwa[QM_KS];
  float zm[GRL_MAX_LAYERS][4];      // ReLU masks: the forward activations of this lane's (rows, column) per layer
  const int HL = h.hid[L - 1];
  if (!dz_parts) qm_load_b(bw[0], h.ow[0], h.out_dim, HL, 1, h.out_dim, n, q);
#pragma unroll
  for (int li = 1; li < GRL_MAX_LAYERS; ++li)
    if (li < L) qm_load_b(bw[li], h.w[li], h.hid[li], h.hid[li - 1], 1, h.hid[li], n, q);
  if (da) qm_load_b(bwa, h.w0a, h.H0, h.n_xa, 1, h.H0, n, q);
#pragma unroll
  for (int li = 0; li < GRL_MAX_LAYERS; ++li)
    if (li < L) {
      const float* zp = li == 0 ? h.z0 : h.z[li];
      const int H = h.hid[li];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + 4 * q + i;
        zm[li][i] = (row < B && n < H) ? QM_G(zp)[(long)row * H + n] : 0.f;
      }
    }
  // g_li of this lane's (rows 4q.., column n): mask, keep in LDS for the next stage, store.  (Called with compile-time li
  // only -- from unrolled loops -- so that zm / bw stay in registers.)
  auto put = [&](int li, const float (&mask)[4], const qm_f4& acc) {
    const int H = h.hid[li];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * q + i, row = row0 + r;
      const float v = mask[i] > 0.f ? acc[i] : 0.f;
      s.z[li & 1][r][n] = v;
      if (row < B && n < H) {
        float* gp = li == 0 ? h.g0 : h.g[li];          // (nullptr: nothing reads this gradient from memory -- q_chain.h)
        if (gp) QM_GW(gp)[(long)row * (li == 0 ? h.ldg0 : H) + n] = v;
      }
    }
  };
  // the trunk's partial gradients: all requested at once (a loop over a run-time count would wait for each in turn), added
  // in tower order
  float part[QM_MAXP][4];
  if (dz_parts) {
    const __amdgpu_buffer_rsrc_t rp = i2_rsrc(dz_parts);
#pragma unroll
    for (int p = 0; p < QM_MAXP; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + 4 * q + i;
        const bool ok = p < n_parts && row < B && n < HL;
        part[p][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, ok ? (int)((p * part_stride + (long)row * HL + n) * 4) : I2_OOB, 0, 0));
      }
  }
  stage_in(zm);
  // ---- g_{L-1}
  qm_f4 top = {0.f, 0.f, 0.f, 0.f};
  if (dz_parts) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float sum = 0.f;
#pragma unroll
      for (int p = 0; p < QM_MAXP; ++p) sum += part[p][i];      // (zeros beyond n_parts: x + 0 is exact)
      top[i] = sum * dz_scale;
    }
  } else {
    top = qm_mma(s.o, bw[0], c, q);
  }
#pragma unroll
  for (int li = 0; li < GRL_MAX_LAYERS; ++li)
    if (li == L - 1) put(li, zm[li], top);
  __syncthreads();
#pragma unroll
  for (int li = 0; li < GRL_MAX_LAYERS; ++li)
    if (li == L - 1) after(li);
  // ---- hidden layers
#pragma unroll
  for (int li = GRL_MAX_LAYERS - 1; li >= 1; --li)
    if (li < L) {
      put(li - 1, zm[li - 1], qm_mma(s.z[li & 1], bw[li], c, q));
      __syncthreads();
      after(li - 1);
    }
  // ---- d xa
  if (da) {
    const qm_f4 acc = qm_mma(s.z[0], bwa, c, q);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + 4 * q + i;
      if (row < B && n < h.n_xa) QM_GW(da)[(long)row * ld_da + n] = acc[i];
    }
  }
}

// grid (B/16, D+1): tower output + hidden layers of the online net; the gradient w.r.t. the trunk output leaves as one
// partial per tower
__global__ __launch_bounds__(256) void q_bwd_towers_mfma_kernel(QFusedArgs a) {
  __shared__ QmLds s;
  const int row0 = blockIdx.x * HT_RB, tw = blockIdx.y, t = threadIdx.x;
  const HtHead& h = a.bwd_tw[tw];
  // output gradients of the rows: 16 threads per row, up to 64 bins -> 4 per thread, requested before the chain's operands
  const int r = t >> 4, row = row0 + r, o0 = t & 15;
  float dv[4] = {0.f, 0.f, 0.f, 0.f};
  if (tw < a.D) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = o0 + 16 * j;
      if (row < a.B && o < a.nb) dv[j] = QM_G(a.d_adv)[((long)row * a.D + tw) * a.nbp + o];
    }
  } else if (o0 == 0 && row < a.B) {
    dv[0] = QM_G(a.d_v)[(long)row * a.ld_dv];
  }
  qm_bwd_head(h, row0, a.B, s, h.n_xa ? a.dh_part + (long)tw * a.B * a.Ht : nullptr, a.Ht, nullptr, 0, 0, 1.f, [&](const auto&) {
    qm_zero(s);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) s.o[r][o0 + 16 * j] = dv[j];
    __syncthreads();
  });
}

// grid (B/16): trunk -- partials added in tower order, scaled, masked, propagated down to layer 0
__global__ __launch_bounds__(256) void q_bwd_trunk_mfma_kernel(QFusedArgs a) {
  __shared__ QmLds s;
  qm_bwd_head(*a.bwd_tr, blockIdx.x * HT_RB, a.B, s, nullptr, 0, a.dh_part, a.D + 1, (long)a.B * a.Ht, a.trunk_scale, [&](const auto&) {
    qm_zero(s);
    __syncthreads();
  });
}

#endif  // GRL_HOSTEMU
#endif  // GRL_HEADS_TYPES_ONLY

}  // namespace grl
