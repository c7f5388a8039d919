// q_apply_kernels.h -- the tail of a DQN / BDQ update as one launch: slab reduction, per-variable clip_by_norm, Adam,
// the batch means of the loss launch and (prioritised replay) the priority write-back.
#pragma once
#include "elem_kernels.h"
#include "per_kernels.h"

namespace grl {

// DQN / BDQ full update: slab sums -> per-variable clip_by_norm -> Adam in ONE launch, one workgroup per variable
// (= per reduction descriptor: the host checks that every trainable variable is exactly one descriptor).  The sums stay
// in LDS between the phases.  Arithmetic and summation orders are those of reduce_slabs_kernel (slabs in order),
// clip_by_norm_kernel (strided partial sums, tree -- over 1024 instead of 256 threads) and adam_polyak_kernel: three launches at the launch floor
// (4.8 + 8.6 + 4.8 us under graph replay) become one.
#define GRL_QAPPLY_MAX 16384   /* floats of LDS for the summed gradient of one variable */
// QNextArgs: the sampler of the NEXT update riding on this launch (prioritised multi-update calls whose trunk launch has already
// written this update's priorities back and refreshed the block sums: q_chain.h, QChainArgs.per_wb): n_sample workgroups behind
// everything else, workgroup k = per_sample_kernel's workgroup k (sum-tree walk, importance weight, the row's gather).
// uniform_gx > 0 (uniform replay, plan_q "q_pf"): the riders are the workgroups of gather_norm_kernel instead -- uniform_gx x B x 2,
// each drawing its row's index itself (GatherArgs.use_rng with rng_ahead = 1: the counter still holds this update's value)
struct QNextArgs {
  PerArgs per; GatherArgs g; int n_sample, n_blocks;
  int uniform_gx;
};
#ifdef GRL_HOSTEMU
#include "q_apply_kernels_ref1.h"   // tests/hostemu: the emulation build only
#else
// (1024 threads: the largest variable of the reference networks, 101 x 64, is then one batch of loads per phase)
__global__ __launch_bounds__(1024) void q_reduce_clip_adam_kernel(const ReduceDesc* __restrict__ descs, int n_desc, float clip, AdamArgs aa,
                                                                 const float* row_part, int rows, int finish, PerArgs per, const int64_t* per_idx,
                                                                 int n_extra, QNextArgs nx) {
  __shared__ __attribute__((aligned(16))) float gsum[GRL_QAPPLY_MAX];   // (also the index scratch of the write-back block)
  constexpr int NT = 1024;
  __shared__ float red3[3][256];
  __shared__ float red[NT];
  const int t = threadIdx.x;
  if ((int)blockIdx.x >= n_desc + n_extra) {      // the next update's sampler (n_extra: the extra workgroups in front of it)
    if (t >= 256) return;               // (whole waves retire: the barriers below are among the remaining four)
    if (nx.uniform_gx > 0) {
      const int k = (int)blockIdx.x - n_desc - n_extra, gx = nx.uniform_gx;
      gather_norm_body(nx.g, k % gx, (k / gx) % nx.g.B, k / (gx * nx.g.B));
      return;
    }
    static_assert(GRL_QAPPLY_MAX * sizeof(float) >= (2 * PER_BLK + 257) * sizeof(double), "the sampler's trees live in gsum");
    per_sample_body(nx.per, nx.n_blocks, nx.g, 1, (int)blockIdx.x - n_desc - n_extra, (double*)gsum, (double*)gsum + 2 * PER_BLK);
    return;
  }
  if ((int)blockIdx.x == n_desc) {      // extra workgroup: batch means of the loss launch's row sums (deferred q_loss_finish)
    // (finish bit 1: the Philox counter was advanced by the trunk launch -- the sampler riding on THIS launch reads it)
    if (finish & 1) q_finish_sums(const_cast<DevScalars*>(aa.sc), row_part, rows, red3, (finish & 2) == 0);
    return;
  }
  if ((int)blockIdx.x == n_desc + 1) {  // second extra workgroup (prioritised replay): priority write-back of this minibatch
    per_update_body(per, per_idx, (int64_t*)gsum, red);
    return;
  }
  if ((int)blockIdx.x > n_desc + 1) {   // one more per sample (multi-update calls): the block sums the NEXT sampler reads
    if (t >= 256) return;               // (whole waves retire: the barriers below are among the remaining four)
    per_refresh_body(per, per_idx, (int)blockIdx.x - n_desc - 2, (double*)gsum, (double*)gsum + 2 * PER_BLK,
                     (int64_t*)((double*)gsum + 2 * PER_BLK + 256));
    return;
  }
  const ReduceDesc d = descs[blockIdx.x];
  constexpr int U = 8;                  // elements per thread in flight: the loops below are chains of load batches
  // the Adam state of the first batch of elements (all of them for every variable of the reference networks) is requested
  // together with the slabs: one memory round trip per workgroup instead of two
  const int64_t e0 = d.dst - aa.grads;
  float p0[U], m0[U], v0[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = u * NT + t;
    const int64_t e = e0 + (i < d.n ? i : 0);
    p0[u] = aa.params[e]; m0[u] = aa.m[e]; v0[u] = aa.v[e];
  }
  float ss = 0.f;
  for (int base = 0; base < d.n; base += NT * U) {
    float acc[U];
    const __attribute__((address_space(1))) float* sp[U];     // (global, not generic: no FLAT loads on the LDS counter)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * NT + t;
      const int ic = i < d.n ? i : 0;
      sp[u] = (const __attribute__((address_space(1))) float*)d.src + (d.row_len > 0 ? (long)(ic / d.row_len) * d.src_ld + ic % d.row_len : ic);
      acc[u] = 0.f;
    }
    for (int k = 0; k < d.splits; ++k) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = sp[u][(long)k * d.slab_stride];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] += v[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {       // (ascending i per thread, then a tree over the 1024 partial sums)
      const int i = base + u * NT + t;
      if (i < d.n) { gsum[i] = acc[u]; ss += acc[u] * acc[u]; }
    }
  }
  float sc = 1.f;
  if (clip > 0.f) {
    red[t] = ss;
    __syncthreads();
    for (int off = NT / 2; off >= 64; off >>= 1) {
      if (t < off) red[t] += red[t + off];
      __syncthreads();
    }
    if (t < 64) {       // the last six levels live in wave 0 alone: same tree, a wave-level fence instead of a barrier of 16 waves
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (t < off) red[t] += red[t + off];
      }
    }
    __syncthreads();
    sc = clip / fmaxf(sqrtf(red[0]), clip);
  }
  const float alpha = aa.sc->adam_alpha;
  for (int base = 0; base < d.n; base += NT * U) {
    float p[U], m[U], v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * NT + t;
      const int64_t e = e0 + (i < d.n ? i : 0);
      if (base == 0) { p[u] = p0[u]; m[u] = m0[u]; v[u] = v0[u]; }
      else { p[u] = aa.params[e]; m[u] = aa.m[e]; v[u] = aa.v[e]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * NT + t;
      if (i < d.n) {
        float g = gsum[i];
        if (clip > 0.f) g *= sc;
        ((__attribute__((address_space(1))) float*)d.dst)[i] = g;
        adam_elem(grad_scaled(g, aa.grad_scale), p[u], m[u], v[u], alpha, aa.eps);
        aa.params[e0 + i] = p[u]; aa.m[e0 + i] = m[u]; aa.v[e0 + i] = v[u];
      }
    }
  }
}
// the batch means alone (split compute / apply path of a plan whose loss launch defers them)
__global__ __launch_bounds__(256) void q_finish_kernel(DevScalars* sc, const float* row_part, int rows) {
  __shared__ float red3[3][256];
  q_finish_sums(sc, row_part, rows, red3);
}
#endif

}  // namespace grl
