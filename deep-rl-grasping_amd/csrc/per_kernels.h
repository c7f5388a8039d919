// per_kernels.h -- proportional prioritised replay on the device (DQN / BDQ: `prioritized_replay: True`,
// /root/reference/config/gripper_grasp.yaml:102, simplified_object_picking.yaml:101,110; the sampler is
// stable-baselines 2.10.1 `PrioritizedReplayBuffer` [SURVEY.md A.6]):
//
//   add:     p[i] = max_priority ** alpha
//   sample:  mass_k = (u_k + k) * total / B  (stratified), idx_k = smallest i with prefix_sum(i+1) > mass_k
//            w_k = (p[idx_k]/total * N) ** -beta / (p_min/total * N) ** -beta
//   update:  p[idx_k] = (|td_k| + eps) ** alpha ; max_priority = max(max_priority, |td_k| + eps)
//
// Instead of the sum / min segment trees of the reference implementation (pointer chasing, one update
// at a time) the ring is cut into blocks of 1024 priorities: a block-sum pass (HBM-bound, 4 B/transition)
// followed by one workgroup per sample that scans the <= 1024 block sums and then the one block that
// contains its mass, both as LDS segment trees in float64 with the reference's association order.
#pragma once
#include "elem_kernels.h"

namespace grl {

enum { PER_BLK = 1024 };

struct PerState {          // lives in the replay arena next to the priorities
  double total;            // sum of p over the stored transitions
  float p_min;             // min of p over the stored transitions
  float max_priority;      // running max of (|td| + eps), starts at 1
  float beta;              // importance-weight exponent of the current call (annealed by the host)
  float pad;
};

struct PerArgs {
  float* p;                // [cap] priority ** alpha
  double* bsum;            // [n_blocks]
  float* bmin;             // [n_blocks]
  PerState* st;
  DevScalars* sc;          // replay_size, rng_step
  uint64_t seed;
  int B;
  float alpha, eps;
  const float* u;          // [B] explicit uniforms in [0,1) (parity tests) or nullptr: Philox
  int64_t* idx_out;        // [B]
  float* w_out;            // [B]
  const float* prio_in;    // [B] |td| summed over branches (q_loss_kernel)
};

// new transitions enter with the maximal priority seen so far
__global__ __launch_bounds__(256) void per_add_kernel(PerArgs a, int64_t pos, int n, int64_t cap) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < n) a.p[(pos + k) % cap] = powf(a.st->max_priority, a.alpha);
}

// The sums follow the association order of the reference's SumSegmentTree (stable_baselines/common/
// segment_tree.py, v2.10.1): a binary tree over the next power of two >= capacity, node = left + right in
// float64, and the sampler walks it from the root (left child if its sum exceeds the remaining mass, else
// subtract it and go right).  A block of 1024 priorities is an aligned subtree of height 10, the block sums
// are the leaves of the upper tree: both trees are rebuilt level by level in LDS and walked exactly like the
// reference walks its array, so the drawn index is bit-identical to oracle/per.py on the same priorities.
#ifdef GRL_HOSTEMU
#include "per_kernels_ref1.h"   // tests/hostemu: the emulation build only
#else
__global__ __launch_bounds__(256) void per_blocksum_kernel(PerArgs a) {
  __shared__ double ss[256];
  __shared__ float sm[256];
  const int t = threadIdx.x;
  const int64_t size = a.sc->replay_size;
  const int64_t i0 = (int64_t)blockIdx.x * PER_BLK + 4 * t;
  double v[4];
  float m = INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[e] = 0.0;
    if (i0 + e < size) { const float x = a.p[i0 + e]; v[e] = (double)x; m = fminf(m, x); }
  }
  ss[t] = (v[0] + v[1]) + (v[2] + v[3]);    // the aligned 4-leaf subtree
  sm[t] = m;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {   // adjacent pairs: node = left + right, level by level
    if ((t & (2 * off - 1)) == 0) { ss[t] = ss[t] + ss[t + off]; sm[t] = fminf(sm[t], sm[t + off]); }
    __syncthreads();
  }
  if (t == 0) { a.bsum[blockIdx.x] = ss[0]; a.bmin[blockIdx.x] = sm[0]; }
}

// tr[1024 + i] holds leaf i (all 1024 written, barrier done by the caller): builds the internal nodes
// tr[1 .. 1024) level by level; returns the root.  Ends with a barrier.
__device__ __forceinline__ double per_tree_build(double* tr) {
  const int t = threadIdx.x;
  for (int n = PER_BLK / 2; n >= 1; n >>= 1) {
    for (int i = n + t; i < 2 * n; i += 256) tr[i] = tr[2 * i] + tr[2 * i + 1];
    __syncthreads();
  }
  return tr[1];
}
// the reference's find_prefixsum_idx over one 1024-leaf tree; every thread walks (LDS broadcasts)
__device__ __forceinline__ int per_tree_walk(const double* tr, double& rem) {
  int i = 1;
#pragma unroll 1
  for (int lvl = 0; lvl < 10; ++lvl) {
    const double left = tr[2 * i];
    if (left > rem) i = 2 * i;
    else { rem -= left; i = 2 * i + 1; }
  }
  return i - PER_BLK;
}

// g / do_gather: the row's transition is gathered (and VecNormalized) right here once its replay index is known -- the
// gather launch of the update disappears; the launch also opens the update (Adam step size) when g.adam_tick is set.
__global__ __launch_bounds__(256) void per_sample_kernel(PerArgs a, int n_blocks, GatherArgs g, int do_gather) {
  __shared__ double tr[2 * PER_BLK];
  __shared__ float smin[256];
  const int t = threadIdx.x, k = blockIdx.x;
  const int64_t size = a.sc->replay_size;
  // ---- upper tree: leaves = block sums (zero beyond n_blocks, like the unused leaves of the reference's tree)
  float m = INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = 4 * t + e;
    tr[PER_BLK + j] = j < n_blocks ? a.bsum[j] : 0.0;
    if (j < n_blocks) m = fminf(m, a.bmin[j]);
  }
  smin[t] = m;
  __syncthreads();
  const double total = per_tree_build(tr);
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) smin[t] = fminf(smin[t], smin[t + off]);
    __syncthreads();
  }
  const float pmin = smin[0];
  float u;
  if (a.u) u = a.u[k];
  else {
    const uint64_t step = a.sc->rng_step;
    uint32_t c[4] = {(uint32_t)step, (uint32_t)(step >> 32), (uint32_t)k, 0x50455221u};
    philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    u = (float)(c[0] >> 8) * (1.f / 16777216.f);
  }
  double rem = ((double)u + (double)k) * total / (double)a.B;
  const int j = per_tree_walk(tr, rem);
  __syncthreads();                                   // everyone is done reading the upper tree
  // ---- the subtree of block j
  const int64_t b0 = (int64_t)j * PER_BLK;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = 4 * t + e;
    tr[PER_BLK + i] = (j < n_blocks && b0 + i < size) ? (double)a.p[b0 + i] : 0.0;
  }
  __syncthreads();
  per_tree_build(tr);
  const int i = per_tree_walk(tr, rem);
  const int64_t idx = min(b0 + (int64_t)i, size - 1);     // a mass that rounds up to the total walks off the stored range
  if (do_gather) {
    gather_row_device(g, k, idx);
    if (g.adam_tick && k == 0 && t == 0) adam_tick_device(g.sc);
  }
  if (t == 0) {
    a.idx_out[k] = idx;
    const double ps = (double)a.p[idx] / total, pm = (double)pmin / total;
    a.w_out[k] = (float)(pow(ps * (double)size, -(double)a.st->beta) / pow(pm * (double)size, -(double)a.st->beta));
    if (k == 0) { a.st->total = total; a.st->p_min = pmin; if (!a.u) a.sc->rng_used = 1u; }   // q_loss_kernel advances rng_step
  }
}
#endif

// priorities of the minibatch just trained on.  A transition drawn twice keeps the value of its LAST
// occurrence (what a Python loop over the batch leaves behind): every sample writes unless a later sample
// names the same index.  One workgroup; B <= 1024.
#ifdef GRL_HOSTEMU
#include "per_kernels_ref2.h"   // tests/hostemu: the emulation build only
#else
// (workgroups of 256 threads or more: the first 256 do the work, all reach the barriers)
__device__ __forceinline__ void per_update_body(const PerArgs& a, const int64_t* idx, int64_t* sidx, float* smax) {
  const int t = threadIdx.x;
  if (t < 256)
    for (int k = t; k < a.B; k += 256) sidx[k] = idx[k];
  __syncthreads();
  float mx = 0.f;
  if (t < 256) {
    for (int k = t; k < a.B; k += 256) {
      const float pr = a.prio_in[k] + a.eps;
      mx = fmaxf(mx, pr);
      const int64_t me = sidx[k];
      bool later = false;
      for (int j = k + 1; j < a.B; ++j) later = later || sidx[j] == me;
      if (!later) a.p[me] = powf(pr, a.alpha);
    }
    smax[t] = mx;
  }
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) smax[t] = fmaxf(smax[t], smax[t + off]);
    __syncthreads();
  }
  if (t == 0) a.st->max_priority = fmaxf(a.st->max_priority, smax[0]);
}
__global__ __launch_bounds__(256) void per_update_kernel(PerArgs a, const int64_t* idx) {
  __shared__ int64_t sidx[1024];
  __shared__ float smax[256];
  per_update_body(a, idx, sidx, smax);
}
#endif

}  // namespace grl
