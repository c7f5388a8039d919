// per_kernels.h -- proportional prioritised replay on the device (DQN / BDQ: `prioritized_replay: True`,
// /root/reference/config/gripper_grasp.yaml:102, simplified_object_picking.yaml:101,110).  The sampler is
// stable-baselines 2.10.1 `PrioritizedReplayBuffer` + `SegmentTree` (un-vendored dependency; restated operation by
// operation in oracle/per.py, whose header quotes the published code):
//
//   add:     leaf[i] = max_priority ** alpha                              (float64 power of the float32 running maximum)
//   sample:  total_s = it_sum.sum(0, len(storage) - 1)                    = reduce over [0, size - 2]: the published code
//                                                                           leaves the last stored transition out
//            mass_k  = u_k * total_s                                      (2.10.x: np.random.random(size=B) * total)
//                      u_k * L + k * L, L = total_s / B                   (q_per_stratified: baselines / SB < 2.10)
//            idx_k   = find_prefixsum_idx(mass_k)                         (left child if its sum > mass, else subtract, right)
//            w_k     = (leaf[idx_k] / root * N) ** -beta / (min_leaf / root * N) ** -beta
//   update:  leaf[idx_k] = float32(|td_k| + eps) ** float32(alpha)        (NumPy: float32 array ** Python float), stored
//            max_priority = max(max_priority, |td_k| + eps)                 as float64 like the reference's tree
//
// The leaves are FLOAT64 like the reference's `_value` array (8 B per stored transition).  Instead of the array
// segment trees (pointer chasing, one update at a time on the host) the ring is cut into blocks of 1024 leaves: a
// block-sum pass (HBM-bound) followed by one workgroup per sample that rebuilds the upper tree over the <= 1024 block
// sums and then the sub-tree of the one block that contains its mass, both in LDS with the reference's association
// order (node = left + right over the next power of two), and walks them exactly like `find_prefixsum_idx`.
#pragma once
#include "elem_kernels.h"

namespace grl {

enum { PER_BLK = 1024 };

struct PerState {          // lives in the replay arena next to the priorities
  double total;            // it_sum.sum(): root of the sum tree
  double total_s;          // it_sum.sum(0, size - 1): the mass the sampler spreads (excludes the last stored leaf)
  double p_min;            // it_min.min()
  double tail_w;           // right-nested sum of the sampled range inside the block that holds leaf size - 2
  double beta;             // importance-weight exponent of the current call (annealed by the host)
  float max_priority;      // running max of (|td| + eps), starts at 1
  float pad;
};

struct PerArgs {
  double* p;               // [cap] leaves of the sum tree: priority ** alpha
  double* bsum;            // [n_blocks]
  double* bmin;            // [n_blocks]
  PerState* st;
  DevScalars* sc;          // replay_size, rng_step
  uint64_t seed;
  int B;
  int stratified;          // 0: mass = u * total (stable-baselines 2.10.x)   1: u * L + k * L (baselines / SB < 2.10)
  float alpha, eps;        // float32(prioritized_replay_alpha), float32(prioritized_replay_eps): the update's operands
  double alpha64;          // prioritized_replay_alpha as the Python float (the add's exponent)
  const double* u;         // [B] explicit uniforms in [0,1) (parity tests) or nullptr: Philox, 53 bits
  int64_t* idx_out;        // [B]
  float* w_out;            // [B]
  const float* prio_in;    // [B] |td| summed over branches (q_loss_kernel)
};

// float32 power the way NumPy / glibc deliver it (correctly rounded in all but ~1e-8 of the cases): evaluated in
// float64 and rounded once
__device__ __forceinline__ float per_powf(float x, float y) { return (float)pow((double)x, (double)y); }

// `reduce(0, e)` of the reference's SegmentTree over one 1024-leaf tree `tr` (node 1 = root, leaves at 1024 + i):
// _reduce_helper descends towards leaf e and adds the left sibling whenever it turns right, the recursion returning
// `left + (rest)` -- a right-nested sum along the path.  `stop_val(depth)` supplies the value of the node the descent
// ends in when it runs out of this tree (depth 10) before a node's range ends exactly at e.
template <class F>
__device__ __forceinline__ double per_prefix_reduce(const double* tr, int64_t e, int64_t leaf_span, F below) {
  int node = 1, n = 0;
  int64_t ns = 0, ne = (int64_t)PER_BLK * leaf_span - 1;
  double c[10], fin = 0.0;
  for (int d = 0;; ++d) {
    if (ne == e) { fin = tr[node]; break; }
    if (d == 10) { fin = below(); break; }
    const int64_t mid = (ns + ne) >> 1;
    if (e <= mid) { node = 2 * node; ne = mid; }
    else { c[n++] = tr[2 * node]; node = 2 * node + 1; ns = mid + 1; }
  }
  for (int k = n - 1; k >= 0; --k) fin = c[k] + fin;
  return fin;
}

// (GRL_ELEM_TYPES_ONLY: a translation unit that only needs the types and the device bodies -- heads.hip, whose trunk launch can
//  carry the priority write-back -- the kernels themselves are compiled in engine.hip)
#ifndef GRL_ELEM_TYPES_ONLY
// new transitions enter with the maximal priority seen so far
__global__ __launch_bounds__(256) void per_add_kernel(PerArgs a, int64_t pos, int n, int64_t cap) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < n) a.p[(pos + k) % cap] = pow((double)a.st->max_priority, a.alpha64);
}
#endif

// The sums follow the association order of the reference's SumSegmentTree (stable_baselines/common/
// segment_tree.py, v2.10.1): a binary tree over the next power of two >= capacity, node = left + right in
// float64, and the sampler walks it from the root (left child if its sum exceeds the remaining mass, else
// subtract it and go right).  A block of 1024 leaves is an aligned subtree of height 10, the block sums
// are the leaves of the upper tree: both trees are rebuilt level by level in LDS and walked exactly like the
// reference walks its array, so the drawn index is bit-identical to oracle/per.py on the same leaves.

// one uniform in [0, 1) with 53 random bits (np.random.random's resolution) for sample k of minibatch `step`
__device__ __forceinline__ double per_uniform(const PerArgs& a, int k) {
  if (a.u) return a.u[k];
  const uint64_t step = a.sc->rng_step;
  uint32_t c[4] = {(uint32_t)step, (uint32_t)(step >> 32), (uint32_t)k, 0x50455221u};
  philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
  return ((double)(c[0] >> 5) * 67108864.0 + (double)(c[1] >> 6)) * (1.0 / 9007199254740992.0);
}
// mass of sample k.  Products and the sum are separate roundings (no contraction), as NumPy evaluates them.
__device__ __forceinline__ double per_mass(const PerArgs& a, int k, double u, double total_s) {
#pragma clang fp contract(off)
  if (!a.stratified) return u * total_s;
  const double every_range_len = total_s / (double)a.B;
  const double m0 = u * every_range_len, m1 = (double)k * every_range_len;
  return m0 + m1;
}
__device__ __forceinline__ float per_weight(double leaf, double total, double pmin, int64_t size, double beta) {
  const double p_min = pmin / total, max_weight = pow(p_min * (double)size, -beta);
  const double p_sample = leaf / total;
  return (float)(pow(p_sample * (double)size, -beta) / max_weight);
}

#ifdef GRL_HOSTEMU
#include "per_kernels_ref1.h"   // tests/hostemu: the emulation build only
#else
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
// The same tree from the leaves each thread holds in registers (l0 .. l3 = leaves 4 t .. 4 t + 3, also written to
// tr[PER_BLK + 4 t ..] by the caller -- no barrier needed in between), with the minimum over `m` riding on the same
// barriers when `sm` is given (result in sm[0]).  node = left + right at every level, exactly as per_tree_build forms it;
// what changes is the number of workgroup barriers: the two lowest levels come from registers, and from 64 nodes down a
// level is written and read by wave 0 alone -- LDS operations of one wave complete in order, a wave-level fence is enough.
// 4 barriers instead of 10 (18 with the separate minimum reduction).
__device__ __forceinline__ double per_tree_build_regs(double* tr, double l0, double l1, double l2, double l3, double* sm = nullptr,
                                                      double m = 0.0) {
  const int t = threadIdx.x;
  const double s01 = l0 + l1, s23 = l2 + l3;
  tr[PER_BLK / 2 + 2 * t] = s01;
  tr[PER_BLK / 2 + 2 * t + 1] = s23;
  tr[PER_BLK / 4 + t] = s01 + s23;
  if (sm) sm[t] = m;
  __syncthreads();
  for (int n = PER_BLK / 8; n >= 64; n >>= 1) {
    if (t < n) {
      tr[n + t] = tr[2 * (n + t)] + tr[2 * (n + t) + 1];
      if (sm) sm[t] = fmin(sm[t], sm[t + n]);
    }
    __syncthreads();
  }
  if (t < 64) {
#pragma unroll
    for (int n = 32; n >= 1; n >>= 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (t < n) {
        tr[n + t] = tr[2 * (n + t)] + tr[2 * (n + t) + 1];
        if (sm) sm[t] = fmin(sm[t], sm[t + n]);
      }
    }
  }
  __syncthreads();
  return tr[1];
}
// the reference's find_prefixsum_idx over one 1024-leaf tree; every thread walks (LDS broadcasts)
// (two levels per LDS round trip: the left child and the left children of BOTH children are read together -- the same
//  comparisons and subtractions in the same order as one level at a time, five dependent reads instead of ten)
__device__ __forceinline__ int per_tree_walk(const double* tr, double& rem) {
  int i = 1;
#pragma unroll 1
  for (int lvl = 0; lvl < 10; lvl += 2) {
    const double left = tr[2 * i], ll = tr[4 * i], rl = tr[4 * i + 2];
    double next;
    if (left > rem) { i = 2 * i; next = ll; }
    else { rem -= left; i = 2 * i + 1; next = rl; }
    if (next > rem) i = 2 * i;
    else { rem -= next; i = 2 * i + 1; }
  }
  return i - PER_BLK;
}

#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void per_blocksum_kernel(PerArgs a) {
  __shared__ double tr[2 * PER_BLK];
  __shared__ double sm[256];
  const int t = threadIdx.x;
  const int64_t size = a.sc->replay_size;
  const int64_t i0 = (int64_t)blockIdx.x * PER_BLK + 4 * t;
  double m = INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    double v = 0.0;
    if (i0 + e < size) { v = a.p[i0 + e]; m = fmin(m, v); }
    tr[PER_BLK + 4 * t + e] = v;
  }
  sm[t] = m;
  __syncthreads();
  const double root = per_tree_build(tr);
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) sm[t] = fmin(sm[t], sm[t + off]);
    __syncthreads();
  }
  if (t == 0) {
    a.bsum[blockIdx.x] = root;
    a.bmin[blockIdx.x] = sm[0];
    // the block that holds leaf e = size - 2, the last one of the reference's sum(0, size - 1): its part of that sum
    const int64_t e = size - 2;
    if (e >= 0 && e / PER_BLK == (int64_t)blockIdx.x)
      a.st->tail_w = per_prefix_reduce(tr, e % PER_BLK, 1, [] { return 0.0; });
  }
}

#endif
// g / do_gather: the row's transition is gathered (and VecNormalized) right here once its replay index is known -- the
// gather launch of the update disappears; the launch also opens the update (Adam step size) when g.adam_tick is set.
// Sample k by the first 256 threads of a workgroup (`tr`: 2 * PER_BLK doubles, `smin`: 257 doubles of LDS; smin[256] carries
// total_s): per_sample_kernel's workgroups, or extra workgroups of the launch that ends the update before (q_apply_kernels.h).
__device__ __forceinline__ void per_sample_body(const PerArgs& a, int n_blocks, const GatherArgs& g, int do_gather, const int k,
                                                double* tr, double* smin) {
  double& s_total_s = smin[256];
  const int t = threadIdx.x;
  const int64_t size = a.sc->replay_size;
  // ---- upper tree: leaves = block sums (zero beyond n_blocks, like the unused leaves of the reference's tree)
  double m = INFINITY, lf[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = 4 * t + e;
    lf[e] = j < n_blocks ? a.bsum[j] : 0.0;
    tr[PER_BLK + j] = lf[e];
    if (j < n_blocks) m = fmin(m, a.bmin[j]);
  }
  const double total = per_tree_build_regs(tr, lf[0], lf[1], lf[2], lf[3], smin, m);
  const double pmin = smin[0];
  if (t == 0) {
    const double tail = a.st->tail_w;
    s_total_s = size >= 2 ? per_prefix_reduce(tr, size - 2, PER_BLK, [tail] { return tail; }) : 0.0;
  }
  __syncthreads();
  const double total_s = s_total_s;
  double rem = per_mass(a, k, per_uniform(a, k), total_s);
  const int j = per_tree_walk(tr, rem);
  __syncthreads();                                   // everyone is done reading the upper tree
  // ---- the subtree of block j
  const int64_t b0 = (int64_t)j * PER_BLK;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = 4 * t + e;
    lf[e] = (j < n_blocks && b0 + i < size) ? a.p[b0 + i] : 0.0;
    tr[PER_BLK + i] = lf[e];
  }
  per_tree_build_regs(tr, lf[0], lf[1], lf[2], lf[3]);
  const int i = per_tree_walk(tr, rem);
  const int64_t idx = min(b0 + (int64_t)i, size - 1);     // (a mass that rounds up to the full total would walk off the stored range)
  if (do_gather) {
    gather_row_device(g, k, idx);
    if (g.adam_tick && k == 0 && t == 0) adam_tick_device(g.sc);
  }
  if (t < 64) {
    // importance weight (per_weight): its two float64 powers -- a few hundred instructions each -- side by side in lanes 0 and 1
    // of wave 0 instead of one after the other in lane 0; the same operations on the same values
    const double beta = a.st->beta;
    const double base = ((t == 0 ? pmin : a.p[idx]) / total) * (double)size;
    const double pw = t < 2 ? pow(base, -beta) : 0.0;
    const double max_weight = __shfl(pw, 0, 64), p_pow = __shfl(pw, 1, 64);
    if (t == 0) a.w_out[k] = (float)(p_pow / max_weight);
  }
  if (t == 0) {
    a.idx_out[k] = idx;
    if (k == 0) { a.st->total = total; a.st->total_s = total_s; a.st->p_min = pmin; if (!a.u) a.sc->rng_used = 1u; }   // q_loss_kernel advances rng_step
  }
}
#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void per_sample_kernel(PerArgs a, int n_blocks, GatherArgs g, int do_gather) {
  __shared__ double tr[2 * PER_BLK];
  __shared__ double smin[257];
  per_sample_body(a, n_blocks, g, do_gather, (int)blockIdx.x, tr, smin);
}
#endif
#endif

// Block sums kept current by the launch that writes the priorities back (multi-update calls: between two updates of one
// call nothing else touches the leaves, so the block-sum pass over the WHOLE ring in front of every sampler is only
// needed for the first update of a call).  Workgroup of sample k -- if k is the first sample that falls into its
// 1024-leaf block -- rebuilds that block's tree from the stored leaves with the new values of every sampled leaf of the
// block put in (the write-back workgroup of the same launch stores the same values: whichever this one reads is
// overridden here), in the association order of per_blocksum_kernel.  Called by 256 threads (the caller retires the other
// waves of a larger workgroup first); `tr`: 2 * PER_BLK doubles, `sm`: 256 doubles, `sidx`: B int64 of LDS.
__device__ __forceinline__ void per_refresh_body(const PerArgs& a, const int64_t* idx_g, int k, double* tr, double* sm, int64_t* sidx) {
  const int t = threadIdx.x;
  for (int j = t; j < a.B; j += 256) sidx[j] = idx_g[j];      // (one round trip: a scan over global memory waits for every entry)
  __syncthreads();
  const int64_t* idx = sidx;
  const int64_t blk = idx[k] / PER_BLK;
  for (int j = 0; j < k; ++j)
    if (idx[j] / PER_BLK == blk) return;           // (uniform) an earlier sample's workgroup rebuilds this block
  const int64_t size = a.sc->replay_size, b0 = blk * PER_BLK;
  if (t < 256) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = b0 + 4 * t + e;
      tr[PER_BLK + 4 * t + e] = i < size ? a.p[i] : 0.0;
    }
  }
  __syncthreads();
  if (t < 256)
    for (int j = t; j < a.B; j += 256) {
      const int64_t me = idx[j];
      if (me / PER_BLK != blk) continue;
      bool later = false;
      for (int j2 = j + 1; j2 < a.B; ++j2) later = later || idx[j2] == me;
      if (!later) tr[PER_BLK + (int)(me - b0)] = (double)per_powf(a.prio_in[j] + a.eps, a.alpha);
    }
  __syncthreads();
  // the two lowest levels of the tree from registers (node = left + right, as per_tree_build forms them), the levels above
  // together with the minimum: 8 barriers instead of 18
  {
    const double l0 = tr[PER_BLK + 4 * t], l1 = tr[PER_BLK + 4 * t + 1], l2 = tr[PER_BLK + 4 * t + 2], l3 = tr[PER_BLK + 4 * t + 3];
    double m = INFINITY;
    if (b0 + 4 * t < size) m = fmin(m, l0);
    if (b0 + 4 * t + 1 < size) m = fmin(m, l1);
    if (b0 + 4 * t + 2 < size) m = fmin(m, l2);
    if (b0 + 4 * t + 3 < size) m = fmin(m, l3);
    const double s01 = l0 + l1, s23 = l2 + l3;
    tr[PER_BLK / 2 + 2 * t] = s01;
    tr[PER_BLK / 2 + 2 * t + 1] = s23;
    tr[PER_BLK / 4 + t] = s01 + s23;
    sm[t] = m;
  }
  __syncthreads();
  for (int n = PER_BLK / 8; n >= 1; n >>= 1) {
    if (t < n) {
      tr[n + t] = tr[2 * (n + t)] + tr[2 * (n + t) + 1];
      sm[t] = fmin(sm[t], sm[t + n]);
    }
    __syncthreads();
  }
  if (t == 0) {
    a.bsum[blk] = tr[1];
    a.bmin[blk] = sm[0];
    const int64_t e = size - 2;
    if (e >= 0 && e / PER_BLK == blk) a.st->tail_w = per_prefix_reduce(tr, e % PER_BLK, 1, [] { return 0.0; });
  }
}

// priorities of the minibatch just trained on.  A transition drawn twice keeps the value of its LAST
// occurrence (what NumPy's fancy assignment leaves behind): every sample writes unless a later sample
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
      if (!later) a.p[me] = (double)per_powf(pr, a.alpha);
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
#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void per_update_kernel(PerArgs a, const int64_t* idx) {
  __shared__ int64_t sidx[1024];
  __shared__ float smax[256];
  per_update_body(a, idx, sidx, smax);
}
#endif
#endif

}  // namespace grl
