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
// contains its mass, both with LDS prefix scans.  Sums are float64 so that block boundaries do not
// depend on the summation tree.
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

#ifdef GRL_HOSTEMU
inline void per_blocksum_kernel(PerArgs a) {
  if (threadIdx.x != 0) return;
  const int64_t size = a.sc->replay_size;
  const int64_t i0 = (int64_t)blockIdx.x * PER_BLK;
  double s = 0.0;
  float m = INFINITY;
  for (int64_t i = i0; i < std::min(size, i0 + PER_BLK); ++i) { s += (double)a.p[i]; m = fminf(m, a.p[i]); }
  a.bsum[blockIdx.x] = s;
  a.bmin[blockIdx.x] = m;
}
inline void per_sample_kernel(PerArgs a, int n_blocks) {
  if (threadIdx.x != 0) return;
  const int k = blockIdx.x;
  const int64_t size = a.sc->replay_size;
  double total = 0.0;
  float pmin = INFINITY;
  for (int j = 0; j < n_blocks; ++j) { total += a.bsum[j]; pmin = fminf(pmin, a.bmin[j]); }
  float u;
  if (a.u) u = a.u[k];
  else {
    const uint64_t step = a.sc->rng_step;
    uint32_t c[4] = {(uint32_t)step, (uint32_t)(step >> 32), (uint32_t)k, 0x50455221u};
    philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    u = (float)(c[0] >> 8) * (1.f / 16777216.f);
  }
  const double mass = ((double)u + (double)k) * total / (double)a.B;
  double acc = 0.0;
  int j = 0;
  for (; j < n_blocks - 1; ++j) { if (acc + a.bsum[j] > mass) break; acc += a.bsum[j]; }
  int64_t i = (int64_t)j * PER_BLK;
  const int64_t iend = std::min(size, i + PER_BLK);
  for (; i < iend - 1; ++i) { if (acc + (double)a.p[i] > mass) break; acc += (double)a.p[i]; }
  a.idx_out[k] = i;
  const double ps = (double)a.p[i] / total, pm = (double)pmin / total;
  a.w_out[k] = (float)(pow(ps * (double)size, -(double)a.st->beta) / pow(pm * (double)size, -(double)a.st->beta));
  if (k == 0) { a.st->total = total; a.st->p_min = pmin; if (!a.u) a.sc->rng_used = 1u; }
}
#else
__global__ __launch_bounds__(256) void per_blocksum_kernel(PerArgs a) {
  __shared__ double ss[256];
  __shared__ float sm[256];
  const int t = threadIdx.x;
  const int64_t size = a.sc->replay_size;
  const int64_t i0 = (int64_t)blockIdx.x * PER_BLK + 4 * t;
  double s = 0.0;
  float m = INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (i0 + e < size) { const float v = a.p[i0 + e]; s += (double)v; m = fminf(m, v); }
  ss[t] = s; sm[t] = m;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) { ss[t] += ss[t + off]; sm[t] = fminf(sm[t], sm[t + off]); }
    __syncthreads();
  }
  if (t == 0) { a.bsum[blockIdx.x] = ss[0]; a.bmin[blockIdx.x] = sm[0]; }
}

// scan of 1024 doubles held 4 per thread (v[0..3] are consecutive elements 4t..4t+3): v becomes the
// inclusive prefix, ex the exclusive one -- ex of an element IS the inclusive value of its predecessor
// (same bits), so the intervals [ex, v) tile [0, total) without gaps or overlaps
__device__ __forceinline__ void per_scan1024(double (&v)[4], double (&ex)[4], double* lds, double& total) {
  const int t = threadIdx.x;
  v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
  lds[t] = v[3];
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {   // Hillis-Steele over the 256 per-thread totals
    const double x = t >= off ? lds[t - off] : 0.0;
    __syncthreads();
    lds[t] += x;
    __syncthreads();
  }
  const double before = t > 0 ? lds[t - 1] : 0.0;
  total = lds[255];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] += before;
  ex[0] = before; ex[1] = v[0]; ex[2] = v[1]; ex[3] = v[2];
  __syncthreads();
}

__global__ __launch_bounds__(256) void per_sample_kernel(PerArgs a, int n_blocks) {
  __shared__ double lds[256];
  __shared__ float smin[256];
  __shared__ int sh_j, sh_i;
  __shared__ double sh_before;
  const int t = threadIdx.x, k = blockIdx.x;
  const int64_t size = a.sc->replay_size;
  // ---- level 1: prefix over the block sums (n_blocks <= 1024)
  double v[4];
  float m = INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = 4 * t + e;
    v[e] = j < n_blocks ? a.bsum[j] : 0.0;
    if (j < n_blocks) m = fminf(m, a.bmin[j]);
  }
  smin[t] = m;
  double total, vx[4];
  per_scan1024(v, vx, lds, total);
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
  const double mass = ((double)u + (double)k) * total / (double)a.B;
  if (t == 0) { sh_j = n_blocks - 1; sh_before = total; }   // overwritten: exactly one block interval holds the mass
  __syncthreads();
  // the first block whose inclusive prefix exceeds the mass: exactly one (thread, e) sees the crossing
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = 4 * t + e;
    if (j < n_blocks && vx[e] <= mass && v[e] > mass) { sh_j = j; sh_before = vx[e]; }
  }
  __syncthreads();
  const int j = sh_j;
  const double before = sh_before;
  // ---- level 2: prefix inside block j
  const int64_t b0 = (int64_t)j * PER_BLK;
  const int64_t nin = min((int64_t)PER_BLK, size - b0);
  double w[4];
  float pv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = 4 * t + e;
    pv[e] = i < nin ? a.p[b0 + i] : 0.f;
    w[e] = (double)pv[e];
  }
  double btotal, wx[4];
  per_scan1024(w, wx, lds, btotal);
  if (t == 0) sh_i = (int)nin - 1;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = 4 * t + e;
    // (a mass within rounding of the block's end finds no crossing and keeps the last element)
    if (i < nin && before + wx[e] <= mass && before + w[e] > mass) sh_i = i;
  }
  __syncthreads();
  if (t == 0) {
    const int64_t idx = b0 + sh_i;
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
inline void per_update_kernel(PerArgs a, const int64_t* idx) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float mx = a.st->max_priority;
  for (int k = 0; k < a.B; ++k) {
    const float pr = a.prio_in[k] + a.eps;
    a.p[idx[k]] = powf(pr, a.alpha);
    mx = fmaxf(mx, pr);
  }
  a.st->max_priority = mx;
}
#else
__global__ __launch_bounds__(256) void per_update_kernel(PerArgs a, const int64_t* idx) {
  __shared__ int64_t sidx[1024];
  __shared__ float smax[256];
  const int t = threadIdx.x;
  for (int k = t; k < a.B; k += 256) sidx[k] = idx[k];
  __syncthreads();
  float mx = 0.f;
  for (int k = t; k < a.B; k += 256) {
    const float pr = a.prio_in[k] + a.eps;
    mx = fmaxf(mx, pr);
    const int64_t me = sidx[k];
    bool later = false;
    for (int j = k + 1; j < a.B; ++j) later = later || sidx[j] == me;
    if (!later) a.p[me] = powf(pr, a.alpha);
  }
  smax[t] = mx;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) smax[t] = fmaxf(smax[t], smax[t + off]);
    __syncthreads();
  }
  if (t == 0) a.st->max_priority = fmaxf(a.st->max_priority, smax[0]);
}
#endif

}  // namespace grl
