// dp_kernels.h -- the one exchange step of the data-parallel update (the gradient bucket of every rank
// summed, the mean applied by every replica) as a hand-written two-shot all-reduce over peer-mapped memory, inside the
// update's hipGraph -- no host round trip and no collective library call per update.  RCCL (grasp_rl/parallel.py)
// stays as the baseline path.
//
// Every rank owns TWO allocations (flags, data), both exported with hipIpcGetMemHandle and mapped by all peers (over xGMI between the
// GPUs of a node; two processes on one GPU map each other's buffers the same way):
//
//   flags (fine-grained, a few hundred bytes): per channel (DpCtl)
//       ready[p]  written by rank p (remotely) once its gradients of exchange e sit in ITS src          -> e
//       done[q]   written by rank q (remotely) once the sums of ITS chunk of exchange e sit in ITS red  -> e
//       epoch (completed exchanges), error, block counters
//   data: src[n] this rank's gradients as published, red[n] the sums of the chunk this rank owns, gathered[n] (overlapped
//       update).  Stored write-through and loaded at system scope (below), so its caching policy does not matter to the
//       kernels: fine-grained by default (the conservative choice between GPUs), ordinary device memory with
//       GRL_DP_COARSE_DATA=1 -- 226 us per update either way on one MI355X.  (Round 3 first fenced every access instead:
//       5.4 MB then took 53 us to publish, 31 us to reduce and 28 us to apply; it was the fences, not the memory type.)
//
//   publish   grads -> src (16-byte copies); the last block stores ready[me] = e into every rank's flags
//   reduce    reduce-scatter: rank r owns chunk r = [r*c, (r+1)*c); it waits for ready[*] == e and adds the chunk of all
//             ranks IN RANK ORDER (remote 16-byte loads; one rank forms each sum, so every replica receives the same bits,
//             whatever the arrival order) into ITS red; its last block then stores done[r] = e everywhere
//   apply     all-gather by PULL fused with Adam + Polyak: after done[*] == e every rank reads each sum from the red of the
//             chunk's owner (grad_scale 1 / world) -- nothing is written remotely but flags; its last block advances epoch
//
// A rank overwrites its src for exchange e+1 only after done[*] == e, i.e. after every owner finished reading it; an owner
// overwrites its red only after ready[*] == e+1, i.e. after every rank's apply of e (stream order on that rank).
// Visibility: exchanged data is stored write-through and loaded past the caches (see dp_st_sys / dp_ld_sys below).
// Waits are BOUNDED (a peer that never arrives sets error, which the host reports -- the kernels never hang).
#pragma once
#include "elem_kernels.h"

namespace grl {

enum { DP_MAX_WORLD = 16, DP_SPIN_LIMIT = 1 << 22, DP_MAX_RANGES = 6, DP_CHANNELS = 2 };

// One CHANNEL = one independent exchange with its own flags (a buffer carries DP_CHANNELS of them).  The plain update uses
// channel 0 over the whole bucket.  The overlapped update (engine.hip, grl_allreduce_set_overlap) exchanges the dense
// layers' gradients on channel 0 from a side lane of the graph while the convolution backward runs, then the
// convolution gradients on channel 1; the apply kernel waits for both.
struct DpCtl {
  uint32_t epoch;        // completed exchanges of this channel
  uint32_t error;        // 1: a bounded wait ran out
  uint32_t cnt_publish, cnt_reduce, cnt_apply;
  uint32_t pad[3];
  uint32_t ready[DP_MAX_WORLD];
  uint32_t done[DP_MAX_WORLD];
};

// What a channel moves: up to DP_MAX_RANGES pieces of the flat bucket (offsets and lengths multiples of 4 floats), seen
// as one VIRTUAL array of n floats that is cut into world chunks.  vstart[r] = virtual position of piece r (vstart[n_ranges] = n).
struct DpArgs {
  int rank, world, n_ranges, pad_;
  int64_t n, chunk;                 // virtual floats; virtual floats per rank chunk (multiple of 4)
  int64_t start[DP_MAX_RANGES], vstart[DP_MAX_RANGES + 1];
  const float* grads;               // this rank's gradient bucket (caller's grads arena)
  DpCtl* ctl[DP_MAX_WORLD];         // this channel's flags in the exchange buffers of all ranks as mapped HERE ([rank] = own)
  float* src[DP_MAX_WORLD];         // bucket-shaped arrays of all ranks: gradients as published
  float* red[DP_MAX_WORLD];         //                                    the sums of the chunk the rank owns
  float* gathered;                  // local bucket-shaped array holding every sum of this channel (dp_gather_kernel), or nullptr
};

// How data and flags travel (cdna_hip_programming.md, Guideline 16, "publish / consume"; here at SYSTEM scope because the
// other side may be another GPU):
//   * exchanged data is stored WRITE-THROUGH (buffer stores with sc0 | sc1): nothing stays dirty in an L2, so no block needs
//     a release fence (an L2 write-back per block -- issued by every wave it cost 50 us per 5 MB kernel, by one thread per
//     block still 13 us); every storing wave drains its stores (s_waitcnt vmcnt(0)), the block's barrier follows, ONE thread
//     counts the block in with a relaxed device-scope atomic, and the last block of the grid stores the flags;
//   * exchanged data is loaded with system-scope loads (sc0 | sc1: they are served from memory, never from a stale line), so
//     no block needs an acquire fence (a cache invalidation per block) either;
//   * flags are polled by ONE thread per block with relaxed loads.
#ifdef GRL_HOSTEMU
struct dp_f4 {     // (g++ has no ext_vector_type)
  float v[4];
  dp_f4& operator+=(const dp_f4& o) { for (int k = 0; k < 4; ++k) v[k] += o.v[k]; return *this; }
};
static inline dp_f4 dp_ld_sys(const float* base, int64_t qd) { return ((const dp_f4*)base)[qd]; }
static inline void dp_st_sys(float* base, int64_t qd, const dp_f4& v) { ((dp_f4*)base)[qd] = v; }
static inline uint32_t dp_load_flag(const uint32_t* p) { return *p; }
static inline void dp_store_flag(uint32_t* p, uint32_t v) { *p = v; }
static inline bool dp_block_done(uint32_t* counter) {
  if (threadIdx.x != 0) return false;
  if (++*counter != gridDim.x) return false;
  *counter = 0u;
  return true;
}
#else
typedef float dp_f4 __attribute__((ext_vector_type(4)));
enum { DP_SYS = 1 | 16 };     // buffer-instruction cache policy: sc0 | sc1 = system scope
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dp_rsrc(const float* p) {
  const uint64_t a = (uint64_t)p;     // (made provably wave-uniform: no waterfall loop around the buffer instructions)
  const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ dp_f4 dp_ld_sys(const float* base, int64_t qd) {
  return __builtin_bit_cast(dp_f4, __builtin_amdgcn_raw_buffer_load_b128(dp_rsrc(base), (int)(qd << 4), 0, DP_SYS));
}
__device__ __forceinline__ void dp_st_sys(float* base, int64_t qd, const dp_f4& v) {
  typedef unsigned int dp_u4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dp_u4, v), dp_rsrc(base), (int)(qd << 4), 0, DP_SYS);
}
__device__ __forceinline__ uint32_t dp_load_flag(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void dp_store_flag(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// End of a block's data phase.  Returns true in thread 0 of the last block of the grid (which then stores the flags).
__device__ __forceinline__ bool dp_block_done(uint32_t* counter) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave: its write-through stores have reached memory
  __syncthreads();
  if (threadIdx.x != 0) return false;
  const uint32_t old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old != gridDim.x - 1) return false;
  __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}
#endif
// thread 0 of the block waits until flags[0 .. world) >= target; false on time-out (error flag set)
__device__ __forceinline__ bool dp_wait_all(DpCtl* mine, const uint32_t* flags, int world, uint32_t target) {
  bool ok = true;
  for (int p = 0; p < world && ok; ++p) {
    int spins = 0;
    while ((int32_t)(dp_load_flag(flags + p) - target) < 0) {
      if (++spins > DP_SPIN_LIMIT) { mine->error = 1u; ok = false; break; }
#ifndef GRL_HOSTEMU
      __builtin_amdgcn_s_sleep(8);
#endif
    }
  }
  return ok;
}
// virtual quad (4-float group) -> quad of the bucket
__device__ __forceinline__ int64_t dp_quad(const DpArgs& a, int64_t vq) {
  const int64_t v = vq << 2;
  int r = 0;
  for (int k = 1; k < a.n_ranges; ++k) r = v >= a.vstart[k] ? k : r;
  return (a.start[r] + (v - a.vstart[r])) >> 2;
}

__global__ __launch_bounds__(256) void dp_publish_kernel(DpArgs a) {
  DpCtl* mine = a.ctl[a.rank];
  const uint32_t e = mine->epoch + 1u;
  const int64_t n4 = a.n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t qd = dp_quad(a, i);
    dp_st_sys(a.src[a.rank], qd, ((const dp_f4*)a.grads)[qd]);
  }
  if (dp_block_done(&mine->cnt_publish))
    for (int p = 0; p < a.world; ++p) dp_store_flag(&a.ctl[p]->ready[a.rank], e);
}

// The last launch of the gradient computation (reduce_slabs_kernel: every trainable element = the sum of its split-reduction
// slabs; the loss workgroup forms the entropy coefficient's gradient) publishing what it sums: each sum also goes, write-through,
// into this rank's src -- no copy kernel.  The pieces of the channel are exactly what the launch's descriptors cover.
__global__ __launch_bounds__(256) void dp_reduce_slabs_publish_kernel(const ReduceDesc* __restrict__ descs, const int2* __restrict__ tiles,
                                                                     int n_tiles, LossArgs la, int has_loss, AdamArgs aa, DpArgs a) {
  DpCtl* mine = a.ctl[a.rank];
  const uint32_t e = mine->epoch + 1u;
  reduce_slabs_body(descs, tiles, n_tiles, la, has_loss, aa, 0, (int)blockIdx.x, a.src[a.rank]);
  if (has_loss && (int)blockIdx.x == n_tiles) {
    __syncthreads();
    if (threadIdx.x == 0) st_sys_f1(a.src[a.rank] + (la.g_log_ent_coef - aa.grads), la.g_log_ent_coef[0]);
  }
  if (dp_block_done(&mine->cnt_publish))
    for (int p = 0; p < a.world; ++p) dp_store_flag(&a.ctl[p]->ready[a.rank], e);
}

__global__ __launch_bounds__(256) void dp_reduce_kernel(DpArgs a) {
  DpCtl* mine = a.ctl[a.rank];
  const uint32_t e = mine->epoch + 1u;
  __shared__ int ok_s;
  if (threadIdx.x == 0) ok_s = dp_wait_all(mine, mine->ready, a.world, e) ? 1 : 0;
  __syncthreads();
  const int64_t lo = (int64_t)a.rank * a.chunk, hi = min(a.n, lo + a.chunk);
  if (ok_s) {
    const int64_t q0 = lo >> 2, q1 = hi >> 2;      // (n, chunk and every piece are multiples of 4 floats)
    for (int64_t i = q0 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < q1; i += (int64_t)gridDim.x * 256) {
      const int64_t qd = dp_quad(a, i);
      dp_f4 s = dp_ld_sys(a.src[0], qd);
      for (int p = 1; p < a.world; ++p) s += dp_ld_sys(a.src[p], qd);
      dp_st_sys(a.red[a.rank], qd, s);
    }
  }
  if (dp_block_done(&mine->cnt_reduce))
    for (int q = 0; q < a.world; ++q) dp_store_flag(&a.ctl[q]->done[a.rank], e);
}

// All-gather half of a channel as a kernel of its own (the overlapped update runs it on the side lane, so that the pull over
// xGMI is hidden as well): every sum from the red array of its chunk's owner into this rank's `gathered` array.
__global__ __launch_bounds__(256) void dp_gather_kernel(DpArgs d) {
  DpCtl* mine = d.ctl[d.rank];
  __shared__ int ok_s;
  if (threadIdx.x == 0) ok_s = dp_wait_all(mine, mine->done, d.world, mine->epoch + 1u) ? 1 : 0;
  __syncthreads();
  if (!ok_s) return;
  const int64_t n4 = d.n >> 2, c4 = d.chunk >> 2;
  for (int owner = 0; owner < d.world; ++owner)
    for (int64_t i = owner * c4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < min(n4, (owner + 1) * c4); i += (int64_t)gridDim.x * 256) {
      const int64_t qd = dp_quad(d, i);
      ((dp_f4*)d.gathered)[qd] = dp_ld_sys(d.red[owner], qd);
    }
}

// Adam + Polyak of adam_polyak_kernel on the sums, each read from the red array of its chunk's owner (grad_scale = 1 / world).
// d: the channel whose counters pace the launch (0); d2: a second channel (overlapped update), or world = 0.  The pieces of
// the two channels partition the trainable bucket.
__device__ __forceinline__ void dp_apply_channel(const AdamArgs& a, const DpArgs& d, float alpha) {
  const int64_t n4 = d.n >> 2, c4 = d.chunk >> 2;
  for (int owner = 0; owner < d.world; ++owner)      // (one owner at a time: the array a wave loads from is wave-uniform)
  for (int64_t i = owner * c4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < min(n4, (owner + 1) * c4); i += (int64_t)gridDim.x * 256) {
    const int64_t qd = dp_quad(d, i);
    const dp_f4 g4 = d.gathered ? ((const dp_f4*)d.gathered)[qd] : dp_ld_sys(d.red[owner], qd);
    const int64_t e0 = qd << 2;
    dp_f4 m4 = *(const dp_f4*)(a.m + e0), v4 = *(const dp_f4*)(a.v + e0), p4 = *(const dp_f4*)(a.params + e0);
#ifdef GRL_HOSTEMU
    float* gp = (float*)g4.v; float* mp = m4.v; float* vp = v4.v; float* pp = p4.v;
#else
    const float gp[4] = {g4.x, g4.y, g4.z, g4.w};
    float mp[4] = {m4.x, m4.y, m4.z, m4.w}, vp[4] = {v4.x, v4.y, v4.z, v4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#endif
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      adam_elem(grad_scaled(gp[u], a.grad_scale), pp[u], mp[u], vp[u], alpha, a.eps);
      const int64_t k = e0 + u - a.src_ofs;
      if (k >= 0 && k < a.n_polyak) a.target[k] = polyak_elem(a.target[k], pp[u], a.tau);
    }
#ifndef GRL_HOSTEMU
    m4 = dp_f4{mp[0], mp[1], mp[2], mp[3]}; v4 = dp_f4{vp[0], vp[1], vp[2], vp[3]}; p4 = dp_f4{pp[0], pp[1], pp[2], pp[3]};
#endif
    *(dp_f4*)(a.m + e0) = m4; *(dp_f4*)(a.v + e0) = v4; *(dp_f4*)(a.params + e0) = p4;
  }
}

__global__ __launch_bounds__(256) void dp_apply_kernel(AdamArgs a, DpArgs d, DpArgs d2) {
  DpCtl* mine = d.ctl[d.rank];
  const uint32_t e = mine->epoch + 1u;
  __shared__ int ok_s;
  if (threadIdx.x == 0) {
    bool ok = dp_wait_all(mine, mine->done, d.world, e);
    if (ok && d2.world > 0) {
      DpCtl* m2 = d2.ctl[d2.rank];
      ok = dp_wait_all(m2, m2->done, d2.world, m2->epoch + 1u);
    }
    ok_s = ok ? 1 : 0;
  }
  __syncthreads();
  if (ok_s) {
    const float alpha = a.sc->adam_alpha;
    dp_apply_channel(a, d, alpha);
    if (d2.world > 0) dp_apply_channel(a, d2, alpha);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t old = atomicAdd(&mine->cnt_apply, 1u);
    if (old == gridDim.x - 1) {
      mine->cnt_apply = 0u;
      mine->epoch = e;        // read again only by the next exchange's kernels (stream order)
      if (d2.world > 0) { DpCtl* m2 = d2.ctl[d2.rank]; m2->epoch = m2->epoch + 1u; }
    }
  }
}

}  // namespace grl
