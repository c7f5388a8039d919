// dp_kernels.h -- the one exchange step of the data-parallel update (the gradient bucket of every rank
// summed, the mean applied by every replica) as hand-written all-reduces over peer-mapped memory, inside the update's
// hipGraph -- no host round trip and no collective library call per update.  RCCL (grasp_rl/parallel.py) stays as the
// baseline path.  Also here: the cross-rank merge of the VecNormalize running statistics kept on the device (8e: the
// per-pixel (count, mean, var) of every rank's env-step batch, Chan-merged in rank order by every replica).
//
// Every rank owns TWO allocations (flags, data), both exported with hipIpcGetMemHandle and mapped by all peers (over xGMI
// between the GPUs of a node; several processes on one GPU map each other's buffers the same way):
//
//   flags (fine-grained, a few hundred bytes): per channel (DpCtl)
//       epoch     exchanges begun on this channel (advanced by the exchange's first kernel)
//       ready[p]  written by rank p (remotely): its contribution to exchange e sits in ITS source array          -> e
//       done[q]   written by rank q (remotely): the sums of ITS chunk of exchange e sit in ITS red array         -> e
//       error     a bounded wait ran out, here or on a peer (sticky, poisons every rank)
//   data: three bucket-shaped arrays per rank: src | red | gathered
//
// Two-shot exchange (any world size; bytes per rank 2 (W-1)/W n):
//   K1  the last launch of the gradient computation (the slab reduction that forms every gradient) stores each sum
//       write-through into THIS rank's src as it forms it -- no copy kernel; one thread advances the epoch
//   W   announces ready[me] = e to every rank, waits for ready[*] >= e
//   K2  reduce-scatter: adds chunk me of all ranks IN RANK ORDER (remote 16-byte loads; one rank forms each sum, so every
//       replica receives the same bits) into ITS red
//   W   announces done[me] = e, waits for done[*] >= e
//   K3  all-gather by PULL fused with Adam + Polyak: reads each sum from the red of the chunk's owner (grad_scale
//       1 / world) -- nothing is ever written remotely but flags
// One-shot exchange (small worlds; bytes per rank (W-1) n, one flag round and one kernel less):
//   K1  as above into source buffer e & 1 (src / red alternate: a rank overwrites the buffer of exchange e - 2 only after
//       ready[*] >= e - 1, i.e. after every peer finished reading it -- no second flag round needed)
//   W   announces ready, waits
//   K2' every rank adds all W contributions of every element in rank order inside Adam
//
// Announcing and waiting is the job of ONE wave: dp_wait_kernel (one 64-thread block) sits between the kernel that wrote
// the data and the kernel that reads the peers' -- a kernel boundary on one stream (or graph edge) completes and drains
// every write-through store of the kernel before it, so the data kernels neither count themselves in nor drain nor fence
// (round 3 ended every data kernel with drain + barrier + atomic per block and a last-block election: 11 + 10.5 + 19 us
// for 5.4 MB), and they never SPIN: a full grid of polling blocks fills the machine's wave slots, and when the peer it
// waits for is another process on the same GPU (or this process's own later kernels) nothing else can be dispatched --
// measured: 4 processes on one MI355X dead-locked until the wait ran out, 2 processes lost 7 s on their first exchange.  Exchanged data is stored WRITE-THROUGH (buffer stores with sc0 | sc1) and
// loaded past the caches (sc0 | sc1 loads): nothing stays dirty in an L2 and nothing stale is read from one
// (cdna_hip_programming.md, Guideline 16, at SYSTEM scope because the other side may be another GPU).
// Waits are BOUNDED by the 100 MHz wall clock (GRL_TUNE dp_timeout_ms, default 120 s -- like a collective library a rank
// simply waits for a peer that is late, e.g. one running an evaluation callback); a rank that gives up raises `error`
// on EVERY rank and in a host-visible mailbox, and stops announcing, so that no replica applies a partial exchange
// silently: grl_train_step_allreduce / grl_allreduce_status fail on every rank.
#pragma once
#include "elem_kernels.h"

namespace grl {

enum { DP_MAX_WORLD = 16, DP_MAX_RANGES = 6, DP_CHANNELS = 3 };   // channel 2: running-statistics merge (dp_norm_*)
enum { DP_LOOK_TICKS = 20000000 };     // 0.2 s of the 100 MHz clock: a wait this long looks at the host's bound (dp_wait_all)

struct DpCtl {
  uint32_t epoch;        // exchanges begun on this channel
  uint32_t error;        // 1: a bounded wait ran out (here, or on a peer that then poisoned this rank)
  uint32_t next_buf;     // one-shot: source buffer of the NEXT exchange (written by K2', read by K1)
  uint32_t pad[5];
  uint32_t ready[DP_MAX_WORLD];
  uint32_t done[DP_MAX_WORLD];
};

// What a channel moves: up to DP_MAX_RANGES pieces of the flat bucket (offsets and lengths multiples of 4 floats), seen
// as one VIRTUAL array of n floats that is cut into world chunks.  vstart[r] = virtual position of piece r (vstart[n_ranges] = n).
struct DpArgs {
  int rank, world, n_ranges, oneshot;
  int64_t n, chunk;                 // virtual floats; virtual floats per rank chunk (multiple of 4)
  int64_t start[DP_MAX_RANGES], vstart[DP_MAX_RANGES + 1];
  uint64_t timeout_ticks;           // bound of every wait, in ticks of the 100 MHz wall clock
  uint32_t* host_err;               // page-locked mailbox of THIS rank's host: set when a wait ran out or a peer poisoned us
  DpCtl* ctl[DP_MAX_WORLD];         // this channel's flags in the exchange buffers of all ranks as mapped HERE ([rank] = own)
  float* src[DP_MAX_WORLD];         // bucket-shaped arrays of all ranks: gradients as published
  float* red[DP_MAX_WORLD];         //   two-shot: the sums of the chunk the rank owns; one-shot: the second source buffer
  float* gathered;                  // local bucket-shaped array holding every sum of this channel (dp_gather_kernel), or nullptr
};

#ifdef GRL_HOSTEMU
struct dp_f4 {     // (g++ has no ext_vector_type)
  float v[4];
  dp_f4& operator+=(const dp_f4& o) { for (int k = 0; k < 4; ++k) v[k] += o.v[k]; return *this; }
};
static inline dp_f4 dp_ld_sys(const float* base, int64_t qd) { return ((const dp_f4*)base)[qd]; }
static inline void dp_st_sys(float* base, int64_t qd, const dp_f4& v) { ((dp_f4*)base)[qd] = v; }
static inline uint32_t dp_load_flag(const uint32_t* p) { return *p; }
static inline void dp_store_flag(uint32_t* p, uint32_t v) { *p = v; }
static inline uint64_t dp_clock() { static uint64_t t = 0; return t += 1000; }   // (every poll "takes" 10 us)
static inline void dp_sleep() {}
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
__device__ __forceinline__ uint64_t dp_clock() { return wall_clock64(); }
__device__ __forceinline__ void dp_sleep() { __builtin_amdgcn_s_sleep(8); }
#endif

// this rank gives up: sticky error here, on every peer and in the host's mailbox
__device__ __forceinline__ void dp_fail(const DpArgs& a) {
  for (int p = 0; p < a.world; ++p) dp_store_flag(&a.ctl[p]->error, 1u);
  if (a.host_err) dp_store_flag(a.host_err, 1u);
}
// one thread: announce `mine[rank] = e` in every rank's flags -- unless this rank has already failed
__device__ __forceinline__ void dp_announce(const DpArgs& a, bool ready, uint32_t e) {
  if (dp_load_flag(&a.ctl[a.rank]->error)) { if (a.host_err) dp_store_flag(a.host_err, 1u); return; }
  for (int p = 0; p < a.world; ++p) dp_store_flag(ready ? &a.ctl[p]->ready[a.rank] : &a.ctl[p]->done[a.rank], e);
}
// one thread: wait until flags[0 .. world) >= target; false when the wait ran out or the channel is poisoned
__device__ __forceinline__ bool dp_wait_all(const DpArgs& a, const uint32_t* flags, uint32_t target) {
  DpCtl* mine = a.ctl[a.rank];
  const uint64_t t0 = dp_clock();
  for (int p = 0; p < a.world; ++p) {
    int polls = 0;
    while ((int32_t)(dp_load_flag(flags + p) - target) < 0) {
      if ((++polls & 63) == 0) {
        if (dp_load_flag(&mine->error)) { if (a.host_err) dp_store_flag(a.host_err, 1u); return false; }
        // the bound: `timeout_ticks` as captured (GRL_TUNE dp_timeout_ms), or -- grl_allreduce_set_timeout -- word 1 of the host
        // mailbox in milliseconds; that word is a read over the bus, so it is looked at only once the wait is already long
        const uint64_t waited = dp_clock() - t0;
        if (waited > (a.timeout_ticks < DP_LOOK_TICKS ? a.timeout_ticks : (uint64_t)DP_LOOK_TICKS)) {
          const uint32_t ms = a.host_err ? dp_load_flag(a.host_err + 1) : 0u;
          if (waited > (ms ? (uint64_t)ms * 100000ull : a.timeout_ticks)) { dp_fail(a); return false; }
        }
      }
      dp_sleep();
    }
  }
  return dp_load_flag(&mine->error) == 0u;
}
// virtual quad (4-float group) -> quad of the bucket
__device__ __forceinline__ int64_t dp_quad(const DpArgs& a, int64_t vq) {
  const int64_t v = vq << 2;
  int r = 0;
  for (int k = 1; k < a.n_ranges; ++k) r = v >= a.vstart[k] ? k : r;
  return (a.start[r] + (v - a.vstart[r])) >> 2;
}

// K1: the launch that ends a gradient computation (reduce_slabs_kernel: every trainable element = the sum of its
// split-reduction slabs; the loss workgroup forms the entropy coefficient's gradient) publishing what it sums: each sum
// also goes, write-through, into this rank's source array.  The pieces of the channel are exactly what the launch's
// descriptors cover.  Optionally carries the replay gather of the next update like reduce_slabs_gather_kernel (gx > 0).
// Block 0 advances the channel's epoch (no thread of this launch reads it).
template <bool GROUPED>
__global__ __launch_bounds__(256) void dp_reduce_slabs_kernel(const ReduceDesc* __restrict__ descs, const int2* __restrict__ tiles,
                                                             int n_tiles, LossArgs la, int has_loss, AdamArgs aa, DpArgs a,
                                                             GatherArgs ga, int gx) {
  DpCtl* mine = a.ctl[a.rank];
  float* out = (a.oneshot && (mine->next_buf & 1u)) ? a.red[a.rank] : a.src[a.rank];
  const long nb = n_tiles + has_loss, total = (long)gridDim.x, x = (long)blockIdx.x;
  const long before = x * nb / total, upto = (x + 1) * nb / total;     // (gx == 0: total == nb, every block reduces)
  if (upto > before) {
    reduce_slabs_body(descs, tiles, n_tiles, la, has_loss, aa, 0, (int)before, out);
    if (has_loss && (int)before == n_tiles) {
      __syncthreads();
      if (threadIdx.x == 0) st_sys_f1(out + (la.g_log_ent_coef - aa.grads), la.g_log_ent_coef[0]);
    }
    if (before == 0 && threadIdx.x == 0) mine->epoch = mine->epoch + 1u;
    return;
  }
  gather_norm_dispatch<GROUPED>(ga, gx, (int)(x - before));
}

// W: the only kernel that waits.  One wave: announce `ready` (which = 0) or `done` (which = 1) for the current exchange in
// every rank's flags, wait for every rank's.  It also keeps next_buf in step (the source buffer of the NEXT exchange).
__global__ __launch_bounds__(64) void dp_wait_kernel(DpArgs a, int which) {
  if (threadIdx.x != 0) return;
  DpCtl* mine = a.ctl[a.rank];
  const uint32_t e = mine->epoch;
  dp_announce(a, which == 0, e);
  if (which == 0) mine->next_buf = (e + 1u) & 1u;
  (void)dp_wait_all(a, which == 0 ? mine->ready : mine->done, e);
}
// a data kernel's entry check: the channel is healthy (the wait in front of it succeeded)
__device__ __forceinline__ bool dp_healthy(const DpArgs& a) {
  __shared__ int ok_s;
  if (threadIdx.x == 0) ok_s = dp_load_flag(&a.ctl[a.rank]->error) == 0u ? 1 : 0;
  __syncthreads();
  return ok_s != 0;
}

// K2 (two-shot)
__global__ __launch_bounds__(256) void dp_reduce_kernel(DpArgs a) {
  if (!dp_healthy(a)) return;
  const int64_t lo = (int64_t)a.rank * a.chunk, hi = min(a.n, lo + a.chunk);
  const int64_t q0 = lo >> 2, q1 = hi >> 2;      // (n, chunk and every piece are multiples of 4 floats; an empty chunk: q0 >= q1)
  for (int64_t i = q0 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < q1; i += (int64_t)gridDim.x * 256) {
    const int64_t qd = dp_quad(a, i);
    dp_f4 s = dp_ld_sys(a.src[0], qd);
    for (int p = 1; p < a.world; ++p) s += dp_ld_sys(a.src[p], qd);
    dp_st_sys(a.red[a.rank], qd, s);
  }
}

// All-gather half of a channel as a kernel of its own (the overlapped update runs it on the side lane, so that the pull over
// xGMI is hidden as well): every sum from the red array of its chunk's owner into this rank's `gathered` array.
__global__ __launch_bounds__(256) void dp_gather_kernel(DpArgs d) {
  if (!dp_healthy(d)) return;
  const int64_t n4 = d.n >> 2, c4 = d.chunk >> 2;
  for (int owner = 0; owner < d.world; ++owner)
    for (int64_t i = owner * c4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < min(n4, (owner + 1) * c4); i += (int64_t)gridDim.x * 256) {
      const int64_t qd = dp_quad(d, i);
      ((dp_f4*)d.gathered)[qd] = dp_ld_sys(d.red[owner], qd);
    }
}

// Adam + Polyak of adam_polyak_kernel on one quad of sums
__device__ __forceinline__ void dp_adam_quad(const AdamArgs& a, int64_t qd, const dp_f4& g4, float alpha) {
  const int64_t e0 = qd << 2;
  dp_f4 m4 = *(const dp_f4*)(a.m + e0), v4 = *(const dp_f4*)(a.v + e0), p4 = *(const dp_f4*)(a.params + e0);
#ifdef GRL_HOSTEMU
  const float* gp = g4.v; float* mp = m4.v; float* vp = v4.v; float* pp = p4.v;
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

// the sums of one channel, each read from the red array of its chunk's owner (or from `gathered`), applied
__device__ __forceinline__ void dp_apply_channel(const AdamArgs& a, const DpArgs& d, float alpha) {
  const int64_t n4 = d.n >> 2, c4 = d.chunk >> 2;
  for (int owner = 0; owner < d.world; ++owner)      // (one owner at a time: the array a wave loads from is wave-uniform)
    for (int64_t i = owner * c4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < min(n4, (owner + 1) * c4); i += (int64_t)gridDim.x * 256) {
      const int64_t qd = dp_quad(d, i);
      dp_adam_quad(a, qd, d.gathered ? ((const dp_f4*)d.gathered)[qd] : dp_ld_sys(d.red[owner], qd), alpha);
    }
}

// K3 (two-shot).  d: a channel whose sums are pulled here (or, with d.gathered set, were pulled by dp_gather_kernel on the
// side lane); d2: a second channel (overlapped update), or world = 0.  The pieces of the two channels partition the
// trainable bucket.  Nothing is applied unless BOTH channels are healthy.
__global__ __launch_bounds__(256) void dp_apply_kernel(AdamArgs a, DpArgs d, DpArgs d2) {
  __shared__ int ok_s;
  if (threadIdx.x == 0)
    ok_s = (dp_load_flag(&d.ctl[d.rank]->error) == 0u && (d2.world == 0 || dp_load_flag(&d2.ctl[d2.rank]->error) == 0u)) ? 1 : 0;
  __syncthreads();
  if (!ok_s) return;
  const float alpha = a.sc->adam_alpha;
  dp_apply_channel(a, d, alpha);
  if (d2.world > 0) dp_apply_channel(a, d2, alpha);
}

// K2' (one-shot): every rank adds the W contributions of every element in rank order and applies the mean
__global__ __launch_bounds__(256) void dp_apply_oneshot_kernel(AdamArgs a, DpArgs d) {
  if (!dp_healthy(d)) return;
  const float alpha = a.sc->adam_alpha;
  const bool odd = (d.ctl[d.rank]->epoch & 1u) != 0u;
  const int64_t n4 = d.n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t qd = dp_quad(d, i);
    dp_f4 s = dp_ld_sys(odd ? d.red[0] : d.src[0], qd);
    for (int p = 1; p < d.world; ++p) s += dp_ld_sys(odd ? d.red[p] : d.src[p], qd);
    dp_adam_quad(a, qd, s, alpha);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Running observation statistics of VecNormalize kept on the device, merged over the ranks: every rank's
// env-step batch gives per-element float32 moments (np.mean / np.var over its n observations, exactly as
// norm_update_kernel forms them); every replica then merges the W batches into its float64 running statistics IN RANK
// ORDER with the Chan update -- the arithmetic of grasp_rl.parallel.share_running_stats on the host (moments widened to
// float64 before they travel), so the replicas stay bit-identical to each other and to the host path.  Exchange memory: a
// fourth region of the data allocation, two moment blocks [mean f32 x elems | var f32 x elems | n] alternating by epoch
// (a block is overwritten two exchanges later, after ready[*] of the exchange in between: every peer has merged it).
struct DpNormArgs {
  DpArgs d;                  // channel 2: flags only (n, chunk, ranges unused)
  NormUpdateArgs nu;         // the local update's arguments (obs staged, running statistics, derived arrays)
  float* mom[DP_MAX_WORLD];  // every rank's pair of moment blocks as mapped here
  int64_t mom_stride;        // floats between the two blocks
};

// N1: this rank's batch moments into its moment block (write-through); thread 0 advances the epoch
__global__ __launch_bounds__(256) void dp_norm_moments_kernel(DpNormArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const NormUpdateArgs& u = a.nu;
  DpCtl* mine = a.d.ctl[a.d.rank];
  float* out = a.mom[a.d.rank] + (int64_t)(mine->next_buf & 1u) * a.mom_stride;
  if (i < u.elems) {
    float bm, bv;
    norm_batch_moments(u, i, bm, bv);
    st_sys_f1(out + i, bm);
    st_sys_f1(out + u.elems + i, bv);
  }
  if (i == 0) {
    st_sys_f1(out + 2 * (int64_t)u.elems, (float)u.n);
    mine->epoch = mine->epoch + 1u;
  }
}
// N2 (behind dp_wait_kernel on channel 2): merge every rank's moments in rank order, refresh the derived arrays
__global__ __launch_bounds__(256) void dp_norm_merge_kernel(DpNormArgs a) {
  if (!dp_healthy(a.d)) return;
  const uint32_t e = a.d.ctl[a.d.rank]->epoch;
  const NormUpdateArgs& u = a.nu;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int64_t off = (int64_t)(e & 1u) * a.mom_stride;
  double cnt = u.count[u.parity];
  double mean = 0.0, var = 0.0;
  if (i < u.elems) { mean = u.mean[i]; var = u.var[i]; }
  for (int p = 0; p < a.d.world; ++p) {
    const float* mp = a.mom[p] + off;
    const int bn = (int)ld_sys_f1(mp + 2 * (int64_t)u.elems);
    if (i < u.elems) norm_chan_merge(mean, var, cnt, ld_sys_f1(mp + i), ld_sys_f1(mp + u.elems + i), bn, false);
    else cnt += (double)bn;
  }
  if (i < u.elems) norm_store(u, i, mean, var);
  if (i == 0) u.count[u.parity ^ 1] = cnt;
}

}  // namespace grl
