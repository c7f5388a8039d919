// dp_kernels.h -- the one exchange step of the data-parallel update (the gradient bucket of every rank
// summed, the mean applied by every replica) as a hand-written two-shot all-reduce over peer-mapped memory, inside the
// update's hipGraph -- no host round trip and no collective library call per update.  RCCL (grasp_rl/parallel.py)
// stays as the baseline path.
//
// Every rank owns ONE exchange buffer (fine-grained device memory, exported with hipIpcGetMemHandle and mapped by all
// peers -- over xGMI between the GPUs of a node; two processes on one GPU map each other's buffer the same way):
//
//   ctl      epoch (completed exchanges), error, two block counters
//   ready[p] written by rank p (remotely) once its gradients of exchange e sit in ITS src    -> e
//   done[q]  written by rank q (remotely) once its reduced chunk of exchange e sits in MY res -> e
//   src[n]   this rank's gradient bucket      res[n]   the summed bucket
//
//   publish      grads -> src (16-byte copies); the last block stores ready[me] = e into every rank's buffer
//   reduce_push  reduce-scatter + all-gather by push: rank r owns chunk r = [r*c, (r+1)*c); it waits for ready[*] == e,
//                adds the chunk of all ranks IN RANK ORDER (one rank forms each sum, so every replica receives the same
//                bits, whatever the arrival order) and stores the sum into the res of every rank; its last block then
//                stores done[r] = e everywhere
//   apply        adam_polyak_kernel reading res (grad_scale 1 / world) after waiting for done[*] == e; its last block
//                advances ctl->epoch
//
// A rank overwrites its src for exchange e+1 only after done[*] == e, i.e. after every peer finished reading it; res is
// overwritten by peers only after this rank published e+1, i.e. after its apply of e (stream order).  Flag stores are
// system-scope releases behind __threadfence_system(), flag loads system-scope acquires.  Waits are BOUNDED (a peer
// that never arrives sets ctl->error, which the host reports -- the kernels never hang).
#pragma once
#include "elem_kernels.h"

namespace grl {

enum { DP_MAX_WORLD = 16, DP_SPIN_LIMIT = 1 << 22 };

struct DpCtl {
  uint32_t epoch;        // completed exchanges
  uint32_t error;        // 1: a bounded wait ran out
  uint32_t cnt_publish, cnt_reduce, cnt_apply;
  uint32_t pad[3];
  uint32_t ready[DP_MAX_WORLD];
  uint32_t done[DP_MAX_WORLD];
};

struct DpArgs {
  int rank, world;
  int64_t n, chunk;                 // bucket floats; floats per rank chunk (multiple of 4)
  const float* grads;               // this rank's gradient bucket (caller's grads arena)
  DpCtl* ctl[DP_MAX_WORLD];         // exchange buffers of all ranks as mapped HERE ([rank] = own)
  float* src[DP_MAX_WORLD];
  float* res[DP_MAX_WORLD];
};

__device__ __forceinline__ void dp_store_flag(uint32_t* p, uint32_t v) {
  __atomic_store_n(p, v, __ATOMIC_RELEASE);      // (system scope is the default of __atomic builtins without a scope)
}
__device__ __forceinline__ uint32_t dp_load_flag(const uint32_t* p) {
  return __atomic_load_n(p, __ATOMIC_ACQUIRE);
}
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

#ifdef GRL_HOSTEMU
static inline void dp_fence() {}
struct dp_f4 {     // (g++ has no ext_vector_type)
  float v[4];
  dp_f4& operator+=(const dp_f4& o) { for (int k = 0; k < 4; ++k) v[k] += o.v[k]; return *this; }
};
#else
__device__ __forceinline__ void dp_fence() { __threadfence_system(); }
typedef float dp_f4 __attribute__((ext_vector_type(4)));
#endif

__global__ __launch_bounds__(256) void dp_publish_kernel(DpArgs a) {
  DpCtl* mine = a.ctl[a.rank];
  const uint32_t e = mine->epoch + 1u;
  const int64_t n4 = a.n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    ((dp_f4*)a.src[a.rank])[i] = ((const dp_f4*)a.grads)[i];
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < a.n; i += 256) a.src[a.rank][i] = a.grads[i];
  dp_fence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t old = atomicAdd(&mine->cnt_publish, 1u);
    if (old == gridDim.x - 1) {
      mine->cnt_publish = 0u;
      dp_fence();
      for (int p = 0; p < a.world; ++p) dp_store_flag(&a.ctl[p]->ready[a.rank], e);
    }
  }
}

__global__ __launch_bounds__(256) void dp_reduce_push_kernel(DpArgs a) {
  DpCtl* mine = a.ctl[a.rank];
  const uint32_t e = mine->epoch + 1u;
  __shared__ int ok_s;
  if (threadIdx.x == 0) ok_s = dp_wait_all(mine, mine->ready, a.world, e) ? 1 : 0;
  __syncthreads();
  const int64_t lo = (int64_t)a.rank * a.chunk, hi = min(a.n, lo + a.chunk);
  if (ok_s) {
    // (chunk and lo are multiples of 4; the ragged tail of the bucket, if any, belongs to the last chunk)
    const int64_t q0 = lo >> 2, q1 = hi >> 2;
    for (int64_t i = q0 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < q1; i += (int64_t)gridDim.x * 256) {
      dp_f4 s = ((const dp_f4*)a.src[0])[i];
      for (int p = 1; p < a.world; ++p) s += ((const dp_f4*)a.src[p])[i];
      for (int q = 0; q < a.world; ++q) ((dp_f4*)a.res[q])[i] = s;
    }
    if (blockIdx.x == 0)
      for (int64_t i = max(lo, (hi >> 2) << 2) + threadIdx.x; i < hi; i += 256) {
        float s = a.src[0][i];
        for (int p = 1; p < a.world; ++p) s += a.src[p][i];
        for (int q = 0; q < a.world; ++q) a.res[q][i] = s;
      }
  }
  dp_fence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t old = atomicAdd(&mine->cnt_reduce, 1u);
    if (old == gridDim.x - 1) {
      mine->cnt_reduce = 0u;
      dp_fence();
      for (int q = 0; q < a.world; ++q) dp_store_flag(&a.ctl[q]->done[a.rank], e);
    }
  }
}

// Adam + Polyak of adam_polyak_kernel over the exchanged bucket (a.grads = res of this rank, grad_scale = 1 / world)
__global__ __launch_bounds__(256) void dp_apply_kernel(AdamArgs a, DpArgs d) {
  DpCtl* mine = d.ctl[d.rank];
  const uint32_t e = mine->epoch + 1u;
  __shared__ int ok_s;
  if (threadIdx.x == 0) ok_s = dp_wait_all(mine, mine->done, d.world, e) ? 1 : 0;
  __syncthreads();
  if (ok_s) {
    const float alpha = a.sc->adam_alpha;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_train; i += (int64_t)gridDim.x * 256) {
      const float g = grad_scaled(a.grads[i], a.grad_scale);
      float m = a.m[i], v = a.v[i], p = a.params[i];
      adam_elem(g, p, m, v, alpha, a.eps);
      a.m[i] = m;
      a.v[i] = v;
      a.params[i] = p;
      const int64_t k = i - a.src_ofs;
      if (k >= 0 && k < a.n_polyak) a.target[k] = polyak_elem(a.target[k], p, a.tau);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t old = atomicAdd(&mine->cnt_apply, 1u);
    if (old == gridDim.x - 1) {
      mine->cnt_apply = 0u;
      mine->epoch = e;        // read again only by the next exchange's kernels (stream order)
    }
  }
}

}  // namespace grl
