// q_kernels.h -- DQN / BDQ networks as row-local chains (stable-baselines
// `deepq` dueling towers, /root/reference/manipulation_main/training/sb_helper.py:159-165, and the branching
// fork, :210-224).  At batch 32-64 the per-layer GEMM launches of the Q networks are pure launch latency
// (ten launches for 0.01 GFLOP); here every (net, tower) pair is one workgroup chain over 16 batch rows,
// built from the head primitives of heads_kernels.h:
//
//   forward   grid (B/16, 3 nets, D+1 towers): shared trunk (recomputed per tower: a few kFLOP) -> tower
//             hidden layers -> advantage / value output.  Layer 0 comes from one GEMM launch (K = obs_dim).
//   backward  grid (B/16, D+1): tower output + hidden layers of the online net; the gradient w.r.t. the trunk
//             output leaves as one partial per tower;
//             grid (B/16): trunk -- partials added in tower order, scaled (BDQ rescales the gradient entering
//             the trunk by 1/(D+1)), masked, propagated down to layer 0.
//
// Weight gradients stay on the implicit-GEMM launch: they read the activations / pre-activation gradients
// these kernels leave in global memory, exactly as the per-layer path does.
#pragma once
#include "heads_kernels.h"

namespace grl {

struct QFusedArgs {
  const HtHead* fwd;      // [3][D+1]: trunk + tower chains of online(s), online(s'), target(s')
  const HtHead* bwd_tw;   // [D+1]: towers of online(s); with a trunk its output is the head's `xa` input
  const HtHead* bwd_tr;   // trunk of online(s), or nullptr
  int B, D, nb, Ht;       // rows, branches, bins per branch, trunk output width (0: no trunk)
  const float* d_adv; const float* d_v;   // loss gradients [B, D*nbp], [B] with stride ld_dv (QLossArgs)
  int nbp, ld_dv;
  float* dh_part;         // [D+1][B, Ht]: gradient w.r.t. the trunk output, one partial per tower
  float trunk_scale;
  int mfma;               // every chain fits q_mfma.h's 64-wide MFMA stages: the launchers take those kernels
  DevScalars* tick_sc;    // forward launch only, optional: this launch OPENS the update -- one thread fixes the Adam step size and
                          // advances the beta powers (adam_tick_device).  Prioritised multi-update calls: the sampler of update
                          // t + 1 rides on the apply launch of update t, which still reads update t's step size
  int tick_rng;           // ... and advances the Philox counter first (uniform multi-update calls, plan_q "q_pf": this update's
                          // minibatch was drawn with counter + 1 by riders of the previous update's apply launch)
};

#ifndef GRL_HEADS_TYPES_ONLY
#ifdef GRL_HOSTEMU
#include "q_kernels_ref1.h"   // tests/hostemu: the emulation build only
#else
__global__ __launch_bounds__(256) void q_fwd_fused_kernel(QFusedArgs a) {
  __shared__ HtLds s;
  if (a.tick_sc && (blockIdx.x | blockIdx.y | blockIdx.z) == 0 && threadIdx.x == 0) { if (a.tick_rng) a.tick_sc->rng_step += 1; adam_tick_device(a.tick_sc); }
  const HtHead& h = a.fwd[blockIdx.y * (a.D + 1) + blockIdx.z];
  ht_fwd_head(h, blockIdx.x * HT_RB, a.B, s, false);
}

__global__ __launch_bounds__(256) void q_bwd_towers_kernel(QFusedArgs a) {
  __shared__ HtLds s;
  const int row0 = blockIdx.x * HT_RB, tw = blockIdx.y, t = threadIdx.x;
  const HtHead& h = a.bwd_tw[tw];
  const int r = t & (HT_RB - 1), row = row0 + r;
  if (tw < a.D) {
    for (int o = t / HT_RB; o < a.nb; o += 256 / HT_RB)
      s.oT[o][r] = row < a.B ? a.d_adv[((long)row * a.D + tw) * a.nbp + o] : 0.f;
  } else if (t < HT_RB) {
    s.oT[0][r] = row < a.B ? a.d_v[(long)row * a.ld_dv] : 0.f;
  }
  __syncthreads();
  ht_bwd_head(h, row0, a.B, s, h.n_xa ? a.dh_part + (long)tw * a.B * a.Ht : nullptr, a.Ht);
}

__global__ __launch_bounds__(256) void q_bwd_trunk_kernel(QFusedArgs a) {
  __shared__ HtLds s;
  ht_bwd_head(*a.bwd_tr, blockIdx.x * HT_RB, a.B, s, nullptr, 0, a.dh_part, a.D + 1, (long)a.B * a.Ht, a.trunk_scale);
}
#endif
#endif  // GRL_HEADS_TYPES_ONLY

}  // namespace grl
