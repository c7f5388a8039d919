// elem_kernels.h -- the HBM-/latency-bound pieces of the SAC update: replay gather + VecNormalize
// normalisation, squashed-Gaussian sampling and its backward, loss/out-gradient kernel, split-slab
// reduction, TF-style Adam fused with the Polyak target update, replay ingest, device RNG.
//
// Math follows SURVEY.md Appendix A (restating stable-baselines 2.10.1 SAC as wired by
// /root/reference/manipulation_main/training/sb_helper.py:104-128).
#pragma once
#ifdef GRL_HOSTEMU
#include "hostemu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include "../../include/grl.h"   // GRL_MAX_LAYERS

namespace grl {

// ------------------------------------------------------------------------------------------------
// device scalars shared by the kernels of one update (lives in the state arena)
struct DevScalars {
  float beta1_power, beta2_power;   // TF AdamOptimizer non-slot variables (start at beta1, beta2)
  float adam_alpha;                 // lr * sqrt(1-b2p)/(1-b1p) of the current step
  uint32_t rng_used;                // set by a gather that drew from the device RNG; the loss reduction then advances rng_step
  // metrics of the last update
  float policy_loss, qf1_loss, qf2_loss, value_loss, ent_loss, ent_coef, entropy, mean_qf1, mean_v;
  float lr;                         // learning rate of the current update (grl_set_learning_rate; starts at cfg.lr)
  // Philox counter of the minibatch whose IMAGES the head launch gathers ahead (plan_sac "gather_ride"): set to rng_step + 1 by
  // the gather that opens a multi-update call, advanced by the reduction launch that ends each update -- a word of its own
  // because the head launch advances rng_step while its riders draw
  uint64_t rng_img;
  uint64_t rng_step;                // Philox counter (one per drawn minibatch)
  int64_t replay_size;              // transitions currently stored
};

// one float written in stream order (grl_set_learning_rate: the step size read by captured graphs)
#ifndef GRL_ELEM_TYPES_ONLY
__global__ void set_f32_kernel(float* p, float v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *p = v;
}
#endif

// ------------------------------------------------------------------------------------------------
// device RNG (Philox4x32-10): replay indices uniform in [0, size) and standard normals
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// ------------------------------------------------------------------------------------------------
// replay gather + VecNormalize.normalize_obs / normalize_reward at sample time (A.1 step 3) +
// observation_input /255 scaling (A.1 step 4).  float64 arithmetic as in the NumPy reference path,
// rounded to float32 once, then the float32 division by 255 TF performs.
struct GatherArgs {
  const int64_t* idx;
  int B, img_elems, n_direct, act_dim;
  const float* rp_obs; const float* rp_next;       // [cap, img_elems]
  const float* rp_dobs; const float* rp_dnext;     // [cap, n_direct]
  const float* rp_act; const float* rp_rew; const float* rp_done;
  const double* mean; const double* stdv;          // [img_elems] (std = sqrt(var + eps))
  const double* dmean; const double* dstd;         // [n_direct]
  const double* ret_std;                           // [1]
  int normalize;                                   // observations (VecNormalize norm_obs)
  int normalize_rew;                               // rewards (VecNormalize norm_reward)
  double clip_obs, clip_rew;
  float scale_div;                                 // 255 for CNN policies, 1 for MLP
  float* x_obs; float* x_obs2; float* x_next;      // destinations, row stride ldx
  int ldx;
  float* d_obs0; float* d_obs1; float* d_next;     // direct-feature destinations, row stride ldd
  int ldd;
  float* act_out; int ld_act;
  float* act_out2; int ld_act2;                    // optional second copy (row-padded for 16-byte loads)
  float* rew_out; float* done_out;
  // device-RNG mode (use_rng): the replay index of row b and its A standard normals are drawn here
  // (Philox4x32-10 keyed by seed, counter = (rng_step, b, stream)); the last workgroup to finish
  // advances rng_step.  Explicit mode reads idx (and the caller has staged eps).
  int use_rng; DevScalars* sc; uint64_t seed;
  int64_t* idx_w; float* eps_w; int n_eps;
  int rgb_u8; // RGB-D ring with byte colours: per stored observation [hw packed dwords R|G<<8|B<<16][hw float32 depth]
  int vec4;   // img_elems, ldx multiples of 4 and 16-byte aligned rows: a thread moves 4 elements (grid.x = ceil(img_elems / 1024))
  // adam_tick: this launch opens an update -- one thread fixes the Adam step size of the update from the
  // beta powers and advances them (TF ApplyAdam: alpha_t = lr * sqrt(1 - beta2_power) / (1 - beta1_power)).
  // Done here, a whole launch chain before the first consumer, so that the fused reduce + Adam launch
  // can read it from any workgroup without ordering against the workgroup that produces the losses.
  int adam_tick;
  // prefetching launches (the gather of update t+1 riding on the last launch of update t, reduce_slabs_gather_kernel):
  // rng_ahead = 1 draws with counter rng_step + 1 (the counter still holds update t's value while that launch runs);
  // quiet = 1: neither mark rng_used nor touch the Adam step size -- the update it prepares has not started yet
  int rng_ahead, quiet;
  // rows > 1 (vec4 only, rows divides B): a workgroup owns one 1024-element tile of `rows` minibatch rows (of obs OR next_obs):
  // the float64 statistics of its element positions are fetched once, the `rows` replay reads are all in flight before the
  // first is used (gather_norm_rows_body); grid = tiles per row x (2 B / rows), linearised (gather_blocks)
  int rows;
  // parts: 0 = everything (default); 1 = the image tiles only; 2 = the per-row extras only (direct features, action, reward,
  // done, index, standard normals -- launched with gx = 1).  Multi-update calls with double-buffered images (plan_sac
  // "gather_ride") gather the IMAGES of update t+1 inside update t's head launch and leave the extras, which update t's
  // backward pass still reads, to the reduction launch that ends update t.
  // img_ctr = 1: draw with the counter DevScalars.rng_img (the riders of the head launch); set_img = 1: leave rng_step + 1 there
  int parts, img_ctr, set_img;
};
#define GATHER_RIDE_ROWS 16       /* row instances per rider workgroup of the head launch (gather_images_rider) */
#define GATHER_ROWS_U8 4          /* rows per workgroup of the grouped form, RGB-D ring with byte colours (GRL_TUNE gather_rows) */
#define GATHER_ROWS_F32 1         /* float32 rings: one row per workgroup measured fastest (profiles/r06_sweep_gather_rows.txt) */
// workgroups of a gather launch with gx tiles per row
static inline int gather_blocks(const GatherArgs& a, int gx) { return a.rows > 1 ? gx * (2 * a.B / a.rows) : gx * a.B * 2; }

// y / scale_div as TF's float32 division gives it.  For the one divisor the image paths use, 255, the quotient is formed
// without the division sequence: q = y r, r = RN(1 / 255), one FMA refinement step q + fma(-255, q, y) r -- bit for bit the
// IEEE quotient for EVERY float |y| <= 2^22, subnormal quotients included (checked exhaustively on the host: 1.26e9 values,
// no mismatch; gfx950 keeps float32 subnormals): three full-rate instructions for about ten.
__device__ __forceinline__ float scale_elem(float y, float scale_div) {
#pragma clang fp contract(off)
  if (scale_div == 255.f && __builtin_fabsf(y) <= 4194304.f) {
    const float r = 1.f / 255.f;
    const float q = y * r;
    return __builtin_fmaf(__builtin_fmaf(-255.f, q, y), r, q);
  }
  return scale_div != 1.f ? y / scale_div : y;
}

__device__ __forceinline__ float norm_elem(float x, double mu, double sd, int normalize, double clip,
                                           float scale_div) {
  float y = x;
  if (normalize) {
    double z = ((double)x - mu) / sd;
    z = __builtin_fmax(__builtin_fmin(z, clip), -clip);      // (np.clip; z is never a NaN: sd >= sqrt(eps) > 0)
    y = (float)z;
  }
  return scale_elem(y, scale_div);
}

// The same value with the float64 division (x - mu) / sd replaced by two FMA refinement steps on (x - mu) * r, r = RN(1 / sd):
// q0 = d r; e = fma(-sd, q, d); q = fma(e, r, q), twice.  The first step leaves q within an ulp of d / sd, and with the
// CORRECTLY ROUNDED reciprocal the second then yields the correctly rounded quotient (Markstein's theorem; the residual
// e is exact in an FMA) -- the bits of the IEEE division, which a workgroup that owns several rows of the same element
// positions (gather_norm_rows_body) pays once per position instead of once per element: on gfx950 the division expands to ~13
// float64 instructions, three of them quarter rate, against these six (sub, mul, four FMAs).  400 000 000 random and adversarial
// (x, mu, sd) on the host: no mismatch even after ONE step.  Ranges: d, sd, r normal numbers (sd >= sqrt(norm_eps)).
__device__ __forceinline__ float norm_elem_rcp(float x, double mu, double sd, double r, double clip, float scale_div) {
#pragma clang fp contract(off)
  const double d = (double)x - mu;
  double q = d * r;
  double e = __builtin_fma(-sd, q, d);
  q = __builtin_fma(e, r, q);
  e = __builtin_fma(-sd, q, d);
  q = __builtin_fma(e, r, q);
  q = __builtin_fmax(__builtin_fmin(q, clip), -clip);
  return scale_elem((float)q, scale_div);
}

// The same for the shape every CNN policy has -- normalised, divisor 255, clip <= 2^22 (so |y| is inside scale_elem's exact range
// by construction) -- WITHOUT the per-element branch to the general division: that branch ends the basic block after every
// element, so the compiler cannot interleave the float64 chains of the independent elements a thread holds (10 dependent float64
// instructions each).  Measured on the riders of the head launch, one wave per SIMD: ~350 cycles per element with the branch.
__device__ __forceinline__ bool norm_fast255(int normalize, double clip, float scale_div) {
  return normalize && scale_div == 255.f && clip >= 0.0 && clip <= 4194304.0;
}
__device__ __forceinline__ float norm_elem_div255(float x, double mu, double sd, double clip) {      // (the division itself)
#pragma clang fp contract(off)
  double q = ((double)x - mu) / sd;
  q = __builtin_fmax(__builtin_fmin(q, clip), -clip);
  const float y = (float)q, r255 = 1.f / 255.f;
  const float t = y * r255;
  return __builtin_fmaf(__builtin_fmaf(-255.f, t, y), r255, t);
}
__device__ __forceinline__ float norm_elem_rcp255(float x, double mu, double sd, double r, double clip) {
#pragma clang fp contract(off)
  const double d = (double)x - mu;
  double q = d * r;
  double e = __builtin_fma(-sd, q, d);
  q = __builtin_fma(e, r, q);
  e = __builtin_fma(-sd, q, d);
  q = __builtin_fma(e, r, q);
  q = __builtin_fmax(__builtin_fmin(q, clip), -clip);
  const float y = (float)q, r255 = 1.f / 255.f;
  const float t = y * r255;
  return __builtin_fmaf(__builtin_fmaf(-255.f, t, y), r255, t);
}

// One minibatch row (observation, next observation, action, reward, done) by the 256 threads of a workgroup: what the
// blocks (*, b, 0..1) of gather_norm_kernel do for row b, element by element -- for launches that already own a row
// (the prioritised sampler knows the replay index of its row and gathers it on the spot, per_kernels.h).
__device__ __forceinline__ void gather_row_device(const GatherArgs& a, int b, int64_t src) {
  const int t = threadIdx.x;
  for (int which = 0; which < 2; ++which) {
    const float* rp = which ? a.rp_next : a.rp_obs;
    float* dst = which ? a.x_next : a.x_obs;
    for (int e = t; e < a.img_elems; e += 256) {
      const float y = norm_elem(rp[src * a.img_elems + e], a.normalize ? a.mean[e] : 0.0, a.normalize ? a.stdv[e] : 1.0,
                                a.normalize, a.clip_obs, a.scale_div);
      dst[(long)b * a.ldx + e] = y;
      if (!which && a.x_obs2) a.x_obs2[(long)b * a.ldx + e] = y;
    }
    if (t < a.n_direct) {
      const float* rd = which ? a.rp_dnext : a.rp_dobs;
      const float y = norm_elem(rd[src * a.n_direct + t], a.normalize ? a.dmean[t] : 0.0, a.normalize ? a.dstd[t] : 1.0,
                                a.normalize, a.clip_obs, a.scale_div);
      if (which) a.d_next[(long)b * a.ldd + t] = y;
      else { a.d_obs0[(long)b * a.ldd + t] = y; a.d_obs1[(long)b * a.ldd + t] = y; }
    }
  }
  if (t < a.act_dim) {
    const float av = a.rp_act[src * a.act_dim + t];
    a.act_out[(long)b * a.ld_act + t] = av;
    if (a.act_out2) a.act_out2[(long)b * a.ld_act2 + t] = av;
  }
  if (t == 64) {
    float r = a.rp_rew[src];
    if (a.normalize_rew) {
      double z = (double)r / a.ret_std[0];
      z = z < -a.clip_rew ? -a.clip_rew : (z > a.clip_rew ? a.clip_rew : z);
      r = (float)z;
    }
    a.rew_out[b] = r;
  }
  if (t == 65) a.done_out[b] = a.rp_done[src];
}
// the Adam step size of the update that starts (TF ApplyAdam), and the beta powers for the next one
__device__ __forceinline__ void adam_tick_device(DevScalars* sc) {
  sc->adam_alpha = sc->lr * sqrtf(1.f - sc->beta2_power) / (1.f - sc->beta1_power);
  sc->beta1_power *= 0.9f;
  sc->beta2_power *= 0.999f;
}

// block (bx, b, which) of the gather grid: elements [bx * per_block, ...) of row b of obs (which 0) / next_obs (1)
__device__ __forceinline__ void gather_norm_body(const GatherArgs& a, const int bx, const int b, const int which) {
  int64_t src;
  uint64_t step = 0;
  if (a.use_rng) {
    step = a.img_ctr ? a.sc->rng_img : a.sc->rng_step + (uint64_t)a.rng_ahead;
    const int64_t size = a.sc->replay_size;
    uint32_t c[4] = {(uint32_t)step, (uint32_t)(step >> 32), (uint32_t)b, 0u};
    philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    const uint64_t u = ((uint64_t)c[0] << 32) | c[1];
    src = size > 0 ? (int64_t)__umul64hi(u, (uint64_t)size) : 0;
  } else {
    src = a.idx[b];
  }
#ifndef GRL_HOSTEMU
  if (a.parts == 2) {
  } else if (a.vec4) {
    typedef float gn_f4 __attribute__((ext_vector_type(4)));
    typedef double gn_d4 __attribute__((ext_vector_type(4)));
    const int e4 = (bx * 256 + threadIdx.x) * 4;
    if (e4 < a.img_elems) {
      const float* rp = which ? a.rp_next : a.rp_obs;
      gn_f4 x;
      if (a.rgb_u8) {            // one pixel per thread: colours from the packed dword, depth from the float plane
        const int hwp = a.img_elems >> 2, px = e4 >> 2;
        const float* ob = rp + src * (2 * hwp);
        const uint32_t w = ((const uint32_t*)ob)[px];
        x = gn_f4{(float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), ob[hwp + px]};
      } else {
        x = *(const gn_f4*)(rp + src * a.img_elems + e4);
      }
      gn_d4 mu = {0.0, 0.0, 0.0, 0.0}, sd = {1.0, 1.0, 1.0, 1.0};
      if (a.normalize) { mu = *(const gn_d4*)(a.mean + e4); sd = *(const gn_d4*)(a.stdv + e4); }
      gn_f4 y;
      if (norm_fast255(a.normalize, a.clip_obs, a.scale_div)) {      // (uniform; four independent chains, no branch between them)
        y.x = norm_elem_div255(x.x, mu.x, sd.x, a.clip_obs);
        y.y = norm_elem_div255(x.y, mu.y, sd.y, a.clip_obs);
        y.z = norm_elem_div255(x.z, mu.z, sd.z, a.clip_obs);
        y.w = norm_elem_div255(x.w, mu.w, sd.w, a.clip_obs);
      } else {
      y.x = norm_elem(x.x, mu.x, sd.x, a.normalize, a.clip_obs, a.scale_div);
      y.y = norm_elem(x.y, mu.y, sd.y, a.normalize, a.clip_obs, a.scale_div);
      y.z = norm_elem(x.z, mu.z, sd.z, a.normalize, a.clip_obs, a.scale_div);
      y.w = norm_elem(x.w, mu.w, sd.w, a.normalize, a.clip_obs, a.scale_div);
      }
      if (which) {
        *(gn_f4*)(a.x_next + (long)b * a.ldx + e4) = y;
      } else {
        *(gn_f4*)(a.x_obs + (long)b * a.ldx + e4) = y;
        if (a.x_obs2) *(gn_f4*)(a.x_obs2 + (long)b * a.ldx + e4) = y;
      }
    }
  } else
#endif
  {
  const int e = bx * 256 + threadIdx.x;
  if (e < a.img_elems && a.parts != 2) {
    const float* rp = which ? a.rp_next : a.rp_obs;
    float x;
    if (a.rgb_u8) {
      const int hwp = a.img_elems >> 2, px = e >> 2, ch = e & 3;
      const float* ob = rp + src * (2 * hwp);
      x = ch == 3 ? ob[hwp + px] : (float)((((const uint32_t*)ob)[px] >> (8 * ch)) & 255u);
    } else {
      x = rp[src * a.img_elems + e];
    }
    const float y = norm_elem(x, a.normalize ? a.mean[e] : 0.0, a.normalize ? a.stdv[e] : 1.0,
                              a.normalize, a.clip_obs, a.scale_div);
    if (which) {
      a.x_next[(long)b * a.ldx + e] = y;
    } else {
      a.x_obs[(long)b * a.ldx + e] = y;
      if (a.x_obs2) a.x_obs2[(long)b * a.ldx + e] = y;
    }
  }
  }
  if (bx == 0 && a.parts != 1) {
    const int t = threadIdx.x;
    if (t < a.n_direct) {
      const float* rp = which ? a.rp_dnext : a.rp_dobs;
      const float x = rp[src * a.n_direct + t];
      const float y = norm_elem(x, a.normalize ? a.dmean[t] : 0.0, a.normalize ? a.dstd[t] : 1.0,
                                a.normalize, a.clip_obs, a.scale_div);
      if (which) {
        a.d_next[(long)b * a.ldd + t] = y;
      } else {
        a.d_obs0[(long)b * a.ldd + t] = y;
        a.d_obs1[(long)b * a.ldd + t] = y;
      }
    }
    if (which == 0) {
      if (t < a.act_dim) {
        const float av = a.rp_act[src * a.act_dim + t];
        a.act_out[(long)b * a.ld_act + t] = av;
        if (a.act_out2) a.act_out2[(long)b * a.ld_act2 + t] = av;
      }
      if (t == 64) {
        float r = a.rp_rew[src];
        if (a.normalize_rew) {
          double z = (double)r / a.ret_std[0];
          z = z < -a.clip_rew ? -a.clip_rew : (z > a.clip_rew ? a.clip_rew : z);
          r = (float)z;
        }
        a.rew_out[b] = r;
      }
      if (t == 65) a.done_out[b] = a.rp_done[src];
      if (a.use_rng) {
        if (t == 66) a.idx_w[b] = src;
        if (t >= 128 && 2 * (t - 128) < a.n_eps) {   // Box-Muller pairs, as rng_kernel
          const int j0 = 2 * (t - 128);
          uint32_t d[4] = {(uint32_t)step, (uint32_t)(step >> 32), (uint32_t)b, (uint32_t)(1 + j0)};
          philox4x32_10(d, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
          const float u1 = ((float)(d[0] >> 8) + 0.5f) * (1.f / 16777216.f);
          const float u2 = ((float)(d[1] >> 8) + 0.5f) * (1.f / 16777216.f);
          const float rad = sqrtf(-2.f * logf(u1));
          float sn, cs;
          sincosf(6.283185307179586f * u2, &sn, &cs);
          a.eps_w[b * a.n_eps + j0] = rad * cs;
          if (j0 + 1 < a.n_eps) a.eps_w[b * a.n_eps + j0 + 1] = rad * sn;
        }
      }
    }
  }
  // rng_step itself is advanced by the (single-workgroup) loss reduction later in the update: a counter
  // bumped by the last of these 8192 workgroups would serialise 8192 same-address atomics (~100 us)
  if (bx == 0 && b == 0 && which == 0 && threadIdx.x == 0 && !a.quiet && a.parts != 1) {
    if (a.use_rng) a.sc->rng_used = 1u;
    if (a.use_rng && a.set_img) a.sc->rng_img = step + 1;
    if (a.adam_tick) adam_tick_device(a.sc);
  }
}
// the per-row duties of tile 0 (direct features, action, reward, done flag, the row's index and standard normals in device-RNG
// mode): what gather_norm_body does under `bx == 0`, for the grouped form
__device__ __forceinline__ void gather_row_extras(const GatherArgs& a, const int b, const int which, const int64_t src, const uint64_t step) {
  const int t = threadIdx.x;
  if (t < a.n_direct) {
    const float* rp = which ? a.rp_dnext : a.rp_dobs;
    const float y = norm_elem(rp[src * a.n_direct + t], a.normalize ? a.dmean[t] : 0.0, a.normalize ? a.dstd[t] : 1.0,
                              a.normalize, a.clip_obs, a.scale_div);
    if (which) a.d_next[(long)b * a.ldd + t] = y;
    else { a.d_obs0[(long)b * a.ldd + t] = y; a.d_obs1[(long)b * a.ldd + t] = y; }
  }
  if (which) return;
  if (t < a.act_dim) {
    const float av = a.rp_act[src * a.act_dim + t];
    a.act_out[(long)b * a.ld_act + t] = av;
    if (a.act_out2) a.act_out2[(long)b * a.ld_act2 + t] = av;
  }
  if (t == 64) {
    float r = a.rp_rew[src];
    if (a.normalize_rew) {
      double z = (double)r / a.ret_std[0];
      z = z < -a.clip_rew ? -a.clip_rew : (z > a.clip_rew ? a.clip_rew : z);
      r = (float)z;
    }
    a.rew_out[b] = r;
  }
  if (t == 65) a.done_out[b] = a.rp_done[src];
  if (a.use_rng) {
    if (t == 66) a.idx_w[b] = src;
    if (t >= 128 && 2 * (t - 128) < a.n_eps) {   // Box-Muller pairs, as gather_norm_body
      const int j0 = 2 * (t - 128);
      uint32_t d[4] = {(uint32_t)step, (uint32_t)(step >> 32), (uint32_t)b, (uint32_t)(1 + j0)};
      philox4x32_10(d, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
      const float u1 = ((float)(d[0] >> 8) + 0.5f) * (1.f / 16777216.f);
      const float u2 = ((float)(d[1] >> 8) + 0.5f) * (1.f / 16777216.f);
      const float rad = sqrtf(-2.f * logf(u1));
      float sn, cs;
      sincosf(6.283185307179586f * u2, &sn, &cs);
      a.eps_w[b * a.n_eps + j0] = rad * cs;
      if (j0 + 1 < a.n_eps) a.eps_w[b * a.n_eps + j0 + 1] = rad * sn;
    }
  }
}

#ifndef GRL_HOSTEMU
// Grouped form (GatherArgs.rows = R > 1): workgroup (bx, grp) moves tile bx -- elements [1024 bx, 1024 bx + 1024) -- of the R
// row instances grp R .. grp R + R - 1 of the 2 B (obs rows first, then next_obs rows; R divides B, so a group is all obs or all
// next_obs).  Same arithmetic per element as gather_norm_body (norm_elem), hence the same bits.  What changes is the shape of
// the latency chain: one Philox draw per lane (lane r holds row r's index, the others read it with a lane broadcast), the
// statistics of the thread's four element positions loaded ONCE, and the R replay reads issued back to back before the first
// is consumed -- a workgroup lives about as long as a one-row workgroup does and there are R times fewer of them.
// BRANCH_FREE: take norm_elem_rcp255's loop when the arguments allow it.  It keeps all R results live (the point: R x 4 chains
// in flight), which the kernels built for 8 waves per SIMD cannot afford (reduce_slabs_gather_kernel: 36 bytes of scratch per
// lane and 4 660 against 4 740 updates/s on the RGB-D ring): those keep the one-element-at-a-time loop.
template <int R, bool BRANCH_FREE = false>
__device__ __forceinline__ void gather_norm_rows_body(const GatherArgs& a, const int bx, const int grp) {
  typedef float gn_f4 __attribute__((ext_vector_type(4)));
  typedef double gn_d4 __attribute__((ext_vector_type(4)));
  const int i0 = grp * R, which = i0 >= a.B ? 1 : 0, b0 = i0 - which * a.B;
  const int lane = (int)(threadIdx.x & 63u);
  uint64_t step = 0;
  int64_t mine;
  if (a.use_rng) {
    step = a.img_ctr ? a.sc->rng_img : a.sc->rng_step + (uint64_t)a.rng_ahead;
    const int64_t size = a.sc->replay_size;
    uint32_t c[4] = {(uint32_t)step, (uint32_t)(step >> 32), (uint32_t)(b0 + (lane % R)), 0u};
    philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    const uint64_t u = ((uint64_t)c[0] << 32) | c[1];
    mine = size > 0 ? (int64_t)__umul64hi(u, (uint64_t)size) : 0;
  } else {
    mine = a.idx[b0 + (lane % R)];
  }
  int64_t src[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, r);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)mine >> 32), r);
    src[r] = (int64_t)(((uint64_t)hi << 32) | lo);
  }
  const int e4 = (bx * 256 + (int)threadIdx.x) * 4;
  if (e4 < a.img_elems && a.parts != 2) {
    const float* rp = which ? a.rp_next : a.rp_obs;
    gn_f4 x[R];
    if (a.rgb_u8) {            // one pixel per thread: colours from the packed dword, depth from the float plane
      const int hwp = a.img_elems >> 2, px = e4 >> 2;
      uint32_t w[R];
      float d[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float* ob = rp + src[r] * (2 * hwp);
        w[r] = ((const uint32_t*)ob)[px];
        d[r] = ob[hwp + px];
      }
#pragma unroll
      for (int r = 0; r < R; ++r) x[r] = gn_f4{(float)(w[r] & 255u), (float)((w[r] >> 8) & 255u), (float)((w[r] >> 16) & 255u), d[r]};
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) x[r] = *(const gn_f4*)(rp + src[r] * a.img_elems + e4);
    }
    gn_d4 mu = {0.0, 0.0, 0.0, 0.0}, sd = {1.0, 1.0, 1.0, 1.0}, rc = {1.0, 1.0, 1.0, 1.0};
    if (a.normalize) {
      mu = *(const gn_d4*)(a.mean + e4); sd = *(const gn_d4*)(a.stdv + e4);
      rc.x = 1.0 / sd.x; rc.y = 1.0 / sd.y; rc.z = 1.0 / sd.z; rc.w = 1.0 / sd.w;     // IEEE divisions: correctly rounded reciprocals
    }
    const bool fast = BRANCH_FREE && norm_fast255(a.normalize, a.clip_obs, a.scale_div);      // (uniform: one branch around the loop, none inside)
    if (fast) {
      gn_f4 y[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        y[r].x = norm_elem_rcp255(x[r].x, mu.x, sd.x, rc.x, a.clip_obs);
        y[r].y = norm_elem_rcp255(x[r].y, mu.y, sd.y, rc.y, a.clip_obs);
        y[r].z = norm_elem_rcp255(x[r].z, mu.z, sd.z, rc.z, a.clip_obs);
        y[r].w = norm_elem_rcp255(x[r].w, mu.w, sd.w, rc.w, a.clip_obs);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const long o = (long)(b0 + r) * a.ldx + e4;
        if (which) {
          *(gn_f4*)(a.x_next + o) = y[r];
        } else {
          *(gn_f4*)(a.x_obs + o) = y[r];
          if (a.x_obs2) *(gn_f4*)(a.x_obs2 + o) = y[r];
        }
      }
    } else
#pragma unroll
    for (int r = 0; r < R; ++r) {
      gn_f4 y;
      if (a.normalize) {
        y.x = norm_elem_rcp(x[r].x, mu.x, sd.x, rc.x, a.clip_obs, a.scale_div);
        y.y = norm_elem_rcp(x[r].y, mu.y, sd.y, rc.y, a.clip_obs, a.scale_div);
        y.z = norm_elem_rcp(x[r].z, mu.z, sd.z, rc.z, a.clip_obs, a.scale_div);
        y.w = norm_elem_rcp(x[r].w, mu.w, sd.w, rc.w, a.clip_obs, a.scale_div);
      } else {
        y.x = norm_elem(x[r].x, 0.0, 1.0, 0, a.clip_obs, a.scale_div);
        y.y = norm_elem(x[r].y, 0.0, 1.0, 0, a.clip_obs, a.scale_div);
        y.z = norm_elem(x[r].z, 0.0, 1.0, 0, a.clip_obs, a.scale_div);
        y.w = norm_elem(x[r].w, 0.0, 1.0, 0, a.clip_obs, a.scale_div);
      }
      const long o = (long)(b0 + r) * a.ldx + e4;
      if (which) {
        *(gn_f4*)(a.x_next + o) = y;
      } else {
        *(gn_f4*)(a.x_obs + o) = y;
        if (a.x_obs2) *(gn_f4*)(a.x_obs2 + o) = y;
      }
    }
  }
  if (bx == 0 && a.parts != 1) {
#pragma unroll 1
    for (int r = 0; r < R; ++r) gather_row_extras(a, b0 + r, which, src[r], step);
    if (grp == 0 && threadIdx.x == 0 && !a.quiet) {
      if (a.use_rng) a.sc->rng_used = 1u;
      if (a.use_rng && a.set_img) a.sc->rng_img = step + 1;
      if (a.adam_tick) adam_tick_device(a.sc);
    }
  }
}
#endif

// Workgroup r of the image gather that rides on the head launch (heads_mfma.h; GatherArgs.parts == 1, vec4, B a multiple of
// GATHER_RIDE_ROWS): tile r % gx of the GATHER_RIDE_ROWS row instances of group r / gx -- the head kernel is compiled for one
// wave per SIMD (245 registers), so a rider workgroup is alone on its CU and keeps 16 replay reads in flight per lane instead of
// relying on neighbours to hide its latency.  Same arithmetic per element as every other form of the gather.
static inline int gather_rider_blocks(const GatherArgs& a, int gx) { return gx * (2 * a.B / a.rows); }
__device__ __forceinline__ void gather_images_rider(const GatherArgs& a, const int gx, const int r) {
#ifndef GRL_HOSTEMU
  if (a.rows == 16) gather_norm_rows_body<16, true>(a, r % gx, r / gx);
  else if (a.rows == 8) gather_norm_rows_body<8, true>(a, r % gx, r / gx);
  else gather_norm_rows_body<4, true>(a, r % gx, r / gx);
#else
  for (int k = 0; k < a.rows; ++k) {
    const int i = (r / gx) * a.rows + k;
    gather_norm_body(a, r % gx, i % a.B, i / a.B);
  }
#endif
}

// workgroup r of a linearised gather grid (gather_blocks) with gx tiles per row.  GROUPED is a compile-time choice of the
// LAUNCH (the plan picks the kernel by GatherArgs.rows): a kernel that could take either form is allocated the registers of
// the larger one -- 69 instead of 64 for the reduction + gather launch, one wave per SIMD less for the one-row form
template <bool GROUPED>
__device__ __forceinline__ void gather_norm_dispatch(const GatherArgs& a, const int gx, const int r) {
#ifndef GRL_HOSTEMU
  if (GROUPED) {
    const int bx = r % gx, grp = r / gx;
    // (two instantiations only: the launch that carries this body is compiled for the largest of them -- with R = 16 in the
    // list reduce_slabs_gather_kernel went from 64 to 95 registers and spilled)
    if (a.rows == 2) gather_norm_rows_body<2>(a, bx, grp);
    else gather_norm_rows_body<4>(a, bx, grp);
    return;
  }
#endif
  gather_norm_body(a, r % gx, (r / gx) % a.B, r / (gx * a.B));
}

#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void gather_norm_kernel(GatherArgs a) {
  gather_norm_body(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}
// linearised grid (gather_blocks): the grouped form
// (8 waves per SIMD = 64 registers: the 2 048 workgroups of an RGB-D minibatch -- 16 tiles x 128 row groups -- then fit the
//  chip's 2 048 slots in ONE round; at 69 registers, 7 per SIMD, the last 256 ran as a round of their own)
__global__ __launch_bounds__(256, 8) void gather_norm_lin_kernel(GatherArgs a, int gx) {
  gather_norm_dispatch<true>(a, gx, (int)blockIdx.x);
}
#endif

// ------------------------------------------------------------------------------------------------
// replay ingest: env-layout observation [n, HW, Cobs] (or [n, D]) -> image block [HW*Cimg] + direct
struct IngestArgs {
  const float* obs; const float* next_obs; const float* act; const float* rew; const float* done;
  int n, hw, c_obs, c_img, n_direct, act_dim, vec_dim;   // vec_dim > 0: MLP vector observation
  int64_t pos, cap;
  float* rp_obs; float* rp_next; float* rp_dobs; float* rp_dnext; float* rp_act; float* rp_rew;
  float* rp_done;
  int rgb_u8;   // RGB-D ring with byte colours (grl_config.replay_rgb_u8)
  // grl_replay_add_observed: transition k's next observation is row next_row[k] of next_alt when next_row[k] >= 0 (the
  // terminal observation of an episode that ended: the observed row already holds the first one of the next episode)
  const int* next_row; const float* next_alt;
  int64_t* size_out; int64_t new_size;   // the ring's fill level after this call (DevScalars.replay_size), written here
};

#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void ingest_kernel(IngestArgs a) {
  const int k = blockIdx.y;
  const int which = blockIdx.z;
  const int64_t dst = (a.pos + k) % a.cap;
  const long row_elems = a.vec_dim > 0 ? a.vec_dim : (long)a.hw * a.c_obs;
  const int alt = (which && a.next_row) ? a.next_row[k] : -1;
  // (rebased so that row k below addresses the chosen row)
  const float* src = alt >= 0 ? a.next_alt + ((long)alt - k) * row_elems : (which ? a.next_obs : a.obs);
  float* img = which ? a.rp_next : a.rp_obs;
  float* dir = which ? a.rp_dnext : a.rp_dobs;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (a.vec_dim > 0) {
    if (e < a.vec_dim) img[dst * a.vec_dim + e] = src[(long)k * a.vec_dim + e];
  } else {
    const int img_elems = a.hw * a.c_img;
    if (a.rgb_u8) {
      if (e < a.hw) {           // one pixel per thread: R, G, B -> one dword (round to nearest, clamped), depth as is
        const float* p = src + ((long)k * a.hw + e) * a.c_obs;
        uint32_t w = 0;
        for (int ch = 0; ch < 3; ++ch) w |= (uint32_t)fminf(fmaxf(rintf(p[ch]), 0.f), 255.f) << (8 * ch);
        float* ob = img + dst * (2 * a.hw);
        ((uint32_t*)ob)[e] = w;
        ob[a.hw + e] = p[3];
      }
    } else if (e < img_elems) {
      const int px = e / a.c_img, c = e - px * a.c_img;
      img[dst * img_elems + e] = src[((long)k * a.hw + px) * a.c_obs + c];
    }
    if (blockIdx.x == 0 && threadIdx.x < a.n_direct)   // flatten(x[..., -1])[:, :n]  (custom_obs_policy.py:28-30)
      dir[dst * a.n_direct + threadIdx.x] =
          src[((long)k * a.hw + threadIdx.x) * a.c_obs + (a.c_obs - 1)];
  }
  if (blockIdx.x == 0 && which == 0) {
    if (threadIdx.x < a.act_dim) a.rp_act[dst * a.act_dim + threadIdx.x] = a.act[(long)k * a.act_dim + threadIdx.x];
    if (threadIdx.x == 64) a.rp_rew[dst] = a.rew[k];
    if (threadIdx.x == 65) a.rp_done[dst] = a.done[k];
    if (threadIdx.x == 66 && k == 0) *a.size_out = a.new_size;
  }
}
#endif

// observation of grl_act: already VecNormalize-d by the env wrapper; only split + /255
struct ActIngestArgs {
  const float* obs; int n, hw, c_obs, c_img, n_direct, vec_dim;
  float scale_div;
  float* x; int ldx;
  float* d; int ldd;
  // normalize = 1: obs are RAW observations, VecNormalize.normalize_obs is applied here with the statistics kept on
  // the device (grl_norm_update) -- the same float64 expression as at replay-sample time (norm_elem)
  int normalize; double clip_obs;
  const double* mean; const double* stdv; const double* dmean; const double* dstd;
};

#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void act_ingest_kernel(ActIngestArgs a) {
  const int k = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (a.vec_dim > 0) {
    if (e < a.vec_dim) {
      const float x = a.obs[(long)k * a.vec_dim + e];
      a.x[(long)k * a.ldx + e] = a.normalize ? norm_elem(x, a.mean[e], a.stdv[e], 1, a.clip_obs, a.scale_div) : x;
    }
    return;
  }
  const int img_elems = a.hw * a.c_img;
  if (e < img_elems) {
    const int px = e / a.c_img, c = e - px * a.c_img;
    const float x = a.obs[((long)k * a.hw + px) * a.c_obs + c];
    a.x[(long)k * a.ldx + e] = a.normalize ? norm_elem(x, a.mean[e], a.stdv[e], 1, a.clip_obs, a.scale_div) : x / a.scale_div;
  }
  if (blockIdx.x == 0 && threadIdx.x < a.n_direct) {
    const int q = threadIdx.x;
    const float x = a.obs[((long)k * a.hw + q) * a.c_obs + (a.c_obs - 1)];
    a.d[(long)k * a.ldd + q] = a.normalize ? norm_elem(x, a.dmean[q], a.dstd[q], 1, a.clip_obs, a.scale_div) : x / a.scale_div;
  }
}
#endif

// ------------------------------------------------------------------------------------------------
// VecNormalize running statistics on the device (stable-baselines RunningMeanStd.update + update_from_moments, reached
// from VecNormalize.step_wait at every env step; wrapper created at sb_helper.py:117-119).  One thread per element of
// the env-layout observation:
//   batch_mean = np.mean(obs, axis=0), batch_var = np.var(obs, axis=0)   -- FLOAT32 like the observations DummyVecEnv
//       hands over (row-by-row accumulation, sum / n; mean of squared deviations from that mean)
//   delta = batch_mean - mean;  tot = count + n;  new_mean = mean + delta * n / tot
//   m2 = var * count + batch_var * n + delta**2 * count * n / (count + n);  new_var = m2 / (count + n)     -- float64,
//       except `batch_var * n`, which NumPy evaluates in float32 (float32 array times Python int)
// and the derived arrays the sampling / acting kernels read (mean, sqrt(var + eps) in their image / direct layout).
// count is double buffered (count[parity] read by every thread, count[parity ^ 1] written by one).
struct NormUpdateArgs {
  const float* obs; int n, elems;
  double* mean; double* var; double* count; int parity;
  double eps;
  int hw, c_obs, c_img, n_direct, vec;
  double* s_mean; double* s_std; double* s_dmean; double* s_dstd;
};
// float32 batch moments of element e over the n staged observations, as np.mean / np.var(axis=0) form them
__device__ __forceinline__ void norm_batch_moments(const NormUpdateArgs& a, int e, float& bm, float& bv) {
#pragma clang fp contract(off)
  float acc = a.obs[e];
  for (int i = 1; i < a.n; ++i) acc += a.obs[(long)i * a.elems + e];
  bm = acc / (float)a.n;
  const float d0 = a.obs[e] - bm;
  float acc2 = d0 * d0;
  for (int i = 1; i < a.n; ++i) {
    const float d = a.obs[(long)i * a.elems + e] - bm;
    acc2 += d * d;
  }
  bv = acc2 / (float)a.n;
}
// RunningMeanStd.update_from_moments on one element.  f32_product: `batch_var * batch_count` as NumPy evaluates it for a
// float32 array and a Python int (the single-process path); data parallel, the moments were widened to float64 before
// they travelled (grasp_rl.parallel.gather_moments) and the product is a float64 one.
__device__ __forceinline__ void norm_chan_merge(double& mean, double& var, double& cnt, float bm, float bv, int n, bool f32_product) {
#pragma clang fp contract(off)
  const double bc = (double)n;
  const double delta = (double)bm - mean;
  const double tot = cnt + bc;
  const double new_mean = mean + delta * bc / tot;
  const double m_a = var * cnt;
  const double m_b = f32_product ? (double)(bv * (float)n) : (double)bv * bc;
  const double m2 = m_a + m_b + delta * delta * cnt * bc / (cnt + bc);
  var = m2 / (cnt + bc);
  mean = new_mean;
  cnt = tot;
}
// running statistics of element e and the derived arrays the sampling / acting kernels read
__device__ __forceinline__ void norm_store(const NormUpdateArgs& a, int e, double mean, double var) {
  a.mean[e] = mean;
  a.var[e] = var;
  const double sd = sqrt(var + a.eps);
  if (a.vec) { a.s_mean[e] = mean; a.s_std[e] = sd; }
  else {
    const int px = e / a.c_obs, ch = e - px * a.c_obs;
    if (ch < a.c_img) { a.s_mean[px * a.c_img + ch] = mean; a.s_std[px * a.c_img + ch] = sd; }
    else if (ch == a.c_obs - 1 && px < a.n_direct) { a.s_dmean[px] = mean; a.s_dstd[px] = sd; }
  }
}
#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void norm_update_kernel(NormUpdateArgs a) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  double cnt = a.count[a.parity];
  if (e < a.elems) {
    float bm, bv;
    norm_batch_moments(a, e, bm, bv);
    double mean = a.mean[e], var = a.var[e];
    norm_chan_merge(mean, var, cnt, bm, bv, a.n, true);
    norm_store(a, e, mean, var);
  } else {
    cnt += (double)a.n;
  }
  if (e == 0) a.count[a.parity ^ 1] = cnt;
}
#endif

// ------------------------------------------------------------------------------------------------
// squashed Gaussian policy head (A.3).  mu/ls_raw are the `dense` / `dense_1` outputs.
#define GRL_EPS 1e-6f
#define GRL_LOG_STD_MAX 2.0f
#define GRL_LOG_STD_MIN (-20.0f)

struct SampleArgs {
  const float* mu; const float* ls_raw; const float* eps; int B, A;
  float* pi;       // tanh(mu + std*eps)        [B,A]
  float* det;      // tanh(mu)                  [B,A]
  float* logp;     // [B]
  float* entropy;  // [B]
};

#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void sample_kernel(SampleArgs a) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  float logp = 0.f, ent = 0.f;
  for (int j = 0; j < a.A; ++j) {
    const float mu = a.mu[b * a.A + j];
    const float ls = fminf(fmaxf(a.ls_raw[b * a.A + j], GRL_LOG_STD_MIN), GRL_LOG_STD_MAX);
    const float sd = expf(ls);
    const float u = mu + a.eps[b * a.A + j] * sd;
    const float z = (u - mu) / (sd + GRL_EPS);
    logp += -0.5f * (z * z + 2.f * ls + 1.8378770664093453f);        // log(2*pi)
    ent += ls + 1.4189385332046727f;                                   // 0.5*log(2*pi*e)
    const float t = tanhf(u);
    logp -= logf(1.f - t * t + GRL_EPS);
    a.pi[b * a.A + j] = t;
    if (a.det) a.det[b * a.A + j] = tanhf(mu);
  }
  a.logp[b] = logp;
  a.entropy[b] = ent;
}
#endif

// backward of policy_loss = mean(ent_coef*logp - qf1_pi) through the squashing / sampling.
struct SampleBwdArgs {
  const float* mu; const float* ls_raw; const float* eps; const float* pi;
  const float* da;             // dL/d a_pi from the qf1 head backward   [B,A] (row stride ld_da)
  int ld_da;
  const float* log_ent_coef;   // parameter scalar
  int B, A;
  float* dmu; float* dls;      // [B,A]
};

#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void sample_bwd_kernel(SampleBwdArgs a) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  const float alpha_over_b = expf(a.log_ent_coef[0]) / (float)a.B;
  for (int j = 0; j < a.A; ++j) {
    const float lr = a.ls_raw[b * a.A + j];
    const float ls = fminf(fmaxf(lr, GRL_LOG_STD_MIN), GRL_LOG_STD_MAX);
    const float sd = expf(ls);
    const float ep = a.eps[b * a.A + j];
    const float t = a.pi[b * a.A + j];
    const float omt = 1.f - t * t;
    // d/du [ alpha/B * (-log(1 - t^2 + EPS)) + (dL/da) * t ]
    const float du = a.da[(long)b * a.ld_da + j] * omt + alpha_over_b * (2.f * t * omt / (omt + GRL_EPS));
    a.dmu[b * a.A + j] = du;                       // Gaussian term has no mu-gradient: u - mu = std*eps
    // Gaussian log-likelihood term: -0.5*(z^2 + 2 ls + c), z = sd*eps/(sd+EPS)
    const float den = sd + GRL_EPS;
    const float z = sd * ep / den;
    const float dz_dls = ep * sd * GRL_EPS / (den * den);
    float dls = du * sd * ep + alpha_over_b * (-(z * dz_dls + 1.f));
    if (lr < GRL_LOG_STD_MIN || lr > GRL_LOG_STD_MAX) dls = 0.f;     // clip_by_value gradient
    a.dls[b * a.A + j] = dls;
  }
}
#endif

// one element of TF-1.x Adam (A.5; epsilon outside the bias correction, which lives in alpha)
// The fused multiply-adds are spelled out and contraction is off inside: the scalar, the 4-wide and the stand-alone
// forms of the optimiser must round identically (test_launch_plan_switches_do_not_touch_arithmetic), which is not
// something to leave to where the compiler happens to contract.
__device__ __forceinline__ void adam_elem(float g, float& p, float& m, float& v, float alpha, float eps) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  const float omb1 = 1.f - 0.9f, omb2 = 1.f - 0.999f;
  m = fmaf(g - m, omb1, m);
  v = fmaf(fmaf(g, g, -v), omb2, v);
  p = p - (m * alpha) / (sqrtf(v) + eps);
}
__device__ __forceinline__ float grad_scaled(float g, float scale) {   // data-parallel mean: never folded into an fma
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  return g * scale;
}
// Polyak average of one element (A.4): (1-tau)*target + tau*source
__device__ __forceinline__ float polyak_elem(float tg, float p, float tau) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  return fmaf(tau, p, (1.f - tau) * tg);
}

// ------------------------------------------------------------------------------------------------
// losses (A.4) and their gradients w.r.t. the critic / value outputs; Adam step size; metrics.
struct LossArgs {
  int B; float gamma, target_entropy, lr;
  const float* rew; const float* done;
  const float* v_tgt; const float* qf1; const float* qf2; const float* v;
  const float* qf1_pi; const float* qf2_pi; const float* logp; const float* entropy;
  const float* log_ent_coef;
  float* d_qf1; float* d_qf2; float* d_v; float* d_qf1_pi;   // [B] each, element stride ld_d
  int ld_d;
  float* g_log_ent_coef;                                      // gradient slot in the flat bucket
  DevScalars* sc;
  int write_d;   // 0: the output gradients are produced row-locally by heads_bwd_kernel; only the reductions run here
  int adam_ticked;   // the gather launch of this update already fixed adam_alpha / advanced the beta powers
  // fused apply (reduce_slabs_kernel with fuse_adam): this workgroup also applies Adam to log_ent_coef, the
  // one trainable scalar whose gradient is produced here
  float* ent_param; float* ent_m; float* ent_v;
  int keep_rng;   // 1: leave rng_step alone (multi-update calls whose head launch advances the counter, engine.hip "prefetch")
  int bump_img;   // 1: advance DevScalars.rng_img (the riders of the NEXT head launch draw the minibatch after the next)
};

#ifdef GRL_HOSTEMU
#include "elem_kernels_ref1.h"   // tests/hostemu: the emulation build only
#else
__device__ __forceinline__ void sac_loss_body(const LossArgs& a, int fuse_adam = 0) {
  __shared__ float red[8][256];
  const int t = threadIdx.x;
  const float log_alpha = a.log_ent_coef[0];
  const float alpha = expf(log_alpha);
  const float invB = 1.f / (float)a.B;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = t; b < a.B; b += 256) {
    const float qb = a.rew[b] + (1.f - a.done[b]) * a.gamma * a.v_tgt[b];
    const float e1 = a.qf1[b] - qb, e2 = a.qf2[b] - qb;
    const float lp = a.logp[b];
    const float vb = fminf(a.qf1_pi[b], a.qf2_pi[b]) - alpha * lp;
    const float ev = a.v[b] - vb;
    if (a.write_d) {
      a.d_qf1[b * a.ld_d] = e1 * invB;
      a.d_qf2[b * a.ld_d] = e2 * invB;
      a.d_v[b * a.ld_d] = ev * invB;
      a.d_qf1_pi[b * a.ld_d] = -invB;
    }
    s[0] += 0.5f * e1 * e1;
    s[1] += 0.5f * e2 * e2;
    s[2] += 0.5f * ev * ev;
    s[3] += alpha * lp - a.qf1_pi[b];
    s[4] += lp + a.target_entropy;
    s[5] += a.entropy[b];
    s[6] += a.qf1[b];
    s[7] += a.v[b];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[k][t] = s[k];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off)
      for (int k = 0; k < 8; ++k) red[k][t] += red[k][t + off];
    __syncthreads();
  }
  if (t == 0) {
    DevScalars* sc = a.sc;
    sc->qf1_loss = red[0][0] * invB;
    sc->qf2_loss = red[1][0] * invB;
    sc->value_loss = red[2][0] * invB;
    sc->policy_loss = red[3][0] * invB;
    const float mean_lp_h = red[4][0] * invB;
    sc->ent_loss = -log_alpha * mean_lp_h;
    a.g_log_ent_coef[0] = -mean_lp_h;
    sc->ent_coef = alpha;
    sc->entropy = red[5][0] * invB;
    sc->mean_qf1 = red[6][0] * invB;
    sc->mean_v = red[7][0] * invB;
    // TF ApplyAdam: alpha_t = lr * sqrt(1 - beta2_power) / (1 - beta1_power); powers advance after
    if (!a.adam_ticked) {
      sc->adam_alpha = sc->lr * sqrtf(1.f - sc->beta2_power) / (1.f - sc->beta1_power);
      sc->beta1_power *= 0.9f;
      sc->beta2_power *= 0.999f;
    }
    if (fuse_adam) adam_elem(a.g_log_ent_coef[0], a.ent_param[0], a.ent_m[0], a.ent_v[0], sc->adam_alpha, 1e-8f);
    if (sc->rng_used && !a.keep_rng) { sc->rng_step += 1; sc->rng_used = 0u; }   // this minibatch came from the device RNG
    if (a.bump_img) sc->rng_img += 1;
  }
}
#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void sac_loss_kernel(LossArgs a) { sac_loss_body(a); }
#endif

#endif  // GRL_HOSTEMU

// ------------------------------------------------------------------------------------------------
// DQN / BDQ: dueling aggregation per branch, double-Q target averaged over branches,
// Huber (DQN) or squared (BDQ) TD loss with importance weights, gradients w.r.t. the advantage /
// value outputs of the online net, TD errors and priorities for prioritised replay.
struct QLossArgs {
  int B, D, n;               // rows, branches, bins per branch
  float gamma, lr;
  int huber, double_q;
  const float* adv0; const float* v0;     // online(s)      [B, D*n], [B]
  const float* adv1;                      // online(s')     (argmax; v and mean do not change it)
  const float* adv2; const float* v2;     // target(s')
  const float* act;                       // [B, D] bin indices stored as floats
  const float* rew; const float* done; const float* weights;   // [B]
  float* d_adv0; float* d_v0;             // gradients [B, D*nbp] (nbp >= n: a branch's bins start 16-byte aligned), [B] with stride ld_dv
  int nbp, ld_dv;                         //   -- the weight-gradient GEMM fetches them 16 bytes at a time (padding stays zero)
  float* td; float* priority;             // [B, D], [B]
  DevScalars* sc;
  float* row_part; unsigned* counter;     // [3 B] per-row partial sums, completion counter (zero between launches)
  int defer_finish;                       // 1: the batch sums / metrics are formed by a later launch from row_part (q_finish_sums):
                                          //    this kernel then needs no device-scope fence, counter or last-workgroup pass
  int loss_sum;                           // BDQ reading switch (grl_config.q_loss_sum_branches): TD loss SUMMED over the branches
                                          //    instead of averaged (gradients D times larger; target and priorities unchanged)
};

__device__ __forceinline__ void q_loss_row(const QLossArgs& a, int b, float* s3) {
  const int D = a.D, n = a.n;
  const float invB = 1.f / (float)a.B, invD = 1.f / (float)D, invn = 1.f / (float)n;
  float qbest = 0.f;
  for (int d = 0; d < D; ++d) {
    const float* sel = (a.double_q ? a.adv1 : a.adv2) + ((long)b * D + d) * n;
    const float* tg = a.adv2 + ((long)b * D + d) * n;
    int best = 0;
    float bv = sel[0], mean2 = 0.f;
    for (int k = 0; k < n; ++k) {
      if (sel[k] > bv) { bv = sel[k]; best = k; }
      mean2 += tg[k];
    }
    qbest += a.v2[b] + tg[best] - mean2 * invn;
  }
  qbest *= invD;
  const float y = a.rew[b] + a.gamma * (1.f - a.done[b]) * qbest;
  const float w = a.weights[b];
  float dv = 0.f, prio = 0.f, lossb = 0.f, qs = 0.f;
  for (int d = 0; d < D; ++d) {
    const float* ad = a.adv0 + ((long)b * D + d) * n;
    float mean0 = 0.f;
    for (int k = 0; k < n; ++k) mean0 += ad[k];
    mean0 *= invn;
    const int ai = (int)a.act[b * D + d];
    const float q_sel = a.v0[b] + ad[ai] - mean0;
    const float tdv = q_sel - y;
    a.td[b * D + d] = tdv;
    prio += fabsf(tdv);
    qs += q_sel;
    float err, dfd;
    if (a.huber) {
      const float at = fabsf(tdv);
      err = at < 1.f ? 0.5f * tdv * tdv : at - 0.5f;
      dfd = fminf(fmaxf(tdv, -1.f), 1.f);
    } else {
      err = tdv * tdv;
      dfd = 2.f * tdv;
    }
    lossb += err;
    const float g = w * invB * (a.loss_sum ? 1.f : invD) * dfd;
    dv += g;
    float* ga = a.d_adv0 + ((long)b * D + d) * a.nbp;
    for (int k = 0; k < n; ++k) ga[k] = g * ((k == ai ? 1.f : 0.f) - invn);
  }
  a.d_v0[(long)b * a.ld_dv] = dv;
  a.priority[b] = prio;
  s3[0] += w * lossb * (a.loss_sum ? 1.f : invD);
  s3[1] += qs * invD;
  s3[2] += prio * invD;
}

// batch means of the row partial sums + RNG tick (the part of q_loss_finish that does not feed the optimiser)
// the Philox counter moves on once the minibatch it drew has been consumed
__device__ __forceinline__ void q_rng_tick(DevScalars* sc) {
  if (sc->rng_used) { sc->rng_step += 1; sc->rng_used = 0u; }   // this minibatch was drawn by the device RNG
}
// tick_rng = false: the counter was advanced earlier in the update (prioritised multi-update calls, "per_pf": the launch these
// sums are formed in also carries the NEXT update's sampler, which reads the counter and marks rng_used again)
__device__ __forceinline__ void q_metrics(DevScalars* sc, int B, float loss, float qm, float tdm, bool tick_rng = true) {
  const float invB = 1.f / (float)B;
  sc->policy_loss = loss * invB;     // reported as the TD loss
  sc->mean_qf1 = qm * invB;
  sc->value_loss = tdm * invB;       // mean |td|
  if (tick_rng) q_rng_tick(sc);
}
// deferred form: one workgroup (256 threads) sums row_part in the fixed tree order of q_loss_kernel's last workgroup
__device__ __forceinline__ void q_finish_sums(DevScalars* sc, const float* row_part, int B, float (*red)[256], bool tick_rng = true) {
  const int t = threadIdx.x;           // workgroups of 256 threads or more: the first 256 carry the sums, all reach the barriers
  if (t < 256) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int r = t; r < B; r += 256) { s0 += row_part[3 * r]; s1 += row_part[3 * r + 1]; s2 += row_part[3 * r + 2]; }
    red[0][t] = s0; red[1][t] = s1; red[2][t] = s2;
  }
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off)
      for (int k = 0; k < 3; ++k) red[k][t] += red[k][t + off];
    __syncthreads();
  }
  if (t == 0) q_metrics(sc, B, red[0][0], red[1][0], red[2][0], tick_rng);
}

__device__ __forceinline__ void q_loss_finish(const QLossArgs& a, float loss, float qm, float tdm) {
  DevScalars* sc = a.sc;
  const float invB = 1.f / (float)a.B;
  sc->policy_loss = loss * invB;     // reported as the TD loss
  sc->mean_qf1 = qm * invB;
  sc->value_loss = tdm * invB;       // mean |td|
  sc->adam_alpha = sc->lr * sqrtf(1.f - sc->beta2_power) / (1.f - sc->beta1_power);
  sc->beta1_power *= 0.9f;
  sc->beta2_power *= 0.999f;
  if (sc->rng_used) { sc->rng_step += 1; sc->rng_used = 0u; }   // this minibatch was drawn by the device RNG
}

#ifdef GRL_HOSTEMU
#include "elem_kernels_ref2.h"   // tests/hostemu: the emulation build only
#else
// fallback for more than 64 bins per branch: one thread per row
#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void q_loss_rows_kernel(QLossArgs a) {
  __shared__ float red[3][256];
  const int t = threadIdx.x;
  float s3[3] = {0, 0, 0};
  for (int b = t; b < a.B; b += 256) q_loss_row(a, b, s3);
  for (int k = 0; k < 3; ++k) red[k][t] = s3[k];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off)
      for (int k = 0; k < 3; ++k) red[k][t] += red[k][t + off];
    __syncthreads();
  }
  if (t == 0) {
    if (a.defer_finish) q_metrics(a.sc, a.B, red[0][0], red[1][0], red[2][0]);   // (single workgroup: nothing to defer but the optimiser tick)
    else q_loss_finish(a, red[0][0], red[1][0], red[2][0]);
  }
}
#endif

// One wavefront per minibatch row, lane = bin (n <= 64): the bin loops of q_loss_row become wave reductions
// (fixed butterfly order), the D branches stay a loop.  Row partial sums go to a.row_part; the last
// workgroup to finish (device-scope counter) adds them in row order and runs q_loss_finish.
__device__ __forceinline__ float q_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
// The same reductions without the LDS crossbar: __shfl_xor is a ds_bpermute, ~120 cycles of dependent latency each, and a
// row of the loss chains 18 of them per branch (the kernel was 10 us of which 7 were these).  DPP moves cost a few cycles:
// four steps (lane ^ 1, lane ^ 2, mirror inside 8, mirror inside 16) leave the sum of each 16-lane row in all of its lanes,
// the four row values are read back through SGPRs and added in row order.  Deterministic, every lane gets the same value.
template <int CTRL> __device__ __forceinline__ float q_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ int q_dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
enum { Q_DPP_XOR1 = 0xB1, Q_DPP_XOR2 = 0x4E, Q_DPP_HALF_MIRROR = 0x141, Q_DPP_MIRROR = 0x140 };
__device__ __forceinline__ float q_lane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float q_wave_sum_dpp(float v) {
  v += q_dpp<Q_DPP_XOR1>(v);
  v += q_dpp<Q_DPP_XOR2>(v);
  v += q_dpp<Q_DPP_HALF_MIRROR>(v);
  v += q_dpp<Q_DPP_MIRROR>(v);
  return (q_lane_f(v, 0) + q_lane_f(v, 16)) + (q_lane_f(v, 32) + q_lane_f(v, 48));
}
// arg max over the wave, ties -> lowest lane (the sequential scan keeps the first maximum); returns the winning lane
__device__ __forceinline__ int q_wave_argmax_dpp(float sv, int si) {
#define Q_AM_STEP(CTRL)                                                     \
  {                                                                         \
    const float ov = q_dpp<CTRL>(sv);                                       \
    const int oi = q_dpp_i<CTRL>(si);                                       \
    if (ov > sv || (ov == sv && oi < si)) { sv = ov; si = oi; }             \
  }
  Q_AM_STEP(Q_DPP_XOR1) Q_AM_STEP(Q_DPP_XOR2) Q_AM_STEP(Q_DPP_HALF_MIRROR) Q_AM_STEP(Q_DPP_MIRROR)
#undef Q_AM_STEP
  float bv = q_lane_f(sv, 0);
  int bi = __builtin_amdgcn_readlane(si, 0);
#pragma unroll
  for (int r = 1; r < 4; ++r) {
    const float ov = q_lane_f(sv, 16 * r);
    const int oi = __builtin_amdgcn_readlane(si, 16 * r);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  return __builtin_amdgcn_readfirstlane(bi);
}
// One minibatch row of the loss on one wavefront, lane = bin (n <= 64, D <= Q_DM), in two steps: the operands of the row --
// requested in one burst (one memory round trip instead of 2 D) -- and the arithmetic.  emit_g(d, g, ai, td) receives the
// loss gradient scale, the stored bin and the TD error of branch d (wave-uniform values, every lane calls it), emit_row(dv,
// priority, weighted loss, mean selected Q) the row's results.  (The backward chains of q_chain.h form the loss of their own
// rows with the same arithmetic on 16-lane groups: qc_rows_loss.)
enum { Q_DM = 8 };
struct QRowIn { float xs[Q_DM], xt[Q_DM], x0[Q_DM]; int ais[Q_DM]; float v2b, v0b, rewb, doneb, w; };
__device__ __forceinline__ void q_row_load(const QLossArgs& a, int b, int lane, QRowIn& in) {
  const int D = a.D, n = a.n;
  const bool on = lane < n;
  const float* sel = a.double_q ? a.adv1 : a.adv2;
#pragma unroll
  for (int d = 0; d < Q_DM; ++d) {
    in.xs[d] = -INFINITY; in.xt[d] = 0.f; in.x0[d] = 0.f; in.ais[d] = 0;
    if (d < D) {
      const long o = ((long)b * D + d) * n + lane;
      if (on) { in.xs[d] = sel[o]; in.xt[d] = a.adv2[o]; in.x0[d] = a.adv0[o]; }
      in.ais[d] = (int)a.act[b * D + d];
    }
  }
  in.v2b = a.v2[b]; in.v0b = a.v0[b]; in.rewb = a.rew[b]; in.doneb = a.done[b]; in.w = a.weights[b];
}
template <class EmitG, class EmitRow>
__device__ __forceinline__ void q_row_loss(const QLossArgs& a, const QRowIn& in, int lane, EmitG&& emit_g, EmitRow&& emit_row) {
  const int D = a.D, n = a.n;
  const float invB = 1.f / (float)a.B, invD = 1.f / (float)D, invn = 1.f / (float)n;
  float qbest = 0.f;
#pragma unroll
  for (int d = 0; d < Q_DM; ++d) {
    if (d < D) {
      const float tg = in.xt[d];
      const int si = q_wave_argmax_dpp(in.xs[d], lane);      // (wave-uniform)
      const float mean2 = q_wave_sum_dpp(tg);
      qbest += in.v2b + q_lane_f(tg, si) - mean2 * invn;
    }
  }
  qbest *= invD;
  const float y = in.rewb + a.gamma * (1.f - in.doneb) * qbest;
  float dv = 0.f, prio = 0.f, lossb = 0.f, qs = 0.f;
#pragma unroll
  for (int d = 0; d < Q_DM; ++d) {
    if (d < D) {
      const float ad = in.x0[d];
      const float mean0 = q_wave_sum_dpp(ad) * invn;
      const int ai = in.ais[d];
      const float q_sel = in.v0b + q_lane_f(ad, __builtin_amdgcn_readfirstlane(ai)) - mean0;
      const float tdv = q_sel - y;
      prio += fabsf(tdv);
      qs += q_sel;
      float err, dfd;
      if (a.huber) {
        const float at = fabsf(tdv);
        err = at < 1.f ? 0.5f * tdv * tdv : at - 0.5f;
        dfd = fminf(fmaxf(tdv, -1.f), 1.f);
      } else {
        err = tdv * tdv;
        dfd = 2.f * tdv;
      }
      lossb += err;
      const float g = in.w * invB * (a.loss_sum ? 1.f : invD) * dfd;
      dv += g;
      emit_g(d, g, ai, tdv);
    }
  }
  emit_row(dv, prio, in.w * lossb * (a.loss_sum ? 1.f : invD), qs * invD);
}
#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void q_loss_kernel(QLossArgs a) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int D = a.D, n = a.n;
  if (b < a.B && D <= Q_DM) {
    QRowIn in;
    q_row_load(a, b, lane, in);
    const float invD = 1.f / (float)D, invn = 1.f / (float)n;
    q_row_loss(a, in, lane,
               [&](int d, float g, int ai, float tdv) {
                 if (lane < n) a.d_adv0[((long)b * D + d) * a.nbp + lane] = g * ((lane == ai ? 1.f : 0.f) - invn);
                 if (lane == 0) a.td[b * D + d] = tdv;
               },
               [&](float dv, float prio, float wloss, float qsel) {
                 if (lane == 0) {
                   a.d_v0[(long)b * a.ld_dv] = dv;
                   a.priority[b] = prio;
                   a.row_part[3 * b] = wloss;
                   a.row_part[3 * b + 1] = qsel;
                   a.row_part[3 * b + 2] = prio * invD;
                 }
               });
  } else if (b < a.B) {
    const float invB = 1.f / (float)a.B, invD = 1.f / (float)D, invn = 1.f / (float)n;
    const bool on = lane < n;
    float qbest = 0.f;
    for (int d = 0; d < D; ++d) {
      const long o = ((long)b * D + d) * n + lane;
      float sv = on ? (a.double_q ? a.adv1 : a.adv2)[o] : -INFINITY;
      const float tg = on ? a.adv2[o] : 0.f;
      int si = lane;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {      // arg max, ties -> lowest bin (the sequential scan keeps the first maximum)
        const float ov = __shfl_xor(sv, m, 64);
        const int oi = __shfl_xor(si, m, 64);
        if (ov > sv || (ov == sv && oi < si)) { sv = ov; si = oi; }
      }
      const float mean2 = q_wave_sum(tg);
      qbest += a.v2[b] + __shfl(tg, si, 64) - mean2 * invn;
    }
    qbest *= invD;
    const float y = a.rew[b] + a.gamma * (1.f - a.done[b]) * qbest;
    const float w = a.weights[b];
    float dv = 0.f, prio = 0.f, lossb = 0.f, qs = 0.f;
    for (int d = 0; d < D; ++d) {
      const long o = ((long)b * D + d) * n + lane;
      const float ad = on ? a.adv0[o] : 0.f;
      const float mean0 = q_wave_sum(ad) * invn;
      const int ai = (int)a.act[b * D + d];
      const float q_sel = a.v0[b] + __shfl(ad, ai, 64) - mean0;
      const float tdv = q_sel - y;
      prio += fabsf(tdv);
      qs += q_sel;
      float err, dfd;
      if (a.huber) {
        const float at = fabsf(tdv);
        err = at < 1.f ? 0.5f * tdv * tdv : at - 0.5f;
        dfd = fminf(fmaxf(tdv, -1.f), 1.f);
      } else {
        err = tdv * tdv;
        dfd = 2.f * tdv;
      }
      lossb += err;
      const float g = w * invB * (a.loss_sum ? 1.f : invD) * dfd;
      dv += g;
      if (on) a.d_adv0[((long)b * D + d) * a.nbp + lane] = g * ((lane == ai ? 1.f : 0.f) - invn);
      if (lane == 0) a.td[b * D + d] = tdv;
    }
    if (lane == 0) {
      a.d_v0[(long)b * a.ld_dv] = dv;
      a.priority[b] = prio;
      a.row_part[3 * b] = w * lossb * (a.loss_sum ? 1.f : invD);
      a.row_part[3 * b + 1] = qs * invD;
      a.row_part[3 * b + 2] = prio * invD;
    }
  }
  if (a.defer_finish) return;   // uniform: the sums are formed from row_part by the launch that applies the update
  // ---- last workgroup: batch sums in row order
  __shared__ int last;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = atomicAdd(a.counter, 1u);
    last = done == gridDim.x - 1;
    if (last) *a.counter = 0u;                 // ready for the next launch (graph replay)
  }
  __syncthreads();
  if (last) {                                   // fixed tree order: deterministic
    __shared__ float red[3][256];
    __threadfence();
    const int t = threadIdx.x;
    const volatile float* rp = a.row_part;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int r = t; r < a.B; r += 256) { s0 += rp[3 * r]; s1 += rp[3 * r + 1]; s2 += rp[3 * r + 2]; }
    red[0][t] = s0; red[1][t] = s1; red[2][t] = s2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (t < off)
        for (int k = 0; k < 3; ++k) red[k][t] += red[k][t + off];
      __syncthreads();
    }
    if (t == 0) q_loss_finish(a, red[0][0], red[1][0], red[2][0]);
  }
}
#endif
#endif

// tf.clip_by_norm per variable: g <- g * clip / max(||g||, clip); one workgroup per variable
struct VarSeg { int64_t off; int64_t n; };

#ifdef GRL_HOSTEMU
#include "elem_kernels_ref3.h"   // tests/hostemu: the emulation build only
#else
#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void clip_by_norm_kernel(float* grads, const VarSeg* segs, float clip) {
  __shared__ float red[256];
  const VarSeg sg = segs[blockIdx.x];
  const int t = threadIdx.x;
  float ss = 0.f;
  for (int64_t i = t; i < sg.n; i += 256) { const float g = grads[sg.off + i]; ss += g * g; }
  red[t] = ss;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) red[t] += red[t + off];
    __syncthreads();
  }
  const float sc = clip / fmaxf(sqrtf(red[0]), clip);
  for (int64_t i = t; i < sg.n; i += 256) grads[sg.off + i] *= sc;
}
#endif
#endif

// Q-values of the act path: q[b, d, k] = v[b] + adv[b, d, k] - mean_k adv[b, d, :]
#ifndef GRL_ELEM_TYPES_ONLY
// q_host / done (optional): the values also go to page-locked coherent host memory and every workgroup counts itself into
// `done` once its stores are out -- grl_act polls that instead of a device-to-host copy + stream synchronisation (act_mfma.h)
__global__ __launch_bounds__(256) void dueling_kernel(const float* adv, const float* v, int rows, int D, int n,
                                                     float* q, float* q_host, unsigned* done) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < rows * D) {
    const int b = i / D;
    const float* a = adv + (long)i * n;
    float m = 0.f;
    for (int k = 0; k < n; ++k) m += a[k];
    m /= (float)n;
    for (int k = 0; k < n; ++k) {
      const float val = v[b] + a[k] - m;
      q[(long)i * n + k] = val;
      if (q_host) q_host[(long)i * n + k] = val;
    }
  }
#ifndef GRL_HOSTEMU
  if (done) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#else
  if (done && threadIdx.x == 0) *done += 1u;      // (emulation: launches run synchronously, the host's poll finds the count at once)
#endif
}
#endif

// ------------------------------------------------------------------------------------------------
// sum split slabs of weight gradients into the flat gradient bucket
struct AdamArgs {
  float* params; const float* grads; float* m; float* v;
  int64_t n_train;
  const DevScalars* sc;
  float grad_scale, tau;
  float eps;                   // 1e-8 (TF AdamOptimizer), 1e-7 (Keras Adam)
  int64_t src_ofs, n_polyak;   // source range [src_ofs, src_ofs+n_polyak) inside params
  float* target;               // target block
  // reduce_slabs with the fused apply: 1 = do not write the summed gradients to the bucket (5.4 MB of stores per update).  Only
  // for the first / middle updates of a multi-update call: no caller can look at their gradients (grl_get_gradients after the
  // call sees the LAST update's, whose launch keeps the store), and the data-parallel launches publish through `mirror`
  int skip_bucket = 0;
};

#ifndef GRL_RS_QUADS
#define GRL_RS_QUADS 1     // (measurement builds: scripts/ab_build.sh build "-DGRL_RS_QUADS=2" engine)
#endif
enum { RS_QUADS = GRL_RS_QUADS };     // quads (4 consecutive outputs) per thread of the 16-byte reduction path: a tile is 1024 * RS_QUADS outputs
struct ReduceDesc {
  float* dst; const float* src; int32_t n; int32_t splits; int64_t slab_stride;
  int32_t vec;   // 1: n, slab_stride multiples of 4 and dst / src 16-byte aligned -- a thread sums RS_QUADS x 4 consecutive outputs
                 //    (tile = 1024 * RS_QUADS outputs instead of 256); set by the host (reduce_tiles)
  int32_t row_len, src_ld;   // row_len > 0: output i comes from src[(i / row_len) * src_ld + i % row_len] -- a column
                             // block of a slab whose rows hold several variables side by side (conv1 of both nets)
};

// flat work list: block b sums 256 consecutive outputs of descriptor tiles[b].x starting at tiles[b].y
// (slab loads are independent: unrolled so several are in flight; the add order stays k = 0, 1, ...)
// ... and, as workgroup n_tiles when has_loss is set, the batch reductions of the SAC losses (metrics,
// entropy-coefficient gradient, Adam step size): they are needed by the apply kernel only.
// System-scope WRITE-THROUGH stores (buffer store with sc0 | sc1 / a system-scope atomic store): what a kernel uses for data
// that another GPU reads while kernels are running -- nothing stays dirty in an L2, a drained store has reached memory
// (csrc/dp_kernels.h: the exchange step of the data-parallel update).
#ifdef GRL_HOSTEMU
struct sys_f4 { float v[4]; };
static inline void st_sys_quad(float* base, int64_t quad, const float (&v)[4]) { for (int k = 0; k < 4; ++k) base[4 * quad + k] = v[k]; }
static inline void st_sys_f1(float* p, float v) { *p = v; }
static inline float ld_sys_f1(const float* p) { return *p; }
#else
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sys_rsrc(const float* p) {
  const uint64_t a = (uint64_t)p;     // (made provably wave-uniform: no waterfall loop around the buffer instructions)
  const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
enum { SYS_SCOPE = 1 | 16 };     // buffer-instruction cache policy: sc0 | sc1
__device__ __forceinline__ void st_sys_quad(float* base, int64_t quad, const float (&v)[4]) {
  typedef unsigned int sys_u4 __attribute__((ext_vector_type(4)));
  typedef float sys_f4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sys_u4, sys_f4{v[0], v[1], v[2], v[3]}), sys_rsrc(base), (int)(quad << 4), 0, SYS_SCOPE);
}
__device__ __forceinline__ void st_sys_f1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float ld_sys_f1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
#endif

// mirror (optional): a second, bucket-shaped array that receives every sum write-through (the exchange buffer of the
// data-parallel update: the reduction publishes the gradients itself, no copy kernel)
__device__ __forceinline__ void reduce_slabs_body(const ReduceDesc* __restrict__ descs, const int2* __restrict__ tiles, int n_tiles,
                                                  const LossArgs& la, int has_loss, const AdamArgs& aa, int fuse_adam, const int blk,
                                                  float* mirror = nullptr) {
  if (blk >= n_tiles) {
    if (has_loss) sac_loss_body(la, fuse_adam);
    return;
  }
  const int2 tl = tiles[blk];
  const ReduceDesc d = descs[tl.x];
#ifndef GRL_HOSTEMU
  if (d.vec) {
    // four consecutive outputs per thread and quad, RS_QUADS quads per thread (1024 outputs apart), 16-byte accesses
    // throughout.  The pass is latency-bound: a workgroup lives for three dependent memory round trips (descriptor, loads,
    // stores).  RS_QUADS = 2 (half the workgroups, twice the bytes in flight per thread) measured SLOWER on the MI355X:
    // 17.2 against 14.2 us for the reduction + Adam launch, 5 628 against 5 732 updates/s (round 5, scripts/ab_build.sh) --
    // many thin workgroups beat few fat ones here; the default stays 1.  Per element the arithmetic is the scalar path's, in
    // the same order.
    typedef float rs_f4 __attribute__((ext_vector_type(4)));
    // (descriptor pointers are generic to the compiler -> FLAT loads; typed as global they are global_load_dwordx4)
    typedef const __attribute__((address_space(1))) rs_f4* rs_gq;
    typedef __attribute__((address_space(1))) rs_f4* rs_gqw;
    constexpr int NQ = RS_QUADS;
    const rs_f4 z4 = {0.f, 0.f, 0.f, 0.f};
    int ii[NQ];
    bool on[NQ], pol[NQ];
    int64_t e[NQ], kp[NQ];
    rs_f4 p[NQ], m[NQ], v[NQ], tg[NQ], s[NQ];
    const float* __restrict__ src[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      ii[u] = tl.y + 4 * threadIdx.x + 1024 * u;
      on[u] = ii[u] < d.n;
      e[u] = fuse_adam ? (d.dst + ii[u]) - aa.grads : 0;
      kp[u] = e[u] - aa.src_ofs;
      pol[u] = on[u] && fuse_adam && kp[u] >= 0 && kp[u] < aa.n_polyak;     // variables are padded to quads: no straddling
      p[u] = m[u] = v[u] = tg[u] = s[u] = z4;
      if (on[u] && fuse_adam) {
        p[u] = *(const rs_f4*)(aa.params + e[u]); m[u] = *(const rs_f4*)(aa.m + e[u]); v[u] = *(const rs_f4*)(aa.v + e[u]);
        if (pol[u]) tg[u] = *(const rs_f4*)(aa.target + kp[u]);
      }
      const int i = on[u] ? ii[u] : 0;
      src[u] = d.src + (d.row_len > 0 ? (long)(i / d.row_len) * d.src_ld + i % d.row_len : i);
    }
    int k = 0;
    // slab loads in batches of RS_BATCH per quad (2 x RS_BATCH in flight per thread), summed in slab order
#ifndef GRL_RS_BATCH
#define GRL_RS_BATCH 6
#endif
    constexpr int RS_BATCH = GRL_RS_BATCH;
    for (; k + RS_BATCH <= d.splits; k += RS_BATCH) {
      rs_f4 vv[NQ][RS_BATCH];
#pragma unroll
      for (int u = 0; u < NQ; ++u)
        if (on[u]) {
#pragma unroll
          for (int j = 0; j < RS_BATCH; ++j) vv[u][j] = *(rs_gq)(src[u] + (long)(k + j) * d.slab_stride);
        }
#pragma unroll
      for (int u = 0; u < NQ; ++u)
        if (on[u]) {
#pragma unroll
          for (int j = 0; j < RS_BATCH; ++j) s[u] += vv[u][j];
        }
    }
    for (; k < d.splits; ++k) {
      rs_f4 vv[NQ];
#pragma unroll
      for (int u = 0; u < NQ; ++u)
        if (on[u]) vv[u] = *(rs_gq)(src[u] + (long)k * d.slab_stride);
#pragma unroll
      for (int u = 0; u < NQ; ++u)
        if (on[u]) s[u] += vv[u];
    }
    const float alpha = fuse_adam ? aa.sc->adam_alpha : 0.f;
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      if (!on[u]) continue;
      // (data parallel: the sums go to the exchange buffer ONLY -- nothing reads this rank's own bucket before the exchange)
      if (mirror) { const float sv[4] = {s[u].x, s[u].y, s[u].z, s[u].w}; st_sys_quad(mirror, ((d.dst + ii[u]) - aa.grads) >> 2, sv); }
      else if (!(fuse_adam && aa.skip_bucket)) *(rs_gqw)(d.dst + ii[u]) = s[u];
      if (fuse_adam) {
        float pe[4] = {p[u].x, p[u].y, p[u].z, p[u].w}, me[4] = {m[u].x, m[u].y, m[u].z, m[u].w}, ve[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        const float ge[4] = {s[u].x, s[u].y, s[u].z, s[u].w};
#pragma unroll
        for (int w = 0; w < 4; ++w) adam_elem(grad_scaled(ge[w], aa.grad_scale), pe[w], me[w], ve[w], alpha, aa.eps);
        const rs_f4 p2 = {pe[0], pe[1], pe[2], pe[3]}, m2 = {me[0], me[1], me[2], me[3]}, v2 = {ve[0], ve[1], ve[2], ve[3]};
        *(rs_f4*)(aa.params + e[u]) = p2; *(rs_f4*)(aa.m + e[u]) = m2; *(rs_f4*)(aa.v + e[u]) = v2;
        if (pol[u]) {
          rs_f4 t2;
          t2.x = polyak_elem(tg[u].x, p2.x, aa.tau); t2.y = polyak_elem(tg[u].y, p2.y, aa.tau);
          t2.z = polyak_elem(tg[u].z, p2.z, aa.tau); t2.w = polyak_elem(tg[u].w, p2.w, aa.tau);
          *(rs_f4*)(aa.target + kp[u]) = t2;
        }
      }
    }
    return;
  }
#endif
  const int i = tl.y + threadIdx.x;
  if (i < d.n) {
    // fused apply: the element's Adam state is requested before the slab sums (independent loads)
    const int64_t e = fuse_adam ? (d.dst + i) - aa.grads : 0;
    float p = 0.f, m = 0.f, v = 0.f, tg = 0.f;
    const int64_t kp = e - aa.src_ofs;
    const bool pol = fuse_adam && kp >= 0 && kp < aa.n_polyak;
    if (fuse_adam) { p = aa.params[e]; m = aa.m[e]; v = aa.v[e]; if (pol) tg = aa.target[kp]; }
    const float* __restrict__ src = d.src + (d.row_len > 0 ? (long)(i / d.row_len) * d.src_ld + i % d.row_len : i);
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= d.splits; k += 8) {
      float vv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) vv[u] = src[(long)(k + u) * d.slab_stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += vv[u];
    }
    for (; k < d.splits; ++k) s += src[(long)k * d.slab_stride];
    if (mirror) st_sys_f1(mirror + ((d.dst + i) - aa.grads), s);
    else if (!(fuse_adam && aa.skip_bucket)) d.dst[i] = s;
    if (fuse_adam) {   // the update every trainable tensor gets from adam_polyak_kernel, element by element
      adam_elem(grad_scaled(s, aa.grad_scale), p, m, v, aa.sc->adam_alpha, aa.eps);
      aa.params[e] = p; aa.m[e] = m; aa.v[e] = v;
      if (pol) aa.target[kp] = polyak_elem(tg, p, aa.tau);
    }
  }
}
#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const ReduceDesc* __restrict__ descs,
                                                          const int2* __restrict__ tiles, int n_tiles,
                                                          LossArgs la, int has_loss, AdamArgs aa, int fuse_adam) {
  reduce_slabs_body(descs, tiles, n_tiles, la, has_loss, aa, fuse_adam, (int)blockIdx.x);
}
#endif
// The same launch carrying, behind its own workgroups, the replay gather of the NEXT update (grid gx x B x 2 of
// gather_norm_kernel, linearised): inside a multi-update call nothing is added to the replay between updates, every
// reader of the minibatch tensors of update t has finished when this last launch of update t starts, and both halves
// are memory / latency bound -- they overlap instead of paying two launches (engine.hip, "prefetch").
#ifndef GRL_ELEM_TYPES_ONLY
template <bool GROUPED>
__global__ __launch_bounds__(256, 8) void reduce_slabs_gather_kernel(const ReduceDesc* __restrict__ descs,
                                                                 const int2* __restrict__ tiles, int n_tiles,
                                                                 LossArgs la, int has_loss, AdamArgs aa, int fuse_adam,
                                                                 GatherArgs ga, int gx) {
  // the two kinds of workgroups are dealt out in proportion (block x is a reduction block iff the running count
  // floor(x * nb / total) steps there): dispatched one kind after the other the launch ran as two phases
  const long nb = n_tiles + has_loss, total = (long)gridDim.x, x = (long)blockIdx.x;
  const long before = x * nb / total, upto = (x + 1) * nb / total;
  if (upto > before) { reduce_slabs_body(descs, tiles, n_tiles, la, has_loss, aa, fuse_adam, (int)before); return; }
  gather_norm_dispatch<GROUPED>(ga, gx, (int)(x - before));
}
#endif

// ------------------------------------------------------------------------------------------------
// TF-1.x Adam (A.5) over the flat trainable block, fused with the Polyak target update (A.4):
// target <- (1-tau)*target + tau*source for the leading `n_polyak` floats of model/values_fn.
#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void adam_polyak_kernel(AdamArgs a) {
  const float alpha = a.sc->adam_alpha;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_train;
       i += (int64_t)gridDim.x * 256) {
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
#endif

struct RngArgs {
  DevScalars* sc; uint64_t seed; int B, A;
  int64_t* idx; float* eps;
  float* ones; int mark;     // optional: ones[b] = 1 (uniform importance weights); mark: set rng_used instead of a separate tick launch
};

#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void rng_kernel(RngArgs a) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  const uint64_t step = a.sc->rng_step;
  if (b < a.B) {
    const int64_t size = a.sc->replay_size;
    uint32_t c[4] = {(uint32_t)step, (uint32_t)(step >> 32), (uint32_t)b, 0u};
    philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    const uint64_t u = ((uint64_t)c[0] << 32) | c[1];
    a.idx[b] = size > 0 ? (int64_t)__umul64hi(u, (uint64_t)size) : 0;
    for (int j0 = 0; j0 < a.A; j0 += 2) {
      uint32_t d[4] = {(uint32_t)step, (uint32_t)(step >> 32), (uint32_t)b, (uint32_t)(1 + j0)};
      philox4x32_10(d, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
      const float u1 = ((float)(d[0] >> 8) + 0.5f) * (1.f / 16777216.f);
      const float u2 = ((float)(d[1] >> 8) + 0.5f) * (1.f / 16777216.f);
      const float rad = sqrtf(-2.f * logf(u1));
      float sn, cs;
      sincosf(6.283185307179586f * u2, &sn, &cs);
      a.eps[b * a.A + j0] = rad * cs;
      if (j0 + 1 < a.A) a.eps[b * a.A + j0 + 1] = rad * sn;
    }
    if (a.ones) a.ones[b] = 1.f;
  }
  if (a.mark && b == 0) a.sc->rng_used = 1u;
}
#endif

#ifndef GRL_ELEM_TYPES_ONLY
__global__ void rng_tick_kernel(DevScalars* sc) { sc->rng_step += 1; }
#endif

#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void fill_kernel(float* p, float v, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}
#endif

// final tanh of the act path: a = deterministic ? tanh(mu) : tanh(mu + exp(clip(ls)) * eps)
#ifndef GRL_ELEM_TYPES_ONLY
__global__ __launch_bounds__(256) void act_out_kernel(const float* mu, const float* ls_raw,
                                                     const float* eps, int n, int A,
                                                     int deterministic, float* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * A) return;
  float u = mu[i];
  if (!deterministic) {
    const float ls = fminf(fmaxf(ls_raw[i], GRL_LOG_STD_MIN), GRL_LOG_STD_MAX);
    u += expf(ls) * eps[i];
  }
  out[i] = tanhf(u);
}
#endif

// ------------------------------------------------------------------------------------------------
// The policy head of the ACT path as one launch: L hidden layers (ReLU), the mu / log_std output layers and the final
// tanh (+ sampling), one workgroup per observation row (16 environments: 16 workgroups; each layer is a few thousand
// multiply-adds per row -- four dense launches and an output launch cost four kernel boundaries instead).
// Weights [K, N] row-major as everywhere; thread t owns output t % Np over the input slice t / Np, partial sums meet in LDS.
struct ActHeadsArgs {
  const float* x; int ldx, K0;
  const float* w[GRL_MAX_LAYERS]; const float* b[GRL_MAX_LAYERS]; int hid[GRL_MAX_LAYERS]; int L;
  const float* ow[2]; const float* ob[2]; int A;
  const float* eps; float* mu; float* ls; float* out; int rows; int deterministic;
  // act_mfma.h only: columns [0, n_sum) of the input are ReLU(sum of n_parts partial sums [rows, ld_parts] + x_bias) -- the
  // extractor's dense layer run as a split-K GEMM; n_parts == 0: every column comes from x
  const float* x_parts; const float* x_bias; int n_parts, n_sum, ld_parts; long part_stride;
  // act_mfma.h only: completion counter in page-locked host memory (one increment per workgroup), or nullptr -- grl_act polls it
  // instead of synchronising the stream
  unsigned* done;
};
enum { ACT_HEADS_MAX_IN = 2048, ACT_HEADS_MAX_HID = 256 };

#ifndef GRL_ELEM_TYPES_ONLY
#ifdef GRL_HOSTEMU
#include "act_heads_ref1.h"   // tests/hostemu: the emulation build only
#else
__device__ inline void act_heads_layer(const float* in, int K, const float* w, const float* b, int N, int relu, float* part,
                                       float* outv) {
  int Np = 1;
  while (Np < N) Np <<= 1;
  const int S = 256 / Np, t = threadIdx.x, o = t % Np, q = t / Np;
  float acc = 0.f;
  if (o < N)
    for (int i = q; i < K; i += S) acc = fmaf(in[i], w[(long)i * N + o], acc);
  part[t] = acc;
  __syncthreads();
  if (t < N) {
    float v = b[t];
    for (int j = 0; j < S; ++j) v += part[j * Np + t];
    outv[t] = relu ? fmaxf(v, 0.f) : v;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void act_heads_kernel(ActHeadsArgs a) {
  __shared__ float xs[ACT_HEADS_MAX_IN];
  __shared__ float hs[2][ACT_HEADS_MAX_HID];
  __shared__ float part[256];
  __shared__ float o2[2][64];
  const int r = blockIdx.x;
  if (r >= a.rows) return;
  for (int i = threadIdx.x; i < a.K0; i += 256) xs[i] = a.x[(long)r * a.ldx + i];
  __syncthreads();
  const float* in = xs;
  int K = a.K0;
  for (int l = 0; l < a.L; ++l) {
    act_heads_layer(in, K, a.w[l], a.b[l], a.hid[l], 1, part, hs[l & 1]);
    in = hs[l & 1];
    K = a.hid[l];
  }
  for (int k = 0; k < 2; ++k) act_heads_layer(in, K, a.ow[k], a.ob[k], a.A, 0, part, o2[k]);
  if (threadIdx.x < a.A) {
    const int i = r * a.A + threadIdx.x;
    float u = o2[0][threadIdx.x];
    a.mu[i] = u;
    a.ls[i] = o2[1][threadIdx.x];
    if (!a.deterministic) {
      const float ls = fminf(fmaxf(o2[1][threadIdx.x], GRL_LOG_STD_MIN), GRL_LOG_STD_MAX);
      u += expf(ls) * a.eps[i];
    }
    a.out[i] = tanhf(u);
  }
}
#endif  // GRL_HOSTEMU
#endif

#ifndef GRL_ELEM_TYPES_ONLY
// ---- what THIS build offers, answered at the kernel boundary (launch.h: the plans carry no build switches of their own)
// The TD-loss launch: the emulation's reference form walks all rows in one call and always finishes the batch sums itself; the
// device uses four rows per workgroup up to 64 bins, one workgroup beyond
static inline bool q_loss_finishes_itself(int n_bins) {
#ifdef GRL_HOSTEMU
  (void)n_bins;
  return true;
#else
  return n_bins <= 64;
#endif
}
static inline void launch_q_loss(const QLossArgs& a, hipStream_t s) {
#ifdef GRL_HOSTEMU
  hipLaunchKernelGGL(q_loss_kernel, dim3(1), dim3(256), 0, s, a);
#else
  if (a.n <= 64) hipLaunchKernelGGL(q_loss_kernel, dim3((a.B + 3) / 4), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(q_loss_rows_kernel, dim3(1), dim3(256), 0, s, a);
#endif
}
// 16-byte forms of the element-wise kernels (gather_norm's vec4 rows, the reduction's quads): device build only
static inline bool elem_vec4_built() {
#ifdef GRL_HOSTEMU
  return false;
#else
  return true;
#endif
}
#endif

}  // namespace grl
