// q_kernels_ref1.h -- TEST-ONLY reference form of the kernels of csrc/q_kernels.h (sequential loops over the same descriptors),
// included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).
// Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.
inline void q_fwd_fused_kernel(QFusedArgs a) {
  if (threadIdx.x != 0) return;
  if (a.tick_sc && (blockIdx.x | blockIdx.y | blockIdx.z) == 0) { if (a.tick_rng) a.tick_sc->rng_step += 1; adam_tick_device(a.tick_sc); }
  const HtHead& h = a.fwd[blockIdx.y * (a.D + 1) + blockIdx.z];
  for (int row = blockIdx.x * HT_RB; row < std::min(a.B, (int)blockIdx.x * HT_RB + HT_RB); ++row)
    ht_ref_fwd_head(h, row, nullptr);
}
inline void q_bwd_towers_kernel(QFusedArgs a) {
  if (threadIdx.x != 0) return;
  const int tw = blockIdx.y;
  const HtHead& h = a.bwd_tw[tw];
  for (int row = blockIdx.x * HT_RB; row < std::min(a.B, (int)blockIdx.x * HT_RB + HT_RB); ++row) {
    float dv[2 * HT_MAXA];
    if (tw < a.D)
      for (int o = 0; o < a.nb; ++o) dv[o] = a.d_adv[((long)row * a.D + tw) * a.nbp + o];
    else dv[0] = a.d_v[(long)row * a.ld_dv];
    ht_ref_bwd_head(h, row, dv, h.n_xa ? a.dh_part + ((long)tw * a.B + row) * a.Ht : nullptr);
  }
}
inline void q_bwd_trunk_kernel(QFusedArgs a) {
  if (threadIdx.x != 0) return;
  for (int row = blockIdx.x * HT_RB; row < std::min(a.B, (int)blockIdx.x * HT_RB + HT_RB); ++row) {
    float dz[HT_MAXW];
    for (int n = 0; n < a.Ht; ++n) {
      float s = 0.f;
      for (int p = 0; p <= a.D; ++p) s += a.dh_part[((long)p * a.B + row) * a.Ht + n];
      dz[n] = s * a.trunk_scale;
    }
    ht_ref_bwd_head(*a.bwd_tr, row, nullptr, nullptr, dz);
  }
}
