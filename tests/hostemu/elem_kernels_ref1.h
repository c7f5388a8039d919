// elem_kernels_ref1.h -- TEST-ONLY reference form of the kernels of csrc/elem_kernels.h (sequential loops over the same descriptors),
// included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).
// Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.
// TEST-ONLY sequential form (see hostemu.h)
inline void sac_loss_body(const LossArgs& a, int fuse_adam = 0) {
  if (threadIdx.x != 0) return;
  const float log_alpha = a.log_ent_coef[0];
  const float alpha = expf(log_alpha);
  const float invB = 1.f / (float)a.B;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = 0; b < a.B; ++b) {
    const float qb = a.rew[b] + (1.f - a.done[b]) * a.gamma * a.v_tgt[b];
    const float e1 = a.qf1[b] - qb, e2 = a.qf2[b] - qb;
    const float lp = a.logp[b];
    const float vb = fminf(a.qf1_pi[b], a.qf2_pi[b]) - alpha * lp;
    const float ev = a.v[b] - vb;
    if (a.write_d) { a.d_qf1[b * a.ld_d] = e1 * invB; a.d_qf2[b * a.ld_d] = e2 * invB; a.d_v[b * a.ld_d] = ev * invB; a.d_qf1_pi[b * a.ld_d] = -invB; }
    s[0] += 0.5f * e1 * e1; s[1] += 0.5f * e2 * e2; s[2] += 0.5f * ev * ev;
    s[3] += alpha * lp - a.qf1_pi[b]; s[4] += lp + a.target_entropy; s[5] += a.entropy[b];
    s[6] += a.qf1[b]; s[7] += a.v[b];
  }
  DevScalars* sc = a.sc;
  sc->qf1_loss = s[0] * invB; sc->qf2_loss = s[1] * invB; sc->value_loss = s[2] * invB;
  sc->policy_loss = s[3] * invB;
  const float mean_lp_h = s[4] * invB;
  sc->ent_loss = -log_alpha * mean_lp_h;
  a.g_log_ent_coef[0] = -mean_lp_h;
  sc->ent_coef = alpha; sc->entropy = s[5] * invB; sc->mean_qf1 = s[6] * invB; sc->mean_v = s[7] * invB;
  if (!a.adam_ticked) {
    sc->adam_alpha = sc->lr * sqrtf(1.f - sc->beta2_power) / (1.f - sc->beta1_power);
    sc->beta1_power *= 0.9f;
    sc->beta2_power *= 0.999f;
  }
  if (fuse_adam) adam_elem(-mean_lp_h, a.ent_param[0], a.ent_m[0], a.ent_v[0], sc->adam_alpha, 1e-8f);
  if (sc->rng_used && !a.keep_rng) { sc->rng_step += 1; sc->rng_used = 0u; }
  if (a.bump_img) sc->rng_img += 1;
}
inline void sac_loss_kernel(LossArgs a) { sac_loss_body(a); }
