// heads_mfma_ref1.h -- TEST-ONLY reference form of the kernels of csrc/heads_mfma.h (sequential loops over the same descriptors),
// included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).
// Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.
// ------------------------------------------------------------------------------------------------
// TEST-ONLY sequential form (see hostemu.h): one "thread" walks the rows of a block for one type.
// zs: [L][HT_MAXW] activations of this row, outs: [n_out * out_dim]
inline void hm_ref_fwd(const HtHead& h, int row, const float* xa_row, bool store, float* zs, float* outs) {
  for (int n = 0; n < h.H0; ++n) {
    float acc = h.u[(long)row * h.ldu + n];
    for (int sp = 1; sp < h.u_split; ++sp) acc += h.u[sp * h.u_stride + (long)row * h.ldu + n];
    for (int a = 0; a < h.n_xa; ++a) acc = fmaf(xa_row[a], h.w0a[a * h.H0 + n], acc);
    zs[n] = fmaxf(acc + h.b0[n], 0.f);
    if (store && h.z0) h.z0[(long)row * h.H0 + n] = zs[n];
  }
  for (int l = 1; l < h.L; ++l) {
    float* zi = zs + (l - 1) * HT_MAXW;
    float* zo = zs + l * HT_MAXW;
    for (int n = 0; n < h.hid[l]; ++n) {
      float acc = 0.f;
      for (int k = 0; k < h.hid[l - 1]; ++k) acc = fmaf(zi[k], h.w[l][k * h.hid[l] + n], acc);
      zo[n] = fmaxf(acc + h.b[l][n], 0.f);
      if (store && h.z[l]) h.z[l][(long)row * h.hid[l] + n] = zo[n];
    }
  }
  const int HL = h.hid[h.L - 1];
  const float* zl = zs + (h.L - 1) * HT_MAXW;
  for (int k = 0; k < h.n_out; ++k)
    for (int o = 0; o < h.out_dim; ++o) {
      float acc = 0.f;
      for (int n = 0; n < HL; ++n) acc = fmaf(zl[n], h.ow[k][n * h.out_dim + o], acc);
      outs[k * h.out_dim + o] = acc + h.ob[k][o];
      if (store) h.out[k][(long)row * (h.ld_out ? h.ld_out : h.out_dim) + o] = outs[k * h.out_dim + o];
    }
}
// dvals: n_out * out_dim output gradients; zs as left by hm_ref_fwd; writes g (when the head has g pointers), da_row
inline void hm_ref_bwd(const HtHead& h, int row, const float* zs, const float* dvals, float* da_row) {
  float gin[HT_MAXW], gout[HT_MAXW];
  const int HL = h.hid[h.L - 1];
  const float* zl = zs + (h.L - 1) * HT_MAXW;
  for (int n = 0; n < HL; ++n) {
    float acc = 0.f;
    for (int k = 0; k < h.n_out; ++k)
      for (int o = 0; o < h.out_dim; ++o) acc = fmaf(dvals[k * h.out_dim + o], h.ow[k][n * h.out_dim + o], acc);
    gin[n] = zl[n] > 0.f ? acc : 0.f;
  }
  for (int l = h.L - 1; l >= 1; --l) {
    if (h.g[l]) for (int n = 0; n < h.hid[l]; ++n) h.g[l][(long)row * h.hid[l] + n] = gin[n];
    const float* zp = zs + (l - 1) * HT_MAXW;
    for (int m = 0; m < h.hid[l - 1]; ++m) {
      float acc = 0.f;
      for (int n = 0; n < h.hid[l]; ++n) acc = fmaf(gin[n], h.w[l][m * h.hid[l] + n], acc);
      gout[m] = zp[m] > 0.f ? acc : 0.f;
    }
    for (int m = 0; m < h.hid[l - 1]; ++m) gin[m] = gout[m];
  }
  if (h.g0) for (int n = 0; n < h.H0; ++n) h.g0[(long)row * h.ldg0 + n] = gin[n];
  if (da_row)
    for (int a = 0; a < h.n_xa; ++a) {
      float acc = 0.f;
      for (int n = 0; n < h.H0; ++n) acc = fmaf(gin[n], h.w0a[a * h.H0 + n], acc);
      da_row[a] = acc;
    }
}

template <int W, bool FAST>
void heads_fused_kernel(const HeadsFusedArgs* ap, const GatherArgs g, const int gx, const int n_ride) {
  if (blockIdx.y >= 4) {      // the next update's image gather riding on the launch (every thread takes part)
    const int r = ((int)blockIdx.y - 4) * (int)gridDim.x + (int)blockIdx.x;
    if (r < n_ride) gather_images_rider(g, gx, r);
    return;
  }
  if (threadIdx.x != 0) return;
  const HeadsFusedArgs& a = *ap;
  if (blockIdx.x == 0 && blockIdx.y == 0 && a.sc) {
    if (a.rng_advance) a.sc->rng_step += 1;
    if (a.tick) adam_tick_device(a.sc);
  }
  const int type = blockIdx.y;
  const float alpha = expf(a.log_ent_coef[0]);
  const float invB = 1.f / (float)a.B;
  static thread_local float zs_a[GRL_MAX_LAYERS * HT_MAXW], zs_b[GRL_MAX_LAYERS * HT_MAXW];
  for (int row = blockIdx.x * HT_RB; row < std::min(a.B, (int)blockIdx.x * HT_RB + HT_RB); ++row) {
    float outs[2 * HT_MAXA], o1[2], o2[2], dv[2 * HT_MAXA];
    if (type <= 1) {
      const bool own = type == 0;
      hm_ref_fwd(a.h[0], row, nullptr, own, zs_a, outs);
      float pi[HT_MAXA], lp, en;
      ht_sample_row(outs, outs + a.A, a.eps + (long)row * a.A, a.A, pi, &lp, &en);
      if (own) {
        for (int j = 0; j < a.A; ++j) a.pi_a[(long)row * a.A + j] = pi[j];
        a.logp[row] = lp; a.ent[row] = en;
      }
      hm_ref_fwd(a.h[5], row, pi, false, zs_b, o1);
      if (own) {
        a.h[5].out[0][row] = o1[0];
        dv[0] = -invB;
        a.d_out[4][(long)row * a.ld_d] = dv[0];
        float da[HT_MAXA];
        HtHead q = a.h[5];
        q.g0 = nullptr;
        for (int l = 0; l < GRL_MAX_LAYERS; ++l) q.g[l] = nullptr;
        hm_ref_bwd(q, row, zs_b, dv, da);
        for (int j = 0; j < a.A; ++j) a.da_pi[(long)row * a.A + j] = da[j];
        ht_sample_bwd_row(outs + a.A, a.eps + (long)row * a.A, pi, da, a.A, alpha * invB, a.dmu + (long)row * a.ld_dm,
                          a.dls + (long)row * a.ld_dm);
        for (int j = 0; j < a.A; ++j) { dv[j] = a.dmu[(long)row * a.ld_dm + j]; dv[a.A + j] = a.dls[(long)row * a.ld_dm + j]; }
        hm_ref_bwd(a.h[0], row, zs_a, dv, nullptr);
      } else {
        hm_ref_fwd(a.h[6], row, pi, false, zs_b, o2);
        a.h[6].out[0][row] = o2[0];
        float v[2];
        hm_ref_fwd(a.h[1], row, nullptr, true, zs_a, v);
        dv[0] = (v[0] - (fminf(o1[0], o2[0]) - alpha * lp)) * invB;
        a.d_out[1][(long)row * a.ld_d] = dv[0];
        hm_ref_bwd(a.h[1], row, zs_a, dv, nullptr);
      }
    } else {
      float vt[2], q[2];
      hm_ref_fwd(a.h[4], row, nullptr, type == 2, zs_a, vt);
      const HtHead& h = a.h[type];
      hm_ref_fwd(h, row, h.xa + (long)row * h.ld_xa, true, zs_b, q);
      const float qb = a.rew[row] + (1.f - a.done[row]) * a.gamma * vt[0];
      dv[0] = (q[0] - qb) * invB;
      a.d_out[type][(long)row * a.ld_d] = dv[0];
      hm_ref_bwd(h, row, zs_b, dv, nullptr);
    }
  }
}

