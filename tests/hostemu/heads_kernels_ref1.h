// heads_kernels_ref1.h -- TEST-ONLY reference form of the kernels of csrc/heads_kernels.h (sequential loops over the same descriptors),
// included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).
// Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.
// ------------------------------------------------------------------------------------------------
// TEST-ONLY sequential forms (see hostemu.h): same arithmetic order, one "thread" does a block.
inline void ht_ref_fwd_head(const HtHead& h, int row, const float* xa_row) {
  float zin[HT_MAXW], zout[HT_MAXW];
  for (int n = 0; n < h.H0; ++n) {
    float acc = h.u[(long)row * h.ldu + n];
    for (int sp = 1; sp < h.u_split; ++sp) acc += h.u[sp * h.u_stride + (long)row * h.ldu + n];
    for (int a = 0; a < h.n_xa; ++a) acc = fmaf(xa_row[a], h.w0a[a * h.H0 + n], acc);
    acc += h.b0[n];
    zin[n] = fmaxf(acc, 0.f);
    if (h.z0) h.z0[(long)row * h.H0 + n] = zin[n];
  }
  for (int l = 1; l < h.L; ++l) {
    for (int n = 0; n < h.hid[l]; ++n) {
      float acc = 0.f;
      for (int k = 0; k < h.hid[l - 1]; ++k) acc = fmaf(zin[k], h.w[l][k * h.hid[l] + n], acc);
      zout[n] = fmaxf(acc + h.b[l][n], 0.f);
      if (h.z[l]) h.z[l][(long)row * h.hid[l] + n] = zout[n];
    }
    for (int n = 0; n < h.hid[l]; ++n) zin[n] = zout[n];
  }
  const int HL = h.hid[h.L - 1];
  for (int k = 0; k < h.n_out; ++k)
    for (int o = 0; o < h.out_dim; ++o) {
      float acc = 0.f;
      for (int n = 0; n < HL; ++n) acc = fmaf(zin[n], h.ow[k][n * h.out_dim + o], acc);
      h.out[k][(long)row * (h.ld_out ? h.ld_out : h.out_dim) + o] = acc + h.ob[k][o];
    }
}

inline void heads_fwd_kernel(HeadsFwdArgs a) {
  if (threadIdx.x != 0) return;
  const int chain = blockIdx.y;
  for (int row = blockIdx.x * HT_RB; row < std::min(a.B, (int)blockIdx.x * HT_RB + HT_RB); ++row) {
    if (chain == 0 || chain == 5) {   // both recompute the pi head; chain 0 owns its outputs
      ht_ref_fwd_head(a.h[0], row, nullptr);
      ht_sample_row(a.h[0].out[0] + (long)row * a.A, a.h[0].out[1] + (long)row * a.A, a.eps + (long)row * a.A, a.A,
                    a.pi_a + (long)row * a.A, a.logp + row, a.ent + row);
      ht_ref_fwd_head(a.h[chain == 0 ? 5 : 6], row, a.pi_a + (long)row * a.A);
    } else {
      const HtHead& h = a.h[chain];
      ht_ref_fwd_head(h, row, h.n_xa ? h.xa + (long)row * h.ld_xa : nullptr);
    }
  }
}

// dvals: n_out * out_dim output gradients of this row; or dz_last != nullptr: the gradient w.r.t. the last
// hidden activation is given directly (a trunk whose consumers were differentiated elsewhere)
inline void ht_ref_bwd_head(const HtHead& h, int row, const float* dvals, float* da_row, const float* dz_last = nullptr) {
  float gin[HT_MAXW], gout[HT_MAXW];
  const int HL = h.hid[h.L - 1];
  const float* zl = h.L == 1 ? h.z0 : h.z[h.L - 1];
  for (int n = 0; n < HL; ++n) {
    float acc = 0.f;
    if (dz_last) acc = dz_last[n];
    else
      for (int k = 0; k < h.n_out; ++k)
        for (int o = 0; o < h.out_dim; ++o) acc = fmaf(dvals[k * h.out_dim + o], h.ow[k][n * h.out_dim + o], acc);
    gin[n] = zl[(long)row * HL + n] > 0.f ? acc : 0.f;
  }
  for (int l = h.L - 1; l >= 1; --l) {
    for (int n = 0; n < h.hid[l]; ++n) h.g[l][(long)row * h.hid[l] + n] = gin[n];
    const float* zp = l == 1 ? h.z0 : h.z[l - 1];
    for (int m = 0; m < h.hid[l - 1]; ++m) {
      float acc = 0.f;
      for (int n = 0; n < h.hid[l]; ++n) acc = fmaf(gin[n], h.w[l][m * h.hid[l] + n], acc);
      gout[m] = zp[(long)row * h.hid[l - 1] + m] > 0.f ? acc : 0.f;
    }
    for (int m = 0; m < h.hid[l - 1]; ++m) gin[m] = gout[m];
  }
  for (int n = 0; n < h.H0; ++n) h.g0[(long)row * h.ldg0 + n] = gin[n];
  if (da_row)
    for (int a = 0; a < h.n_xa; ++a) {
      float acc = 0.f;
      for (int n = 0; n < h.H0; ++n) acc = fmaf(gin[n], h.w0a[a * h.H0 + n], acc);
      da_row[a] = acc;
    }
}

inline void heads_bwd_kernel(HeadsBwdArgs a) {
  if (threadIdx.x != 0) return;
  const int chain = blockIdx.y;
  const float alpha_over_b = expf(a.log_ent_coef[0]) / (float)a.B;
  for (int row = blockIdx.x * HT_RB; row < std::min(a.B, (int)blockIdx.x * HT_RB + HT_RB); ++row) {
    float dv[2 * HT_MAXA];
    if (chain == 0) {
      const HtHead& q = a.h[4];
      dv[0] = ht_dout(a, 4, row, expf(a.log_ent_coef[0]));
      a.d_out[4][(long)row * a.ld_d] = dv[0];
      ht_ref_bwd_head(q, row, dv, a.da_pi + (long)row * a.A);
      ht_sample_bwd_row(a.ls_raw + (long)row * a.A, a.eps + (long)row * a.A, a.pi_a + (long)row * a.A,
                        a.da_pi + (long)row * a.A, a.A, alpha_over_b, a.dmu + (long)row * a.ld_dm, a.dls + (long)row * a.ld_dm);
      for (int j = 0; j < a.A; ++j) { dv[j] = a.dmu[(long)row * a.ld_dm + j]; dv[a.A + j] = a.dls[(long)row * a.ld_dm + j]; }
      ht_ref_bwd_head(a.h[0], row, dv, nullptr);
    } else {
      const HtHead& h = a.h[chain];
      dv[0] = ht_dout(a, chain, row, expf(a.log_ent_coef[0]));
      a.d_out[chain][(long)row * a.ld_d] = dv[0];
      ht_ref_bwd_head(h, row, dv, nullptr);
    }
  }
}

