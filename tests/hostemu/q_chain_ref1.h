// q_chain_ref1.h -- TEST-ONLY reference form of the kernels of csrc/q_chain.h (sequential loops over the same descriptors),
// included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).
// Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.

// backward of one head for one row, keeping the gradient rows: g[l][n] = d loss / d pre-activation of layer l (also stored to
// h.g0 / h.g[l] where the descriptor names a tensor).  dvals: the head's output gradients; or dz_last: the gradient w.r.t. the
// last hidden activation (a trunk)
inline void qc_ref_bwd_row(const HtHead& h, int row, const float* dvals, const float* dz_last, float* da_row, float (*g)[HT_MAXW]) {
  float gin[HT_MAXW], gout[HT_MAXW];
  const int HL = h.hid[h.L - 1];
  const float* zl = h.L == 1 ? h.z0 : h.z[h.L - 1];
  for (int n = 0; n < HL; ++n) {
    float acc = 0.f;
    if (dz_last) acc = dz_last[n];
    else
      for (int o = 0; o < h.out_dim; ++o) acc = fmaf(dvals[o], h.ow[0][n * h.out_dim + o], acc);
    gin[n] = zl[(long)row * HL + n] > 0.f ? acc : 0.f;
  }
  for (int l = h.L - 1; l >= 1; --l) {
    for (int n = 0; n < h.hid[l]; ++n) {
      g[l][n] = gin[n];
      if (h.g[l]) h.g[l][(long)row * h.hid[l] + n] = gin[n];
    }
    const float* zp = l == 1 ? h.z0 : h.z[l - 1];
    for (int m = 0; m < h.hid[l - 1]; ++m) {
      float acc = 0.f;
      for (int n = 0; n < h.hid[l]; ++n) acc = fmaf(gin[n], h.w[l][m * h.hid[l] + n], acc);
      gout[m] = zp[(long)row * h.hid[l - 1] + m] > 0.f ? acc : 0.f;
    }
    for (int m = 0; m < h.hid[l - 1]; ++m) gin[m] = gout[m];
  }
  for (int n = 0; n < h.H0; ++n) {
    g[0][n] = gin[n];
    if (h.g0) h.g0[(long)row * h.ldg0 + n] = gin[n];
  }
  if (da_row)
    for (int a = 0; a < h.n_xa; ++a) {
      float acc = 0.f;
      for (int n = 0; n < h.H0; ++n) acc = fmaf(gin[n], h.w0a[a * h.H0 + n], acc);
      da_row[a] = acc;
    }
}
// the slabs of row block rb: zeroed, then one row after the other added (dW[k][n] += x[k] g[n], db[n] += g[n])
inline void qc_ref_zero(const QcHead& y, int rb, int n_layers) {
  for (int l = 0; l < n_layers; ++l) {
    const QcLayer& yl = y.lay[l];
    for (long e = 0; e < (long)yl.K * yl.N; ++e) yl.dw[(long)rb * yl.K * yl.N + e] = 0.f;
    for (int n = 0; n < yl.N; ++n) yl.db[(long)rb * yl.N + n] = 0.f;
  }
}
inline void qc_ref_add(const QcLayer& yl, int rb, const float* x, const float* g) {
  float* dw = yl.dw + (long)rb * yl.K * yl.N;
  for (int k = 0; k < yl.K; ++k)
    for (int n = 0; n < yl.N; ++n) dw[(long)k * yl.N + n] = fmaf(x[k], g[n], dw[(long)k * yl.N + n]);
  for (int n = 0; n < yl.N; ++n) yl.db[(long)rb * yl.N + n] += g[n];
}
// every layer of one head for one row (g rows as qc_ref_bwd_row leaves them; out_g: the output gradients of a tower, or nullptr)
inline void qc_ref_add_row(const HtHead& h, const QcHead& y, int rb, int row, float (*g)[HT_MAXW], const float* out_g) {
  for (int l = 0; l < h.L; ++l) {
    const float* x = l == 0 ? y.xin + (long)row * y.ld_xin : (l == 1 ? h.z0 : h.z[l - 1]) + (long)row * h.hid[l - 1];
    qc_ref_add(y.lay[l], rb, x, g[l]);
  }
  if (out_g) qc_ref_add(y.lay[h.L], rb, (h.L == 1 ? h.z0 : h.z[h.L - 1]) + (long)row * h.hid[h.L - 1], out_g);
}
inline void qc_ref_out_grads(const QFusedArgs& a, int tw, int row, float* dv) {
  for (int o = 0; o < HT_MAXW; ++o) dv[o] = 0.f;
  if (tw < a.D)
    for (int o = 0; o < a.nb; ++o) dv[o] = a.d_adv[((long)row * a.D + tw) * a.nbp + o];
  else dv[0] = a.d_v[(long)row * a.ld_dv];
}

inline void q_bwd_towers_chain_kernel(QChainArgs ca) {
  if (threadIdx.x != 0) return;
  const QFusedArgs& a = ca.f;
  const int rb = blockIdx.x, tw = blockIdx.y;
  const HtHead& h = a.bwd_tw[tw];
  const QcHead& y = ca.tw[tw];
  if (!ca.late) qc_ref_zero(y, rb, h.L + 1);
  for (int row = rb * HT_RB; row < std::min(a.B, rb * HT_RB + HT_RB); ++row) {
    // the loss of the row, formed by every tower's workgroup (the same values from each); the value tower's keeps the row sums
    float s3[3] = {0.f, 0.f, 0.f};
    q_loss_row(ca.l, row, s3);
    if (tw == a.D) { ca.l.row_part[3 * row] = s3[0]; ca.l.row_part[3 * row + 1] = s3[1]; ca.l.row_part[3 * row + 2] = s3[2]; }
    float dv[HT_MAXW], g[GRL_MAX_LAYERS][HT_MAXW];
    qc_ref_out_grads(a, tw, row, dv);
    qc_ref_bwd_row(h, row, dv, nullptr, h.n_xa ? a.dh_part + ((long)tw * a.B + row) * a.Ht : nullptr, g);
    if (!ca.late) qc_ref_add_row(h, y, rb, row, g, dv);
  }
}
inline void q_bwd_trunk_chain_kernel(QChainArgs ca) {
  if (threadIdx.x != 0) return;
  const QFusedArgs& a = ca.f;
  const int rb = blockIdx.x;
  const int y_chain = ca.late ? a.D + 2 : 1;
  if (ca.per_wb > 0 && (int)blockIdx.y >= y_chain) {      // prioritised replay: write-back and block-sum refresh ride on this launch
    if ((int)blockIdx.y == y_chain) {
      if (rb == 0) { per_update_ref(ca.per, ca.per_idx); q_rng_tick(ca.per.sc); }
      return;
    }
    const int k = ((int)blockIdx.y - y_chain - 1) * (int)gridDim.x + rb;
    if (k < ca.per_wb) per_refresh_ref(ca.per, ca.per_idx, k);
    return;
  }
  if (blockIdx.y > 0) {        // the weight gradients of tower blockIdx.y - 1 from the gradient rows its chain stored
    const int tw = blockIdx.y - 1;
    const HtHead& h = a.bwd_tw[tw];
    const QcHead& y = ca.tw[tw];
    qc_ref_zero(y, rb, h.L + 1);
    for (int row = rb * HT_RB; row < std::min(a.B, rb * HT_RB + HT_RB); ++row) {
      float dv[HT_MAXW], g[GRL_MAX_LAYERS][HT_MAXW];
      qc_ref_out_grads(a, tw, row, dv);
      for (int l = 0; l < h.L; ++l)
        for (int n = 0; n < h.hid[l]; ++n) g[l][n] = l == 0 ? h.g0[(long)row * h.ldg0 + n] : h.g[l][(long)row * h.hid[l] + n];
      qc_ref_add_row(h, y, rb, row, g, dv);
    }
    return;
  }
  const HtHead& h = *a.bwd_tr;
  const QcHead& y = *ca.tr;
  qc_ref_zero(y, rb, h.L);
  for (int row = rb * HT_RB; row < std::min(a.B, rb * HT_RB + HT_RB); ++row) {
    float dz[HT_MAXW], g[GRL_MAX_LAYERS][HT_MAXW];
    for (int n = 0; n < a.Ht; ++n) {
      float s = 0.f;
      for (int p = 0; p <= a.D; ++p) s += a.dh_part[((long)p * a.B + row) * a.Ht + n];
      dz[n] = s * a.trunk_scale;
    }
    qc_ref_bwd_row(h, row, nullptr, dz, nullptr, g);
    qc_ref_add_row(h, y, rb, row, g, nullptr);
  }
}
