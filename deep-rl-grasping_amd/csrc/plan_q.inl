// plan_q.inl: launch plan of the DQN / BDQ update incl. prioritised replay (grl_ctx::plan_q) -- part of engine.hip (included there, same translation unit: the plans are methods of grl_ctx).

// --------------------------------------------------------------------------------------------------
// DQN (sb_helper.py:159-165) and BDQ (sb_helper.py:210-224): MLP towers on vector observations.
// One structure covers both: optional shared trunk -> D advantage branches + a state
// value tower, dueling aggregation per branch, double-Q target averaged over branches.
namespace {
struct QNetP {
  std::vector<int64_t> cw, cb;                    // trunk
  std::vector<std::vector<int64_t>> bw, bb;       // [branch][hidden..., out]
  std::vector<int64_t> vw, vb;                    // value tower [hidden..., out]
};
struct QNetAct {
  std::vector<float*> zc;
  std::vector<std::vector<float*>> zb;
  float* adv = nullptr;
  std::vector<float*> zv;
  float* v = nullptr;
};
}  // namespace

int grl_ctx::plan_q() {
  // 1 once the plan ends with the fused reduction + clip + Adam launch: the loss launch then leaves its batch sums to it
  // (no device-scope fence / last-workgroup pass) and the gather launch fixes the Adam step size.  Read at launch time.
  auto q_defer = std::make_shared<int>(0);
  auto q_ga = std::make_shared<GatherArgs>();   // the minibatch gather of this plan (filled below; the prioritised sampler can run it)
  memset(q_ga.get(), 0, sizeof(GatherArgs));
  float* q_row_part = nullptr;
  int q_finish = 0;
  const grl_config& c = cfg;
  cnn = false;
  const int D = c.q_branches, nb = c.q_bins, Lc = c.q_n_common, Lb = c.q_n_branch, Lv = c.q_n_value;
  qD = D; qN = nb;
  A = D; L = 0; B = c.batch_size; NA = std::max(1, c.act_batch);
  img_elems = c.obs_dim; F = c.obs_dim; Fc = 0; ldf = (int)rup(F, 4); C_img = 0; hw = c.img_hw;
  const std::string scope = c.algo == GRL_ALGO_DQN ? "deepq" : "bdq";
  auto fcname = [](int k) { return k == 0 ? std::string("fully_connected") : "fully_connected_" + std::to_string(k); };

  // ---------------- layout (TF creation order of the shipped zips)
  add_var(scope + "/eps:0", {}, false);   // exploration epsilon: stored with the model, never trained
  auto add_net = [&](const std::string& pre, QNetP& P, bool tr) {
    int d = c.obs_dim;
    for (int k = 0; k < Lc; ++k) {
      P.cw.push_back(add_var(pre + "/common_net/" + fcname(k) + "/weights:0", {d, c.q_common[k]}, tr));
      P.cb.push_back(add_var(pre + "/common_net/" + fcname(k) + "/biases:0", {c.q_common[k]}, tr));
      d = c.q_common[k];
    }
    int k = 0;
    P.bw.resize(D); P.bb.resize(D);
    for (int br = 0; br < D; ++br) {
      int dd = d;
      for (int l = 0; l < Lb; ++l, ++k) {
        P.bw[br].push_back(add_var(pre + "/action_value/" + fcname(k) + "/weights:0", {dd, c.q_branch[l]}, tr));
        P.bb[br].push_back(add_var(pre + "/action_value/" + fcname(k) + "/biases:0", {c.q_branch[l]}, tr));
        dd = c.q_branch[l];
      }
      P.bw[br].push_back(add_var(pre + "/action_value/" + fcname(k) + "/weights:0", {dd, nb}, tr));
      P.bb[br].push_back(add_var(pre + "/action_value/" + fcname(k) + "/biases:0", {nb}, tr));
      ++k;
    }
    int dd = d;
    for (int l = 0; l < Lv; ++l) {
      P.vw.push_back(add_var(pre + "/state_value/" + fcname(l) + "/weights:0", {dd, c.q_value[l]}, tr));
      P.vb.push_back(add_var(pre + "/state_value/" + fcname(l) + "/biases:0", {c.q_value[l]}, tr));
      dd = c.q_value[l];
    }
    P.vw.push_back(add_var(pre + "/state_value/" + fcname(Lv) + "/weights:0", {dd, 1}, tr));
    P.vb.push_back(add_var(pre + "/state_value/" + fcname(Lv) + "/biases:0", {1}, tr));
  };
  QNetP Pon, Ptg;
  q_online_off = n_params;
  add_net(scope + "/model", Pon, true);
  q_online_n = n_params - q_online_off;
  n_train = n_params;       // eps sits inside the bucket with a permanently zero gradient
  tgt_off = n_params;
  add_net(scope + "/target_q_func/model", Ptg, false);
  vf_off = 0; n_polyak = 0; ent_off = 0;

  // ---------------- arenas
  params = st.f32(n_params);
  adam_m = st.f32(n_train);
  adam_v = st.f32(n_train);
  sc = (DevScalars*)st.take(sizeof(DevScalars));
  s_mean = (double*)st.take((size_t)img_elems * 8);
  s_std = (double*)st.take((size_t)img_elems * 8);
  s_dmean = (double*)st.take(8); s_dstd = (double*)st.take(8);
  s_ret = (double*)st.take(8);
  grads = gr.f32(n_train);
  const int64_t cap = c.replay_capacity;
  rp_obs = rp.f32(cap * img_elems); rp_next = rp.f32(cap * img_elems);
  rp_dobs = rp.f32(cap); rp_dnext = rp.f32(cap);
  rp_act = rp.f32(cap * A); rp_rew = rp.f32(cap); rp_done = rp.f32(cap);
  stg_n = std::max(NA, 64);
  stg_obs = wk.f32((int64_t)stg_n * c.obs_dim); stg_next = wk.f32((int64_t)stg_n * c.obs_dim);
  stg_act = wk.f32((int64_t)stg_n * A); stg_rew = wk.f32(stg_n); stg_done = wk.f32(stg_n);
  idx_buf = (int64_t*)wk.take((size_t)B * 8);
  eps_buf = wk.f32(std::max(B, B * A));      // importance weights [B]
  for (int n = 0; n < 3; ++n) feat[n] = wk.f32((int64_t)B * ldf);
  act = wk.f32((int64_t)B * A); rew = wk.f32(B); done = wk.f32(B);

  auto alloc_net = [&](QNetAct& a, int rows) {
    for (int k = 0; k < Lc; ++k) a.zc.push_back(wk.f32((int64_t)rows * c.q_common[k]));
    a.zb.resize(D);
    for (int br = 0; br < D; ++br)
      for (int l = 0; l < Lb; ++l) a.zb[br].push_back(wk.f32((int64_t)rows * c.q_branch[l]));
    a.adv = wk.f32((int64_t)rows * D * nb);
    for (int l = 0; l < Lv; ++l) a.zv.push_back(wk.f32((int64_t)rows * c.q_value[l]));
    a.v = wk.f32(rows);
  };
  QNetAct net[3], gact, aact;           // online(s), online(s'), target(s'); gradients; act path
  for (int n = 0; n < 3; ++n) alloc_net(net[n], B);
  alloc_net(gact, B);                   // same shapes: gradient w.r.t. each pre-activation ...
  // ... except the output gradients, which the weight-gradient GEMM fetches 16 bytes at a time: every branch's bins start
  // 16-byte aligned (nbp floats apart), the value gradient has a row stride of 4; the padding stays zero
  const int nbp = (int)rup(nb, 4), ld_dv = 4;
  gact.adv = wk.f32((int64_t)B * D * nbp);
  gact.v = wk.f32((int64_t)B * ld_dv);
  zero_once.push_back({gact.adv, (size_t)B * D * nbp * 4});
  zero_once.push_back({gact.v, (size_t)B * ld_dv * 4});
  for (int n = 0; n < 3; ++n) zero_once.push_back({feat[n], (size_t)B * ldf * 4});   // (row padding [F, ldf) is read by 16-byte loads)
  q_td = wk.f32((int64_t)B * D); q_prio = wk.f32(B);
  const float* P = params;
  const int hdim = Lc > 0 ? c.q_common[Lc - 1] : c.obs_dim;

  // forward stages of one net
  auto fwd_stages = [&](const QNetP& W, const QNetAct& a, const float* x, int ldx, int rows,
                        std::vector<std::vector<IgemmProb>>& st_common, std::vector<std::vector<IgemmProb>>& st_hidden,
                        std::vector<IgemmProb>& st_out) {
    const float* in = x; int ldin = ldx, kin = c.obs_dim;
    st_common.resize(Lc);
    for (int k = 0; k < Lc; ++k) {
      st_common[k].push_back(dense_fwd(in, ldin, kin, nullptr, 0, 0, rows, P + W.cw[k], c.q_common[k], P + W.cb[k],
                                       a.zc[k], c.q_common[k], ACT_RELU));
      in = a.zc[k]; ldin = kin = c.q_common[k];
    }
    const float* h = in; const int ldh = ldin;
    st_hidden.resize(std::max(Lb, Lv));
    for (int br = 0; br < D; ++br) {
      const float* z = h; int ldz = ldh, kz = hdim;
      for (int l = 0; l < Lb; ++l) {
        st_hidden[l].push_back(dense_fwd(z, ldz, kz, nullptr, 0, 0, rows, P + W.bw[br][l], c.q_branch[l],
                                         P + W.bb[br][l], a.zb[br][l], c.q_branch[l], ACT_RELU));
        z = a.zb[br][l]; ldz = kz = c.q_branch[l];
      }
      st_out.push_back(dense_fwd(z, ldz, kz, nullptr, 0, 0, rows, P + W.bw[br][Lb], nb, P + W.bb[br][Lb],
                                 a.adv + br * nb, D * nb, ACT_NONE));
    }
    const float* z = h; int ldz = ldh, kz = hdim;
    for (int l = 0; l < Lv; ++l) {
      st_hidden[l].push_back(dense_fwd(z, ldz, kz, nullptr, 0, 0, rows, P + W.vw[l], c.q_value[l], P + W.vb[l],
                                       a.zv[l], c.q_value[l], ACT_RELU));
      z = a.zv[l]; ldz = kz = c.q_value[l];
    }
    st_out.push_back(dense_fwd(z, ldz, kz, nullptr, 0, 0, rows, P + W.vw[Lv], 1, P + W.vb[Lv], a.v, 1, ACT_NONE));
  };

  // =============================================================== prioritised replay (per_kernels.h)
  per_on = c.q_per != 0;
  if (per_on) {
    memset(&per, 0, sizeof(per));
    per_blocks = (int)((cap + PER_BLK - 1) / PER_BLK);
    per.p = (double*)rp.take((size_t)cap * 8);
    per.bsum = (double*)rp.take((size_t)per_blocks * 8);
    per.bmin = (double*)rp.take((size_t)per_blocks * 8);
    per.st = (PerState*)rp.take(sizeof(PerState));
    per_u = (double*)wk.take((size_t)B * 8);
    per.sc = sc; per.seed = c.seed; per.B = B; per.alpha = c.q_per_alpha; per.eps = c.q_per_eps;
    per.alpha64 = c.q_per_alpha64 != 0.0 ? c.q_per_alpha64 : (double)c.q_per_alpha;
    per.stratified = c.q_per_stratified != 0;
    per.idx_out = idx_buf; per.w_out = eps_buf; per.prio_in = nullptr;   // set below (q_prio)
    for (int mode = 0; mode < 2; ++mode) {
      PerArgs pa = per;
      pa.u = mode ? per_u : nullptr;
      const int nb = per_blocks;
      grl_ctx* self = this;
      for (int with_gather = 0; with_gather < 2; ++with_gather) {
        Op op; op.tag = "per_sample";
        auto sampler = [self, pa, nb, with_gather, q_ga, q_defer](bool block_sums) {
          return [self, pa, nb, with_gather, q_ga, q_defer, block_sums](hipStream_t s) {
            PerArgs q = pa;
            q.prio_in = self->q_prio;
            GatherArgs g = *q_ga;
            g.adam_tick = *q_defer;
            if (block_sums) hipLaunchKernelGGL(per_blocksum_kernel, dim3(nb), dim3(256), 0, s, q);
            hipLaunchKernelGGL(per_sample_kernel, dim3(q.B), dim3(256), 0, s, q, nb, g, with_gather);   // RNG mode: marks rng_used, q_loss ticks
          };
        };
        op.run = sampler(true);
        (with_gather ? (mode ? ops_per_u_g : ops_per_rng_g) : (mode ? ops_per_u : ops_per_rng)).push_back(op);
        if (with_gather && !mode) {     // later updates of a multi-update call: the apply launch of the update before kept the block sums current
          Op oi = op;
          oi.run = sampler(false);
          ops_per_rng_g_inc.push_back(oi);
        }
      }
    }
    {
      grl_ctx* self = this;
      Op op; op.tag = "per_update";
      op.run = [self](hipStream_t s) {
        PerArgs q = self->per;
        q.prio_in = self->q_prio;
        hipLaunchKernelGGL(per_update_kernel, dim3(1), dim3(256), 0, s, q, (const int64_t*)self->idx_buf);
      };
      ops_per_update.push_back(op);
    }
  }

  // =============================================================== RNG (uniform indices; weights = 1)
  {
    Op op; op.tag = "rng";
    RngArgs ra{sc, c.seed, B, 1, idx_buf, eps_buf, eps_buf, 1};   // weights = 1; the loss kernel advances rng_step
    op.run = [ra](hipStream_t s) {
      hipLaunchKernelGGL(rng_kernel, dim3((ra.B + 255) / 256), dim3(256), 0, s, ra);
    };
    ops_rng.push_back(op);
  }
  // =============================================================== forward
  {
    GatherArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.idx = idx_buf; ga.B = B; ga.img_elems = img_elems; ga.n_direct = 0; ga.act_dim = A;
    ga.rp_obs = rp_obs; ga.rp_next = rp_next; ga.rp_dobs = rp_dobs; ga.rp_dnext = rp_dnext;
    ga.rp_act = rp_act; ga.rp_rew = rp_rew; ga.rp_done = rp_done;
    ga.mean = s_mean; ga.stdv = s_std; ga.dmean = s_dmean; ga.dstd = s_dstd; ga.ret_std = s_ret;
    ga.normalize = (c.normalize == 1 || c.normalize == 2); ga.normalize_rew = (c.normalize == 1 || c.normalize == 3);
    ga.clip_obs = c.clip_obs; ga.clip_rew = c.clip_reward; ga.scale_div = 1.f;
    ga.x_obs = feat[0]; ga.x_obs2 = nullptr; ga.x_next = feat[2]; ga.ldx = ldf;
    ga.d_obs0 = ga.d_obs1 = ga.d_next = feat[0]; ga.ldd = ldf;
    ga.act_out = act; ga.ld_act = A; ga.rew_out = rew; ga.done_out = done;
    ga.sc = sc;
    *q_ga = ga;
    Op op; op.tag = "gather_norm";
    op.run = [ga, q_defer](hipStream_t s) {
      GatherArgs g2 = ga;
      g2.adam_tick = *q_defer;        // deferred loss sums: the Adam step size of the update is fixed here, as in the SAC plan
      hipLaunchKernelGGL(gather_norm_kernel, dim3((g2.img_elems + 255) / 256, g2.B, 2), dim3(256), 0, s, g2);
    };
    ops_grads.push_back(op);
  }
  // ---- row-local chains (q_kernels.h) when every width fits the head primitives; else one GEMM launch per layer
  bool fused_q = false;
  {
    bool ok = tune_int("fused_q", 1) && nb <= 64 && Lc + std::max(Lb, Lv) <= GRL_MAX_LAYERS;
    for (int k = 0; k < Lc; ++k) ok = ok && c.q_common[k] <= HT_MAXW;
    for (int l = 0; l < Lb; ++l) ok = ok && c.q_branch[l] <= HT_MAXW;
    for (int l = 0; l < Lv; ++l) ok = ok && c.q_value[l] <= HT_MAXW;
    if (Lc > 0) ok = ok && c.q_common[Lc - 1] <= HT_MAXA;
    fused_q = ok;
  }
  QFusedArgs qf;
  memset(&qf, 0, sizeof(qf));
  // backward chains that form the loss and the weight gradients of their own rows (q_chain.h): five launches per update instead
  // of seven.  Needs the matrix-core chains and the fused reduction + clip + Adam launch (which sums the row-block slabs and
  // the loss's row sums); GRL_TUNE q_chain=0 keeps the loss and weight-gradient launches.
  bool q_chain = false;
  if (fused_q) {
    const QNetP* Wn[3] = {&Pon, &Pon, &Ptg};
    const float* xin[3] = {feat[0], feat[2], feat[2]};
    auto tower_w = [&](const QNetP& W, int tw, int l) { return P + (tw < D ? W.bw[tw][l] : W.vw[l]); };
    auto tower_b = [&](const QNetP& W, int tw, int l) { return P + (tw < D ? W.bb[tw][l] : W.vb[l]); };
    auto tower_hid = [&](int tw, int l) { return tw < D ? c.q_branch[l] : c.q_value[l]; };
    auto tower_z = [&](const QNetAct& a, int tw, int l) { return tw < D ? a.zb[tw][l] : a.zv[l]; };
    std::vector<HtHead> hf, hb;
    std::vector<IgemmProb> l0;
    // matrix-core stages (q_mfma.h) when every width fits their 64-wide shape (GRL_TUNE q_mfma=0 keeps the VALU chains); they
    // also take layer 0 (K = obs_dim <= 128) into the chain: no GEMM launch in front of it (GRL_TUNE q_l0_chain=0 keeps it)
    bool want_mfma = q_mfma_built() && tune_int("q_mfma", 1) != 0 && nb <= QM_W && D + 1 <= QM_MAXP;
    for (int k = 0; k < Lc; ++k) want_mfma = want_mfma && c.q_common[k] <= QM_W;
    for (int l = 0; l < Lb; ++l) want_mfma = want_mfma && c.q_branch[l] <= QM_W;
    for (int l = 0; l < Lv; ++l) want_mfma = want_mfma && c.q_value[l] <= QM_W;
    const bool l0_chain = want_mfma && c.obs_dim <= 2 * QM_W && tune_int("q_l0_chain", 1) != 0;
    for (int n = 0; n < 3; ++n) {
      const QNetP& W = *Wn[n];
      const QNetAct& a = net[n];
      float* u_trunk = nullptr;
      if (Lc > 0 && !l0_chain) {   // layer 0 of the trunk: one GEMM (K = obs_dim), no bias / activation (applied by the chain)
        u_trunk = wk.f32((int64_t)B * c.q_common[0]);
        l0.push_back(dense_fwd(xin[n], ldf, c.obs_dim, nullptr, 0, 0, B, P + W.cw[0], c.q_common[0], nullptr, u_trunk,
                               c.q_common[0], ACT_NONE));
      }
      for (int tw = 0; tw <= D; ++tw) {
        const int Lt = tw < D ? Lb : Lv;
        HtHead h;
        memset(&h, 0, sizeof(h));
        int li = 0;
        if (Lc > 0) {
          h.u = u_trunk; h.ldu = c.q_common[0]; h.b0 = P + W.cb[0]; h.H0 = c.q_common[0];
          if (l0_chain) { h.xa = xin[n]; h.ld_xa = ldf; h.n_xa = c.obs_dim; h.w0a = P + W.cw[0]; }
          h.z0 = tw == 0 ? a.zc[0] : nullptr;             // the trunk is recomputed per tower, stored once
          h.hid[0] = c.q_common[0];
          for (li = 1; li < Lc; ++li) {
            h.w[li] = P + W.cw[li]; h.b[li] = P + W.cb[li]; h.hid[li] = c.q_common[li];
            h.z[li] = tw == 0 ? a.zc[li] : nullptr;
          }
          for (int l = 0; l < Lt; ++l, ++li) {
            h.w[li] = tower_w(W, tw, l); h.b[li] = tower_b(W, tw, l); h.hid[li] = tower_hid(tw, l); h.z[li] = tower_z(a, tw, l);
          }
        } else {      // no trunk: layer 0 of every tower from the GEMM launch
          float* u = nullptr;
          if (l0_chain) { h.xa = xin[n]; h.ld_xa = ldf; h.n_xa = c.obs_dim; h.w0a = tower_w(W, tw, 0); }
          else {
            u = wk.f32((int64_t)B * tower_hid(tw, 0));
            l0.push_back(dense_fwd(xin[n], ldf, c.obs_dim, nullptr, 0, 0, B, tower_w(W, tw, 0), tower_hid(tw, 0), nullptr, u,
                                   tower_hid(tw, 0), ACT_NONE));
          }
          h.u = u; h.ldu = tower_hid(tw, 0); h.b0 = tower_b(W, tw, 0); h.H0 = tower_hid(tw, 0);
          h.z0 = tower_z(a, tw, 0); h.hid[0] = h.H0;
          for (li = 1; li < Lt; ++li) {
            h.w[li] = tower_w(W, tw, li); h.b[li] = tower_b(W, tw, li); h.hid[li] = tower_hid(tw, li); h.z[li] = tower_z(a, tw, li);
          }
        }
        h.L = li;
        h.n_out = 1; h.out_dim = tw < D ? nb : 1;
        h.ow[0] = tower_w(W, tw, Lt); h.ob[0] = tower_b(W, tw, Lt);
        h.out[0] = tw < D ? a.adv + tw * nb : a.v;
        h.ld_out = tw < D ? D * nb : 1;
        hf.push_back(h);
      }
    }
    // backward views of the online net on s
    const QNetAct& a = net[0];
    for (int tw = 0; tw <= D; ++tw) {
      const int Lt = tw < D ? Lb : Lv;
      HtHead h;
      memset(&h, 0, sizeof(h));
      h.H0 = tower_hid(tw, 0); h.L = Lt; h.hid[0] = h.H0;
      h.z0 = tower_z(a, tw, 0); h.g0 = tower_z(gact, tw, 0); h.ldg0 = h.H0;
      if (Lc > 0) { h.n_xa = c.q_common[Lc - 1]; h.w0a = tower_w(Pon, tw, 0); }
      for (int l = 1; l < Lt; ++l) {
        h.w[l] = tower_w(Pon, tw, l); h.hid[l] = tower_hid(tw, l); h.z[l] = tower_z(a, tw, l); h.g[l] = tower_z(gact, tw, l);
      }
      h.n_out = 1; h.out_dim = tw < D ? nb : 1; h.ow[0] = tower_w(Pon, tw, Lt);
      hb.push_back(h);
    }
    qf.B = B; qf.D = D; qf.nb = nb; qf.Ht = Lc > 0 ? c.q_common[Lc - 1] : 0;
    qf.d_adv = gact.adv; qf.d_v = gact.v; qf.nbp = nbp; qf.ld_dv = ld_dv; qf.trunk_scale = c.q_trunk_scale;
    bool trunk_mfma_ok = true;
    std::vector<HtHead> htr;
    if (Lc > 0) {
      HtHead h;
      memset(&h, 0, sizeof(h));
      h.H0 = c.q_common[0]; h.L = Lc; h.z0 = a.zc[0]; h.g0 = gact.zc[0]; h.ldg0 = h.H0; h.hid[0] = h.H0;
      for (int l = 1; l < Lc; ++l) { h.w[l] = P + Pon.cw[l]; h.hid[l] = c.q_common[l]; h.z[l] = a.zc[l]; h.g[l] = gact.zc[l]; }
      htr.push_back(h);
      trunk_mfma_ok = qm_head_ok(h, false);
      qf.dh_part = wk.f32((int64_t)(D + 1) * B * qf.Ht);
    }
    // matrix-core stages (q_mfma.h) when every chain fits their 64-wide shape; GRL_TUNE q_mfma=0 keeps the VALU chains
    bool fits64 = trunk_mfma_ok && nb <= QM_W && D + 1 <= QM_MAXP;       // every backward chain fits the 64-wide stage shape
    for (auto& h : hb) fits64 = fits64 && qm_head_ok(h);
    qf.mfma = want_mfma && fits64;
    for (auto& h : hf) qf.mfma = qf.mfma && qm_head_ok(h, true, true);
    if (want_mfma && !qf.mfma) return fail(GRL_ERR_INVALID, "internal: the Q chains were planned for the matrix-core stages and do not fit them");
    // the chained backward leaves its slabs and row sums to the fused apply launch (below): planned only where that launch is
    // certain to exist -- its LDS (~71 KB) fits the device and no trainable variable outgrows its gradient buffer -- so that a
    // device / shape that offers less keeps the three-launch form instead of failing at the end of the plan
    bool qapply_feasible = tune_int("fused_qapply", 1) != 0 && device_lds_bytes() >= (int)(GRL_QAPPLY_MAX * 4 + 1024 * 4 + 3 * 256 * 4);
    for (auto& v : vars) if (v.trainable && v.numel > GRL_QAPPLY_MAX) qapply_feasible = false;
    q_chain = q_chain_available(qf.mfma != 0, fits64) && tune_int("q_chain", 1) != 0 && qapply_feasible && qc_shape_ok(nb, D, c.obs_dim);
    if (q_chain) {      // nothing reads the trunk's pre-activation gradients from memory any more: its chain keeps them in LDS;
      // the towers' are read back by the weight-gradient workgroups of the trunk launch when there is one (GRL_TUNE q_chain_late=0:
      // every tower chain forms its slabs itself, at its end)
      const bool late = Lc > 0 && tune_int("q_chain_late", 1) != 0;
      if (!late) for (auto& h : hb) { h.g0 = nullptr; for (int l = 0; l < GRL_MAX_LAYERS; ++l) h.g[l] = nullptr; }
      for (auto& h : htr) { h.g0 = nullptr; for (int l = 0; l < GRL_MAX_LAYERS; ++l) h.g[l] = nullptr; }
    }
    qf.fwd = upload_vec(wk, hf);
    qf.bwd_tw = upload_vec(wk, hb);
    if (!htr.empty()) qf.bwd_tr = upload_vec(wk, htr);
    if (getenv("GRL_PLAN_DUMP"))
      fprintf(stderr, "grl plan: q chains       %s%s\n", qf.mfma ? "matrix-core stages (q_mfma.h)" : "VALU stages (q_kernels.h)",
              q_chain ? ", loss + weight gradients inside the backward chains (q_chain.h)" : "");
    if (!l0.empty()) add_launch(ops_grads, "q_l0", 0, l0);
    Op op; op.tag = "q_fwd";
    const QFusedArgs fa = qf;
    op.run = [fa](hipStream_t s) { launch_q_fwd(fa, s); };
    ops_grads.push_back(op);
    {      // the same launch opening the update (Adam step size): the "per_pf" updates below
      QFusedArgs ft = qf;
      ft.tick_sc = sc;
      q_fwd_tick_op = op;
      q_fwd_tick_op.run = [ft](hipStream_t s) { launch_q_fwd(ft, s); };
      have_q_fwd_tick = true;
    }
  } else {
    std::vector<std::vector<IgemmProb>> sc_[3], sh_[3];
    std::vector<IgemmProb> so_[3];
    fwd_stages(Pon, net[0], feat[0], ldf, B, sc_[0], sh_[0], so_[0]);
    fwd_stages(Pon, net[1], feat[2], ldf, B, sc_[1], sh_[1], so_[1]);
    fwd_stages(Ptg, net[2], feat[2], ldf, B, sc_[2], sh_[2], so_[2]);
    auto merged = [&](std::vector<IgemmProb> a, const std::vector<IgemmProb>& b, const std::vector<IgemmProb>& d) {
      a.insert(a.end(), b.begin(), b.end()); a.insert(a.end(), d.begin(), d.end()); return a;
    };
    for (int k = 0; k < Lc; ++k) add_launch(ops_grads, "q_fwd", 0, merged(sc_[0][k], sc_[1][k], sc_[2][k]));
    for (size_t l = 0; l < sh_[0].size(); ++l) add_launch(ops_grads, "q_fwd", 0, merged(sh_[0][l], sh_[1][l], sh_[2][l]));
    add_launch(ops_grads, "q_fwd", 0, merged(so_[0], so_[1], so_[2]));
  }
  QLossArgs q_loss_args;
  memset(&q_loss_args, 0, sizeof(q_loss_args));
  {
    QLossArgs qa;
    qa.B = B; qa.D = D; qa.n = nb; qa.gamma = c.gamma; qa.lr = c.lr; qa.huber = c.q_huber; qa.double_q = c.q_double;
    qa.adv0 = net[0].adv; qa.v0 = net[0].v; qa.adv1 = net[1].adv; qa.adv2 = net[2].adv; qa.v2 = net[2].v;
    qa.act = act; qa.rew = rew; qa.done = done; qa.weights = eps_buf;
    qa.d_adv0 = gact.adv; qa.d_v0 = gact.v; qa.nbp = nbp; qa.ld_dv = ld_dv; qa.td = q_td; qa.priority = q_prio; qa.sc = sc;
    qa.row_part = wk.f32(3 * (int64_t)B);
    qa.counter = (unsigned*)wk.take(16);
    qa.defer_finish = 0;
    qa.loss_sum = (c.algo == GRL_ALGO_BDQ && c.q_loss_sum_branches) ? 1 : 0;
    zero_once.push_back({qa.counter, 16});
    q_row_part = qa.row_part;
    q_finish = q_loss_finishes_itself(qa.n) ? 1 : 0;     // (the one-workgroup fallback for > 64 bins forms its sums itself)
    q_loss_args = qa;
    if (!q_chain) {
      Op op; op.tag = "q_loss";
      op.run = [qa, q_defer](hipStream_t s) {
        QLossArgs q2 = qa;
        q2.defer_finish = *q_defer;
        launch_q_loss(q2, s);
      };
      ops_grads.push_back(op);
    }
  }
  // =============================================================== backward (online net on s)
  {
    const QNetAct& a = net[0];
    if (q_chain) {
      // one slab per row block and variable; the apply launch sums them in row-block order
      const int n_rb = (B + HT_RB - 1) / HT_RB;
      auto layer = [&](int K, int N, int64_t woff, int64_t boff) {
        QcLayer y;
        y.K = K; y.N = N;
        y.dw = wk.f32((int64_t)n_rb * K * N); y.db = wk.f32((int64_t)n_rb * N);
        ReduceDesc r;
        memset(&r, 0, sizeof(r));
        r.src = y.dw; r.splits = n_rb; r.slab_stride = (int64_t)K * N; r.dst = grads + woff; r.n = K * N;
        reduces.push_back(r);
        r.src = y.db; r.slab_stride = N; r.dst = grads + boff; r.n = N;
        reduces.push_back(r);
        return y;
      };
      std::vector<QcHead> ytw;
      for (int tw = 0; tw <= D; ++tw) {
        const int Lt = tw < D ? Lb : Lv;
        const std::vector<int64_t>& W = tw < D ? Pon.bw[tw] : Pon.vw;
        const std::vector<int64_t>& Bv = tw < D ? Pon.bb[tw] : Pon.vb;
        QcHead y;
        memset(&y, 0, sizeof(y));
        y.xin = Lc > 0 ? a.zc[Lc - 1] : feat[0]; y.ld_xin = Lc > 0 ? hdim : ldf;
        int kin = hdim;
        for (int l = 0; l < Lt; ++l) {
          const int n = tw < D ? c.q_branch[l] : c.q_value[l];
          y.lay[l] = layer(kin, n, W[l], Bv[l]);
          kin = n;
        }
        y.lay[Lt] = layer(kin, tw < D ? nb : 1, W[Lt], Bv[Lt]);
        ytw.push_back(y);
      }
      QChainArgs ca;
      memset(&ca, 0, sizeof(ca));
      ca.f = qf; ca.l = q_loss_args; ca.l.defer_finish = 1;
      ca.late = (Lc > 0 && tune_int("q_chain_late", 1) != 0) ? 1 : 0;
      ca.tw = upload_vec(wk, ytw);
      if (Lc > 0) {
        QcHead y;
        memset(&y, 0, sizeof(y));
        y.xin = feat[0]; y.ld_xin = ldf;
        int kin = c.obs_dim;
        for (int k = 0; k < Lc; ++k) { y.lay[k] = layer(kin, c.q_common[k], Pon.cw[k], Pon.cb[k]); kin = c.q_common[k]; }
        ca.tr = upload_vec(wk, std::vector<QcHead>{y});
      }
      Op op; op.tag = "q_bwd";
      op.run = [ca](hipStream_t s) { launch_q_bwd_chain(ca, s); };
      ops_grads.push_back(op);
      if (per_on && Lc > 0 && B <= 1024) {      // the trunk launch carrying the priority write-back + the block-sum refresh ("per_pf")
        grl_ctx* self = this;
        const int rows = B;
        q_bwd_wb_op = op;
        q_bwd_wb_op.run = [ca, self, rows](hipStream_t s) {
          QChainArgs c2 = ca;
          c2.per = self->per;
          c2.per.prio_in = self->q_prio;
          c2.per_idx = (const int64_t*)self->idx_buf;
          c2.per_wb = rows;
          launch_q_bwd_chain(c2, s);
        };
        have_q_bwd_wb = true;
      }
    } else if (fused_q) {
      const QFusedArgs fa = qf;
      Op op; op.tag = "q_bwd";
      op.run = [fa](hipStream_t s) { launch_q_bwd(fa, s); };
      ops_grads.push_back(op);
    } else {
      std::vector<IgemmProb> pr;      // output layers -> last hidden
      for (int br = 0; br < D; ++br)
        pr.push_back(dense_bwd({{gact.adv + br * nbp, D * nbp, nb, P + Pon.bw[br][Lb]}}, B, 0, c.q_branch[Lb - 1],
                               gact.zb[br][Lb - 1], c.q_branch[Lb - 1], a.zb[br][Lb - 1]));
      pr.push_back(dense_bwd({{gact.v, ld_dv, 1, P + Pon.vw[Lv]}}, B, 0, c.q_value[Lv - 1], gact.zv[Lv - 1], c.q_value[Lv - 1],
                             a.zv[Lv - 1]));
      add_launch(ops_grads, "q_bwd", 1, pr);
      for (int l = std::max(Lb, Lv) - 1; l >= 1; --l) {
        std::vector<IgemmProb> p2;
        if (l < Lb)
          for (int br = 0; br < D; ++br)
            p2.push_back(dense_bwd({{gact.zb[br][l], c.q_branch[l], c.q_branch[l], P + Pon.bw[br][l]}}, B, 0,
                                   c.q_branch[l - 1], gact.zb[br][l - 1], c.q_branch[l - 1], a.zb[br][l - 1]));
        if (l < Lv)
          p2.push_back(dense_bwd({{gact.zv[l], c.q_value[l], c.q_value[l], P + Pon.vw[l]}}, B, 0, c.q_value[l - 1],
                                 gact.zv[l - 1], c.q_value[l - 1], a.zv[l - 1]));
        add_launch(ops_grads, "q_bwd", 1, p2);
      }
      if (Lc > 0) {   // into the shared trunk: sum over the D+1 towers in chunks of three reduction parts
        std::vector<BwdPart> towers;
        for (int br = 0; br < D; ++br) towers.push_back({gact.zb[br][0], c.q_branch[0], c.q_branch[0], P + Pon.bw[br][0]});
        towers.push_back({gact.zv[0], c.q_value[0], c.q_value[0], P + Pon.vw[0]});
        for (size_t t0 = 0; t0 < towers.size(); t0 += 3) {
          std::vector<BwdPart> chunk(towers.begin() + t0, towers.begin() + std::min(towers.size(), t0 + 3));
          const bool last = t0 + 3 >= towers.size();
          IgemmProb p = dense_bwd(chunk, B, 0, hdim, gact.zc[Lc - 1], hdim, last ? a.zc[Lc - 1] : nullptr);
          p.accumulate = t0 > 0 ? 1 : 0;
          p.out_scale = c.q_trunk_scale;
          add_launch(ops_grads, "q_bwd", 1, {p});
        }
        for (int k = Lc - 1; k >= 1; --k)
          add_launch(ops_grads, "q_bwd", 1,
                     {dense_bwd({{gact.zc[k], c.q_common[k], c.q_common[k], P + Pon.cw[k]}}, B, 0, c.q_common[k - 1],
                                gact.zc[k - 1], c.q_common[k - 1], a.zc[k - 1])});
      }
    }
    // weight gradients (not here when the chained backward has formed them)
    if (!q_chain) {
    std::vector<IgemmProb> wg;
    // (inputs whose rows are padded to a multiple of 4 floats -- the observations, ldf -- carry the padding columns along:
    //  every problem then fits the vectorised weight-gradient kernel; their slab rows are never reduced)
    auto wgrad = [&](const float* x, int ldx, int kin, const float* g, int ldg, int n, int64_t woff, int64_t boff) {
      IgemmProb p = dense_wgrad(x, ldx, kin, true, g, ldg, n, B, nullptr, 1, ldx >= (int)rup(kin, 4) ? (int)rup(kin, 4) : kin);
      p.c = wk.f32(p.slab_stride * p.split);
      add_wgrad(wg, p, woff, 0, kin, boff);
    };
    const float* in = feat[0]; int ldin = ldf, kin = c.obs_dim;
    for (int k = 0; k < Lc; ++k) {
      wgrad(in, ldin, kin, gact.zc[k], c.q_common[k], c.q_common[k], Pon.cw[k], Pon.cb[k]);
      in = a.zc[k]; ldin = kin = c.q_common[k];
    }
    for (int br = 0; br < D; ++br) {
      const float* z = in; int ldz = ldin, kz = kin;
      for (int l = 0; l < Lb; ++l) {
        wgrad(z, ldz, kz, gact.zb[br][l], c.q_branch[l], c.q_branch[l], Pon.bw[br][l], Pon.bb[br][l]);
        z = a.zb[br][l]; ldz = kz = c.q_branch[l];
      }
      wgrad(z, ldz, kz, gact.adv + br * nbp, D * nbp, nb, Pon.bw[br][Lb], Pon.bb[br][Lb]);
    }
    const float* z = in; int ldz = ldin, kz = kin;
    for (int l = 0; l < Lv; ++l) {
      wgrad(z, ldz, kz, gact.zv[l], c.q_value[l], c.q_value[l], Pon.vw[l], Pon.vb[l]);
      z = a.zv[l]; ldz = kz = c.q_value[l];
    }
    wgrad(z, ldz, kz, gact.v, ld_dv, 1, Pon.vw[Lv], Pon.vb[Lv]);
    add_launch(ops_grads, "q_wgrad", 2, wg);
    }
  }
  {
    std::vector<int2> rt = reduce_tiles();
    d_reduces = upload_vec(wk, reduces);
    int2* d_rt = upload_vec(wk, rt);
    const int ntiles = (int)rt.size();
    ReduceDesc* dr = d_reduces;
    Op op; op.tag = "reduce_slabs";
    LossArgs none;
    memset(&none, 0, sizeof(none));
    op.run = [dr, d_rt, ntiles, none](hipStream_t s) {
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ntiles), dim3(256), 0, s, dr, d_rt, ntiles, none, 0, AdamArgs{}, 0);
    };
    ops_grads.push_back(op);
    // (the in-graph exchange publishes from this reduction: capi.inl, grl_allreduce_connect)
    red_all.tiles = d_rt; red_all.n = ntiles; red_all.has_loss = 0;
    memset(&adam_base, 0, sizeof(adam_base));
    adam_base.params = params; adam_base.grads = grads; adam_base.m = adam_m; adam_base.v = adam_v; adam_base.n_train = n_train; adam_base.sc = sc;
    adam_base.grad_scale = 1.f; adam_base.eps = 1e-8f; adam_base.target = params + tgt_off;
    memset(&loss_args, 0, sizeof(loss_args));
  }
  if (c.q_grad_clip > 0.f) {   // per-variable tf.clip_by_norm, after the data-parallel all-reduce point
    std::vector<VarSeg> segs;
    for (auto& v : vars)
      if (v.trainable) segs.push_back({v.off, v.numel});
    VarSeg* d_segs = upload_vec(wk, segs);
    const int nseg = (int)segs.size();
    float* g = grads; const float clip = c.q_grad_clip;
    Op op; op.tag = "clip_by_norm";
    grl_ctx* self = this;
    // data parallel (grl_apply_grads(1 / W) on the all-reduced SUM of W replicas): clip(mean, c) * W == clip(sum, W c), so the
    // sum is clipped at clip / grad_scale and Adam's 1 / W brings it back -- the mean of the replicas clipped as one gradient
    op.run = [self, g, d_segs, nseg, clip](hipStream_t s) {
      hipLaunchKernelGGL(clip_by_norm_kernel, dim3(nseg), dim3(256), 0, s, g, d_segs, clip / self->grad_scale);
    };
    ops_apply.push_back(op);
  }
  {
    Op op; op.tag = "adam";
    grl_ctx* self = this;
    op.run = [self](hipStream_t s) {
      AdamArgs aa;
      aa.params = self->params; aa.grads = self->grads; aa.m = self->adam_m; aa.v = self->adam_v;
      aa.n_train = self->n_train; aa.sc = self->sc; aa.grad_scale = self->grad_scale; aa.tau = 0.f; aa.eps = 1e-8f;
      aa.src_ofs = 0; aa.n_polyak = 0; aa.target = self->params + self->tgt_off;
      const int blocks = (int)std::min<int64_t>(2048, (self->n_train + 255) / 256);
      hipLaunchKernelGGL(adam_polyak_kernel, dim3(blocks), dim3(256), 0, s, aa);
    };
    ops_apply.push_back(op);
  }
  {
    // Full updates: reduction + clip + Adam as one launch (q_reduce_clip_adam_kernel) when every trainable variable is
    // exactly one reduction descriptor and fits the kernel's LDS buffer.  GRL_TUNE fused_qapply=0 keeps the three launches.
    bool ok = tune_int("fused_qapply", 1) && !ops_grads.empty() && ops_grads.back().tag == "reduce_slabs";
    size_t n_tr = 0;
    for (auto& v : vars) {
      if (!v.trainable) continue;
      ++n_tr;
      int hits = 0;
      for (auto& r : reduces) hits += (r.dst == grads + v.off && r.n == v.numel && r.n <= GRL_QAPPLY_MAX) ? 1 : 0;
      ok = ok && hits == 1;
    }
    ok = ok && n_tr == reduces.size();
    for (auto& r : reduces) ok = ok && r.row_len == 0;     // the kernel reads plain (non-strided) slabs
    // ~71 KB of static LDS per workgroup: fits gfx950's 160 KB; any device that offers less keeps the three launches
    if (ok && device_lds_bytes() < (int)(GRL_QAPPLY_MAX * 4 + 1024 * 4 + 3 * 256 * 4)) ok = false;
    if (getenv("GRL_PLAN_DUMP")) fprintf(stderr, "grl plan: q_apply       reduction + clip + Adam in one launch: %s (%zu variables)\n", ok ? "yes" : "no", n_tr);
    if (ok) {
      ops_grads_apply.assign(ops_grads.begin(), ops_grads.end() - 1);
      Op op; op.tag = "q_apply";
      grl_ctx* self = this;
      const ReduceDesc* dr = d_reduces;
      const int nd = (int)reduces.size();
      const float clip = c.q_grad_clip;
      const float* rp = q_row_part; const int rows = B, fin = q_finish;
      auto qga = q_ga;      // (shared with the samplers: the minibatch gather of this plan)
      const int q_gx = (img_elems + 255) / 256;      // tiles per row of gather_norm_kernel on this plan (scalar form)
      auto apply_op = [self, dr, nd, clip, rp, rows, fin, qga, q_gx](bool with_per, bool refresh = false, bool next_sampler = false, bool next_uniform = false) {
        return [self, dr, nd, clip, rp, rows, fin, with_per, refresh, next_sampler, next_uniform, qga, q_gx](hipStream_t s) {
          AdamArgs aa;
          aa.params = self->params; aa.grads = self->grads; aa.m = self->adam_m; aa.v = self->adam_v;
          aa.n_train = self->n_train; aa.sc = self->sc; aa.grad_scale = self->grad_scale; aa.tau = 0.f; aa.eps = 1e-8f;
          aa.src_ofs = 0; aa.n_polyak = 0; aa.target = self->params + self->tgt_off;
          PerArgs q = self->per;
          q.prio_in = self->q_prio;
          // prioritised replay: one more workgroup writes the new priorities back (per_update_kernel's work)
          // (refresh: one more workgroup per sample rebuilds the block sums its new priority touches, per_refresh_body)
          // (next_sampler: write-back and refresh rode on the trunk launch; `rows` workgroups draw the NEXT update's minibatch)
          const int n_extra = (next_sampler || next_uniform) ? 1 : (with_per ? 2 : 1) + (refresh ? rows : 0);
          QNextArgs nx;
          memset(&nx, 0, sizeof(nx));
          int n_next = 0;
          if (next_sampler) {
            nx.per = q; nx.per.u = nullptr;
            nx.g = *qga; nx.g.adam_tick = 0;          // (the next update's forward launch fixes its Adam step size)
            nx.n_sample = rows; nx.n_blocks = self->per_blocks;
            n_next = rows;
          } else if (next_uniform) {
            // the riders draw row b's index with counter + 1 themselves (the draw of rng_kernel) and leave it in idx_buf; the
            // importance weights stay the ones the call's first update wrote
            nx.g = *qga; nx.g.adam_tick = 0; nx.g.use_rng = 1; nx.g.rng_ahead = 1; nx.g.quiet = 1;
            nx.g.seed = self->cfg.seed; nx.g.idx_w = self->idx_buf; nx.g.n_eps = 0;
            nx.uniform_gx = q_gx;
            n_next = q_gx * nx.g.B * 2;
          }
          // (finish bit 1 = leave the Philox counter alone: the riders of this launch read it; prioritised: the trunk launch
          //  advanced it already; uniform: the next update's forward launch will)
          hipLaunchKernelGGL(q_reduce_clip_adam_kernel, dim3(nd + n_extra + n_next), dim3(1024), 0, s, dr, nd, clip,
                             aa, rp, rows, fin | ((next_sampler || next_uniform) ? 2 : 0), q, (const int64_t*)self->idx_buf, n_extra, nx);
        };
      };
      op.run = apply_op(false);
      ops_grads_apply.push_back(op);
      if (per_on && B <= 1024) {
        for (size_t k = 0; k + 1 < ops_grads_apply.size(); ++k)
          if (ops_grads_apply[k].tag != "gather_norm") ops_grads_apply_per.push_back(ops_grads_apply[k]);   // (the sampler gathers its rows)
        Op po; po.tag = "q_apply";
        po.run = apply_op(true);
        ops_grads_apply_per.push_back(po);
        if (tune_int("per_inc", 1) && !ops_per_rng_g_inc.empty()) {      // multi-update calls on the device RNG (capi: grl_train_step_per)
          ops_grads_apply_per_r = ops_grads_apply_per;
          ops_grads_apply_per_r.back().run = apply_op(true, true);
        }
        // "per_pf": FOUR launches per update inside such a call.  The TD errors are final when the tower chains end, so the
        // priority write-back and the refresh of the block sums ride on the trunk launch (q_chain.h) and the launch that ends
        // update t -- reduction, clip, Adam -- carries the sampler of update t + 1 (sum-tree walks, importance weights, the rows'
        // gather): nothing it writes is read by that launch, the Philox counter was advanced by update t's loss, and the Adam
        // step size of t + 1 is fixed by its forward launch instead of its sampler (the apply launch of t still reads t's).
        //   first : sampler (all block sums) | forward | towers | trunk + write-back + refresh | apply + sampler(t + 1)
        //   middle:                            forward[tick] | towers | trunk + write-back + refresh | apply + sampler(t + 1)
        //   last  :                            forward[tick] | towers | trunk + write-back + refresh | apply
        if (tune_int("per_pf", 1) && q_chain && have_q_fwd_tick && have_q_bwd_wb && !ops_per_rng_g.empty()) {
          ops_per_pf_first = ops_per_rng_g;
          for (int v = 0; v < 3; ++v) {
            std::vector<Op>& dst = v == 0 ? ops_per_pf_first : (v == 1 ? ops_per_pf_mid : ops_per_pf_last);
            for (size_t k = 0; k + 1 < ops_grads_apply_per.size(); ++k) {
              const Op& o = ops_grads_apply_per[k];
              if (o.tag == "q_fwd") dst.push_back(v == 0 ? o : q_fwd_tick_op);
              else if (o.tag == "q_bwd") dst.push_back(q_bwd_wb_op);
              else dst.push_back(o);
            }
            Op ao; ao.tag = "q_apply";
            ao.run = v == 2 ? apply_op(false) : apply_op(false, false, true);
            dst.push_back(ao);
          }
          per_pf_ok = true;
        }
        if (getenv("GRL_PLAN_DUMP"))
          fprintf(stderr, "grl plan: per_pf        multi-update prioritised calls, four launches per update (sampler on the apply launch): %s\n", per_pf_ok ? "yes" : "no");
      }
      // "q_pf": uniform replay, calls of several updates on the device RNG -- FOUR launches per update instead of six: the index draw
      // and the gather of update t + 1 (rng_kernel + gather_norm_kernel) ride on the launch that ends update t, whose forward launch
      // is the first reader of the minibatch tensors; the forward launch of t + 1 opens the update (counter += 1, Adam step size).
      //   first : rng | gather | forward | towers | trunk | apply + draw / gather(t + 1)
      //   middle:                forward[counter += 1, tick] | towers | trunk | apply + draw / gather(t + 1)
      //   last  :                forward[counter += 1, tick] | towers | trunk | apply (counter += 1)
      if (!per_on && tune_int("q_pf", 1) && have_q_fwd_tick && fin && B <= 1024) {
        Op fwd_adv = q_fwd_tick_op;
        {
          QFusedArgs ft = qf;
          ft.tick_sc = sc; ft.tick_rng = 1;
          fwd_adv.run = [ft](hipStream_t s) { launch_q_fwd(ft, s); };
        }
        ops_q_pf_first = ops_rng;
        for (int v = 0; v < 3; ++v) {
          std::vector<Op>& dst = v == 0 ? ops_q_pf_first : (v == 1 ? ops_q_pf_mid : ops_q_pf_last);
          for (size_t k = 0; k + 1 < ops_grads_apply.size(); ++k) {
            const Op& o = ops_grads_apply[k];
            if (o.tag == "gather_norm") { if (v == 0) dst.push_back(o); }
            else if (o.tag == "q_fwd") dst.push_back(v == 0 ? o : fwd_adv);
            else dst.push_back(o);
          }
          Op ao; ao.tag = "q_apply";
          ao.run = v == 2 ? apply_op(false) : apply_op(false, false, false, true);
          dst.push_back(ao);
        }
        q_pf_ok = true;
        if (getenv("GRL_PLAN_DUMP"))
          fprintf(stderr, "grl plan: q_pf          multi-update uniform calls, four launches per update (draw + gather on the apply launch)\n");
      }
      *q_defer = 1;
      if (fin) {     // split compute / apply path of this plan: the batch means as a launch of their own, behind the reduction
        Op fo; fo.tag = "q_finish";
        DevScalars* scp = sc;
        fo.run = [scp, rp, rows](hipStream_t s) { hipLaunchKernelGGL(q_finish_kernel, dim3(1), dim3(256), 0, s, scp, rp, rows); };
        ops_grads.push_back(fo);
      }
    }
  }
  if (q_chain && (!*q_defer || !q_finish))
    return fail(GRL_ERR_INVALID, "internal: the chained Q backward was planned without the fused apply launch that sums its slabs and row sums");
  // =============================================================== act path: Q-values of n observations
  {
    afeat = wk.f32((int64_t)NA * ldf);
    q_aout = wk.f32((int64_t)NA * D * nb);
    a_eps = wk.f32(NA); a_out = q_aout;
    alloc_net(aact, NA);
    ActIngestArgs ia;
    memset(&ia, 0, sizeof(ia));
    ia.obs = stg_obs; ia.n = NA; ia.vec_dim = c.obs_dim; ia.scale_div = 1.f; ia.x = afeat; ia.ldx = ldf; ia.d = afeat; ia.ldd = ldf;
    Op op; op.tag = "act_ingest";
    const int elems = c.obs_dim;
    op.run = [ia, elems](hipStream_t s) {
      hipLaunchKernelGGL(act_ingest_kernel, dim3((elems + 255) / 256, ia.n), dim3(256), 0, s, ia);
    };
    ops_act.push_back(op);
    std::vector<std::vector<IgemmProb>> sc1, sh1;
    std::vector<IgemmProb> so1;
    fwd_stages(Pon, aact, afeat, ldf, NA, sc1, sh1, so1);
    for (auto& v : sc1) add_launch(ops_act, "act_q", 0, v);
    for (auto& v : sh1) add_launch(ops_act, "act_q", 0, v);
    add_launch(ops_act, "act_q", 0, so1);
    const float* adv = aact.adv; const float* vv = aact.v; float* qo = q_aout; const int rows = NA, Dq = D, nq = nb;
    // the last launch also writes the Q-values to coherent host memory and counts its workgroups there: grl_act polls the
    // counter instead of copying device -> host and synchronising the stream (GRL_TUNE act_poll=0 keeps that)
    if (!dry && tune_int("act_poll", 1) != 0 &&
        hipHostMalloc((void**)&q_act_host, (size_t)NA * D * nb * 4, hipHostMallocCoherent) == hipSuccess) {
      if (hipHostMalloc((void**)&act_done_host, 64, hipHostMallocCoherent) == hipSuccess) {
        *act_done_host = 0u;
        act_done_wgs = (unsigned)((rows * Dq + 255) / 256);
      } else {
        (void)hipHostFree(q_act_host);
        q_act_host = nullptr; act_done_host = nullptr;
      }
    }
    float* qh = act_done_wgs ? q_act_host : nullptr;
    unsigned* dn = act_done_wgs ? act_done_host : nullptr;
    Op op2; op2.tag = "dueling";
    op2.run = [adv, vv, qo, rows, Dq, nq, qh, dn](hipStream_t s) {
      hipLaunchKernelGGL(dueling_kernel, dim3((rows * Dq + 255) / 256), dim3(256), 0, s, adv, vv, rows, Dq, nq, qo, qh, dn);
    };
    ops_act.push_back(op2);
  }
  for (int k = 0; k < 8; ++k) enc_w[k] = nullptr;
  dbg["feat_pi"] = {feat[0], (int64_t)B * ldf};
  dbg["feat_tgt"] = {feat[2], (int64_t)B * ldf};
  dbg["adv"] = {net[0].adv, (int64_t)B * D * nb};
  dbg["v"] = {net[0].v, B};
  dbg["adv_next"] = {net[1].adv, (int64_t)B * D * nb};
  dbg["adv_tgt"] = {net[2].adv, (int64_t)B * D * nb};
  dbg["v_tgt"] = {net[2].v, B};
  dbg["td"] = {q_td, (int64_t)B * D};
  dbg["idx_raw"] = {(const float*)idx_buf, (int64_t)2 * B};     // int64 viewed as float pairs
  dbg["weights"] = {eps_buf, B};
  if (per_on) dbg["per_p"] = {(const float*)per.p, 2 * cap};   // float64 leaves, handed out as raw 4-byte words
  dbg["priority"] = {q_prio, B};
  dbg["rew"] = {rew, B}; dbg["done"] = {done, B}; dbg["act"] = {act, (int64_t)B * A};
  dbg["grads"] = {grads, n_train};
  dbg["adam_m"] = {adam_m, n_train};
  dbg["adam_v"] = {adam_v, n_train};
  return GRL_OK;
}
