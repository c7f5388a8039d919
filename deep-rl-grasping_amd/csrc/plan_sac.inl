// plan_sac.inl: launch plan of the SAC update, the act path and the encoder forward (grl_ctx::plan_sac) -- part of engine.hip (included there, same translation unit: the plans are methods of grl_ctx).

int grl_ctx::plan_sac() {
  const grl_config& c = cfg;
  cnn = c.extractor != GRL_EXTRACTOR_MLP;
  A = c.act_dim; L = c.n_layers; B = c.batch_size; NA = std::max(1, c.act_batch);
  for (int l = 0; l < L; ++l) hid[l] = c.layers[l];
  hw = c.img_hw;
  if (cnn) {
    const int nd = c.extractor == GRL_EXTRACTOR_AUGMENTED ? c.n_direct : 0;
    C_img = c.obs_channels - (nd > 0 ? 1 : 0);
    img_elems = hw * hw * C_img;
    F = 512 + nd; Fc = 512;
  } else {
    C_img = 0; img_elems = c.obs_dim; F = c.obs_dim; Fc = 0;
  }
  const int nd = cnn ? F - 512 : 0;
  ldf = (int)rup(F, 4);
  build_layout();

  // ---------------- state arena
  params = st.f32(n_params);
  adam_m = st.f32(n_train);
  adam_v = st.f32(n_train);
  sc = (DevScalars*)st.take(sizeof(DevScalars));
  n_count = (double*)st.take(16);
  s_mean = (double*)st.take((size_t)img_elems * 8);
  s_std = (double*)st.take((size_t)img_elems * 8);
  s_dmean = (double*)st.take((size_t)std::max(nd, 1) * 8);
  s_dstd = (double*)st.take((size_t)std::max(nd, 1) * 8);
  s_ret = (double*)st.take(8);
  n_elems = cnn ? (int64_t)hw * hw * c.obs_channels : c.obs_dim;
  n_mean = (double*)st.take((size_t)n_elems * 8);     // (directly behind s_ret: grl_set_obs_stats uploads the span in one copy)
  n_var = (double*)st.take((size_t)n_elems * 8);
  grads = gr.f32(n_train);

  // ---------------- replay arena
  const int64_t cap = c.replay_capacity;
  // stored observation: img_elems floats, or (replay_rgb_u8) one packed colour dword + one depth float per pixel
  const int64_t obs_store = c.replay_rgb_u8 ? 2 * (int64_t)hw * hw : img_elems;
  rp_obs = rp.f32(cap * obs_store);
  rp_next = rp.f32(cap * obs_store);
  rp_dobs = rp.f32(cap * std::max(nd, 1));
  rp_dnext = rp.f32(cap * std::max(nd, 1));
  rp_act = rp.f32(cap * A);
  rp_rew = rp.f32(cap);
  rp_done = rp.f32(cap);

  // ---------------- staging (host-facing calls)
  const int64_t obs_elems = cnn ? (int64_t)hw * hw * c.obs_channels : c.obs_dim;
  stg_n = std::max(NA, 64);
  stg_obs = wk.f32(stg_n * obs_elems);
  stg_next = wk.f32(stg_n * obs_elems);
  stg_act = wk.f32((int64_t)stg_n * A);
  stg_rew = wk.f32(stg_n);
  stg_done = wk.f32(stg_n);
  n_stage = wk.f32(stg_n * obs_elems);
  // observed observations (grl_observe): the newest env step's and the one before it, uploaded once each; terminal rows
  // and the small per-step arrays of grl_replay_add_observed
  ob_latest = wk.f32(stg_n * obs_elems);
  ob_prev = wk.f32(stg_n * obs_elems);
  ob_term = wk.f32(stg_n * obs_elems);
  ob_elems = obs_elems;

  // ---------------- training workspace
  idx_buf = (int64_t*)wk.take((size_t)B * 8);
  eps_buf = wk.f32((int64_t)B * A);
  for (int n = 0; n < 3; ++n) {
    feat[n] = wk.f32((int64_t)B * ldf);
    zero_once.push_back({feat[n], (size_t)B * ldf * 4});   // row padding [F, ldf) is read by 16-byte loads
  }
  if (cnn) {
    x_obs = wk.f32((int64_t)B * img_elems);
    x_next = wk.f32((int64_t)B * img_elems);
    x_obs_b = tune_int("gather_ride", 1) != 0 ? wk.f32((int64_t)B * img_elems) : nullptr;     // second image buffer ("gather_ride" below)
    // Layer-1 activations of the two TRAINED networks (and their gradients below) sit side by side, pixel stride 64:
    // pi in columns 0..31, values_fn in 32..63.  Both networks read the same observations, so conv1's weight
    // gradient becomes ONE product obs-patches^T x [dY_pi | dY_vf] (N = 64: full 64x64 tiles, the gathered patches
    // read once) instead of two half-empty ones.  The target network's buffer keeps the stride (columns 32..63 idle)
    // so that one set of conv2 tables serves all three.
    ld1 = 64;
    float* a1_pair = wk.f32((int64_t)B * 225 * 64);
    for (int n = 0; n < 3; ++n) {
      a1[n] = n < 2 ? a1_pair + 32 * n : wk.f32((int64_t)B * 225 * 64);
      a2[n] = wk.f32((int64_t)B * 36 * 64);
      a3[n] = wk.f32((int64_t)B * 16 * 64);
    }
  }
  act = wk.f32((int64_t)B * A); rew = wk.f32(B); done = wk.f32(B);
  alloc_head(hPI, B, 2, A); alloc_head(hVF, B, 1, 1); alloc_head(hQF1, B, 1, 1); alloc_head(hQF2, B, 1, 1);
  alloc_head(hTGT, B, 1, 1); alloc_head(hQF1PI, B, 1, 1); alloc_head(hQF2PI, B, 1, 1);
  pi_a = wk.f32((int64_t)B * A); logp = wk.f32(B); ent = wk.f32(B);
  {
    const char* nf = getenv("GRL_NO_FUSED_HEADS");
    fused_heads = !(nf && nf[0] == '1') && A <= HT_MAXA && (hid[0] % 4) == 0;
    for (int l = 0; l < L; ++l) fused_heads = fused_heads && hid[l] <= HT_MAXW;
  }
  Ap = (int)rup(A, 4);
  if (fused_heads) {
    // output gradients packed / row-padded so that the weight-gradient GEMM can fetch them 16 bytes at a time
    float** dd[4] = {&d_v, &d_qf1, &d_qf2, &d_qf1pi};
    for (auto* q : dd) {
      *q = wk.f32((int64_t)B * 4);
      zero_once.push_back({*q, (size_t)B * 16});
    }
    ld_d = 4;
    act_p = wk.f32((int64_t)B * Ap);
    zero_once.push_back({act_p, (size_t)B * Ap * 4});
  } else {
    d_qf1 = wk.f32(B); d_qf2 = wk.f32(B); d_v = wk.f32(B); d_qf1pi = wk.f32(B); ld_d = 1;
  }
  if (fused_heads) {
    {
      l0_split = std::max(1, tune_int("l0_split", 3));       // reduction of the layer-0 GEMM (K = 513) cut into partial sums
      IgemmProb probe = blank();
      probe.M = B; probe.N = hid[0]; probe.K = F;
      set_split(probe, l0_split);
      l0_split = probe.split;
    }
    for (int k = 0; k < 5; ++k) u_l0[k] = wk.f32((int64_t)B * hid[0] * l0_split);
    g0cat = wk.f32((int64_t)B * 3 * hid[0]);
    alloc_hgrad(gPI, B);
    alloc_hgrad(gVF, B, g0cat, 3 * hid[0]);
    alloc_hgrad(gQF1, B, g0cat + hid[0], 3 * hid[0]);
    alloc_hgrad(gQF2, B, g0cat + 2 * hid[0], 3 * hid[0]);
    alloc_hgrad(gQF1PI, B);
  } else {
    alloc_hgrad(gPI, B); alloc_hgrad(gVF, B); alloc_hgrad(gQF1, B); alloc_hgrad(gQF2, B); alloc_hgrad(gQF1PI, B);
  }
  ld_dm = fused_heads ? Ap : A;
  da_pi = wk.f32((int64_t)B * A); dmu = wk.f32((int64_t)B * ld_dm); dls = wk.f32((int64_t)B * ld_dm);
  if (fused_heads) {
    zero_once.push_back({dmu, (size_t)B * ld_dm * 4});
    zero_once.push_back({dls, (size_t)B * ld_dm * 4});
  }
  if (cnn) {
    float* g1_pair = wk.f32((int64_t)B * 225 * 64);
    for (int n = 0; n < 2; ++n) {
      dfeat[n] = wk.f32((int64_t)B * ldf);
      g3[n] = wk.f32((int64_t)B * 16 * 64);
      g2[n] = wk.f32((int64_t)B * 36 * 64);
      g1[n] = g1_pair + 32 * n;
    }
  }

  const float* P = params;
  const float* T = params;   // target block uses absolute offsets too

  // =============================================================== minibatch: replay gather (+ device RNG)
  {
    GatherArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.idx = idx_buf; ga.B = B; ga.img_elems = img_elems; ga.n_direct = nd; ga.act_dim = A;
    ga.rp_obs = rp_obs; ga.rp_next = rp_next; ga.rp_dobs = rp_dobs; ga.rp_dnext = rp_dnext;
    ga.rp_act = rp_act; ga.rp_rew = rp_rew; ga.rp_done = rp_done;
    ga.mean = s_mean; ga.stdv = s_std; ga.dmean = s_dmean; ga.dstd = s_dstd; ga.ret_std = s_ret;
    ga.normalize = (c.normalize == 1 || c.normalize == 2); ga.normalize_rew = (c.normalize == 1 || c.normalize == 3);
    ga.clip_obs = c.clip_obs; ga.clip_rew = c.clip_reward;
    ga.scale_div = cnn ? 255.f : 1.f;
    if (cnn) {
      ga.x_obs = x_obs; ga.x_obs2 = nullptr; ga.x_next = x_next; ga.ldx = img_elems;
      ga.d_obs0 = feat[0] + 512; ga.d_obs1 = feat[1] + 512; ga.d_next = feat[2] + 512; ga.ldd = ldf;
    } else {
      ga.x_obs = feat[0]; ga.x_obs2 = feat[1]; ga.x_next = feat[2]; ga.ldx = ldf;
      ga.d_obs0 = ga.d_obs1 = ga.d_next = feat[0]; ga.ldd = ldf;   // n_direct == 0: never written
    }
    ga.act_out = act; ga.ld_act = A; ga.rew_out = rew; ga.done_out = done;
    ga.act_out2 = act_p; ga.ld_act2 = Ap;
    ga.rgb_u8 = c.replay_rgb_u8;
    ga.sc = sc; ga.seed = c.seed; ga.idx_w = idx_buf; ga.eps_w = eps_buf; ga.n_eps = A;
    ga.adam_tick = fused_heads ? 1 : 0;   // otherwise sac_loss_kernel fixes the step size
    ga.vec4 = elem_vec4_built() && (img_elems % 4 == 0) && (ga.ldx % 4 == 0);
    const int per_block = ga.vec4 ? 1024 : 256;
    // rows per workgroup of the grouped form (elem_kernels.h: gather_norm_rows_body); GRL_TUNE gather_rows=1 keeps one row each
    {
      int rows = tune_int("gather_rows", c.replay_rgb_u8 ? GATHER_ROWS_U8 : GATHER_ROWS_F32);
      if (!ga.vec4 || (rows != 2 && rows != 4) || B % rows) rows = 1;
      ga.rows = rows;
    }
    pf_ga = ga;
    pf_gx = (ga.img_elems + per_block - 1) / per_block;
    for (int mode = 0; mode < 2; ++mode) {
      ga.use_rng = mode;
      Op op; op.tag = "gather_norm";
      op.bytes = 2.0 * B * ((double)img_elems * 4 + (double)obs_store * 4 + 4.0 * nd) + B * (4.0 * A + 8) * 2;
      const int gx0 = pf_gx;
      op.run = [ga, gx0](hipStream_t s) {
        if (ga.rows > 1) hipLaunchKernelGGL(gather_norm_lin_kernel, dim3(gather_blocks(ga, gx0)), dim3(256), 0, s, ga, gx0);
        else hipLaunchKernelGGL(gather_norm_kernel, dim3(gx0, ga.B, 2), dim3(256), 0, s, ga);
      };
      (mode ? ops_rng : ops_gather).push_back(op);
    }
  }

  ConvGeom cg[3];
  ConvFwdTabs ft[3];
  if (cnn) {
    for (int l = 0; l < 3; ++l) {
      cg[l] = cnn_geom(l, C_img);
      if (l == 0) cg[l].ldy = ld1;
      if (l == 1) cg[l].ldx = ld1;
      ft[l] = conv_fwd_tabs(cg[l], B);
    }
    const float* xin[3] = {x_obs, x_obs, x_next};
    // conv1 -> conv2 -> conv3 of the three networks: ONE sample-local launch (conv_stack.h; a workgroup owns a sample of a
    // network and keeps the activation chain in LDS) for 1 / 2 / 4 image channels; GRL_TUNE conv_stack=0 or any other channel
    // count: one implicit-GEMM launch per layer.  The target network's layer-1 / layer-2 activations are not stored.
    {
      const char* nv2 = getenv("GRL_NO_V2");      // (the scalar-gather fallback test runs the per-layer launches on igemm_kernel)
      conv_stack = conv_stack_ok(C_img) && hw == 64 && tune_int("conv_stack", 1) != 0 && !(nv2 && nv2[0] == '1');
    }
    if (conv_stack) {
      ConvStackArgs ca;
      memset(&ca, 0, sizeof(ca));
      for (int n = 0; n < 3; ++n) {
        ConvStackNet& cn = ca.nets[n];
        cn.x = xin[n];
        for (int l = 0; l < 3; ++l) { cn.w[l] = P + ex[n].w[l]; cn.b[l] = P + ex[n].b[l]; }
        cn.a1 = n < 2 ? a1[n] : nullptr; cn.a2 = n < 2 ? a2[n] : nullptr; cn.a3 = a3[n]; cn.ld1 = ld1;
      }
      ca.B = B; ca.n_nets = 3;
      const int Ci = C_img;
      Op op; op.tag = "conv_stack_fwd";
      op.flops = op.flops_exec = 2.0 * B * 3 * (225.0 * 32 * 64 * Ci + 36.0 * 64 * 512 + 16.0 * 64 * 576);
      op.run = [ca, Ci](hipStream_t s) { launch_conv_stack_fwd(Ci, ca, s); };
      ops_grads.push_back(op);
      ride_conv_args = ca;
    } else {
    const char* tags[3] = {"conv1_fwd", "conv2_fwd", "conv3_fwd"};
    for (int l = 0; l < 3; ++l) {
      std::vector<IgemmProb> pr;
      for (int n = 0; n < 3; ++n) {
        const float* in = l == 0 ? xin[n] : (l == 1 ? a1[n] : a2[n]);
        float* out = l == 0 ? a1[n] : (l == 1 ? a2[n] : a3[n]);
        pr.push_back(conv_fwd(in, ft[l], cg[l], P + ex[n].w[l], P + ex[n].b[l], out, ACT_RELU, 0.f));
      }
      add_launch(ops_grads, tags[l], 0, pr);
    }
    }
    std::vector<IgemmProb> pr;
    for (int n = 0; n < 3; ++n)
      pr.push_back(dense_fwd(a3[n], 1024, 1024, nullptr, 0, 0, B, P + ex[n].fw, 512, P + ex[n].fb, feat[n], ldf,
                             ACT_RELU));
    add_launch(ops_grads, "fc_fwd", 0, pr);
  }
  (void)T;

  // description of one head for the row-local kernels (heads_kernels.h)
  auto mk_head = [&](const MlpP& m, const HeadAct& h, const HeadGrad* g, const float* u, const float* xa, int ld_xa,
                     int n_xa) {
    HtHead H;
    memset(&H, 0, sizeof(H));
    H.u = u; H.ldu = hid[0]; H.u_split = l0_split; H.u_stride = (long)B * hid[0];
    H.xa = xa; H.ld_xa = ld_xa; H.n_xa = n_xa;
    H.w0a = P + m.w[0] + (int64_t)F * hid[0];
    H.b0 = P + m.b[0]; H.z0 = h.z[0]; H.H0 = hid[0]; H.L = L;
    for (int l = 0; l < L; ++l) H.hid[l] = hid[l];
    for (int l = 1; l < L; ++l) { H.w[l] = P + m.w[l]; H.b[l] = P + m.b[l]; H.z[l] = h.z[l]; }
    if (g) {
      H.g0 = g->g[0]; H.ldg0 = g->ld0;
      for (int l = 1; l < L; ++l) H.g[l] = g->g[l];
    }
    H.n_out = m.n_out; H.out_dim = m.out_dim;
    for (int k = 0; k < m.n_out; ++k) { H.ow[k] = P + m.ow[k]; H.ob[k] = P + m.ob[k]; H.out[k] = h.out[k]; }
    return H;
  };
  if (fused_heads) {
    // layer 0, feature part only (no bias, no activation): u = feat . W0[0:F]
    const MlpP* ms[5] = {&m_pi, &m_vf, &m_qf1, &m_qf2, &m_tgt};
    const float* fin[5] = {feat[0], feat[1], feat[1], feat[1], feat[2]};
    std::vector<IgemmProb> pr;
    for (int k = 0; k < 5; ++k) {
      IgemmProb p = dense_fwd(fin[k], ldf, F, nullptr, 0, 0, B, P + ms[k]->w[0], hid[0], nullptr, u_l0[k], hid[0], ACT_NONE);
      set_split(p, l0_split);     // partial sums [split][B, H0]: the head chains add them (40 -> 120 workgroups)
      pr.push_back(p);
    }
    add_launch(ops_grads, "heads_l0", 0, pr);
    HeadsFwdArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.h[0] = mk_head(m_pi, hPI, nullptr, u_l0[0], nullptr, 0, 0);
    fa.h[1] = mk_head(m_vf, hVF, nullptr, u_l0[1], nullptr, 0, 0);
    fa.h[2] = mk_head(m_qf1, hQF1, nullptr, u_l0[2], act, A, A);
    fa.h[3] = mk_head(m_qf2, hQF2, nullptr, u_l0[3], act, A, A);
    fa.h[4] = mk_head(m_tgt, hTGT, nullptr, u_l0[4], nullptr, 0, 0);
    fa.h[5] = mk_head(m_qf1, hQF1PI, nullptr, u_l0[2], pi_a, A, A);
    fa.h[6] = mk_head(m_qf2, hQF2PI, nullptr, u_l0[3], pi_a, A, A);
    fa.B = B; fa.A = A; fa.eps = eps_buf; fa.pi_a = pi_a; fa.logp = logp; fa.ent = ent;
    {
      const char* nm = getenv("GRL_NO_HEADS_MFMA");
      // heads_mfma.h: any layers up to 64 wide with 2A <= 64, and the shipped wide shape -- layers [128, 128], A <= 8,
      // whole 16-row blocks (the general form at that width runs out of registers: those stay on heads_kernels.h)
      heads_mfma = !(nm && nm[0] == '1') && 2 * A <= 64;
      bool narrow = true;
      for (int l = 0; l < L; ++l) narrow = narrow && hid[l] <= 64;
      bool wide_ok = L == 2 && hid[0] == 128 && hid[1] == 128 && A <= 8 && B % HT_RB == 0;
      // the 128-wide kernel keeps 75 KB of static LDS per workgroup: fits gfx950's 160 KB; a device that offers less keeps the VALU chains
      if (!narrow && wide_ok && device_lds_bytes() < (int)heads_fused_lds_bytes(HEADS_FAST_128)) wide_ok = false;
      heads_mfma = heads_mfma && (narrow || wide_ok);
    }
    if (heads_mfma) {
      // forward and backward of every head in one launch (heads_mfma.h); d_out / gradients as in the backward args below
      HeadsFusedArgs ha;
      memset(&ha, 0, sizeof(ha));
      ha.h[0] = mk_head(m_pi, hPI, &gPI, u_l0[0], nullptr, 0, 0);
      ha.h[1] = mk_head(m_vf, hVF, &gVF, u_l0[1], nullptr, 0, 0);
      ha.h[2] = mk_head(m_qf1, hQF1, &gQF1, u_l0[2], act, A, A);
      ha.h[3] = mk_head(m_qf2, hQF2, &gQF2, u_l0[3], act, A, A);
      ha.h[4] = mk_head(m_tgt, hTGT, nullptr, u_l0[4], nullptr, 0, 0);
      ha.h[5] = mk_head(m_qf1, hQF1PI, nullptr, u_l0[2], pi_a, A, A);
      ha.h[6] = mk_head(m_qf2, hQF2PI, nullptr, u_l0[3], pi_a, A, A);
      ha.B = B; ha.A = A; ha.eps = eps_buf; ha.pi_a = pi_a; ha.logp = logp; ha.ent = ent;
      ha.log_ent_coef = params + ent_off; ha.da_pi = da_pi; ha.dmu = dmu; ha.dls = dls; ha.ld_dm = ld_dm;
      ha.rew = rew; ha.done = done; ha.gamma = c.gamma;
      ha.d_out[1] = d_v; ha.d_out[2] = d_qf1; ha.d_out[3] = d_qf2; ha.d_out[4] = d_qf1pi; ha.ld_d = ld_d;
      if (tune_int("heads_stamps", 0)) {     // in-kernel phase stamps (scripts/heads_stamps.py)
        ha.stamps = (unsigned long long*)wk.take(4 * 32 * 8);
        zero_once.push_back({ha.stamps, 4 * 32 * 8});
        dbg["heads_stamps"] = {(const float*)ha.stamps, 4 * 32 * 2};
      }
      // argument blocks: [0] the plain update, [1] / [2] updates of a prefetching multi-update call whose head launch
      // opens the update (Adam step size) and, from the second update on, advances the RNG counter
      HeadsFusedArgs hb = ha, hc = ha;
      hb.sc = hc.sc = sc; hb.tick = hc.tick = 1; hb.rng_advance = 0; hc.rng_advance = 1;
      const HeadsFusedArgs* d_ha = upload_vec(wk, std::vector<HeadsFusedArgs>{ha, hb, hc});
      const int nblk = (B + HT_RB - 1) / HT_RB;
      int wmax = 0;
      for (int l = 0; l < L; ++l) wmax = std::max(wmax, hid[l]);
      const bool wide = wmax > 64;            // two column blocks per wave (heads_mfma.h, W = 128)
      // the reference's shapes: layers [64, 64] (gripper_grasp.yaml) and [128, 128] (SAC_real_2m_buffer_128/config.yaml)
      // (the 128-wide fast form also splits its few-output stages over the waves: 2A <= 16)
      const bool fast = L == 2 && hid[0] == hid[1] && (hid[0] == 64 || (hid[0] == 128 && A <= 8)) && B % HT_RB == 0;
      for (int v = 0; v < 3; ++v) {
        Op op; op.tag = "heads";
        const HeadsFusedArgs* dv = d_ha + v;
        const int shape = wide ? HEADS_FAST_128 : (fast ? HEADS_FAST_64 : HEADS_GENERAL_64);
        op.run = [dv, nblk, shape](hipStream_t s) { launch_heads_fused(shape, nblk, s, dv); };
        if (v == 0) ops_grads.push_back(op);
        else pf_heads[v - 1] = op;
        ride_heads_args = d_ha; ride_heads_shape = shape; ride_heads_nblk = nblk;
      }
    } else {
    Op op; op.tag = "heads_fwd";
    op.run = [fa](hipStream_t s) { launch_heads_fwd(fa, s); };
    ops_grads.push_back(op);
    }
  } else {
    // heads forward: pi, vf, qf1, qf2 (data action), target vf
    for (int l = 0; l < L; ++l) {
      std::vector<IgemmProb> pr;
      pr.push_back(head_layer(m_pi, P, hPI, l, feat[0], ldf, F, nullptr, 0, 0, B));
      pr.push_back(head_layer(m_vf, P, hVF, l, feat[1], ldf, F, nullptr, 0, 0, B));
      pr.push_back(head_layer(m_qf1, P, hQF1, l, feat[1], ldf, F, act, A, A, B));
      pr.push_back(head_layer(m_qf2, P, hQF2, l, feat[1], ldf, F, act, A, A, B));
      pr.push_back(head_layer(m_tgt, P, hTGT, l, feat[2], ldf, F, nullptr, 0, 0, B));
      add_launch(ops_grads, "heads_fwd", 0, pr);
    }
    {
      std::vector<IgemmProb> pr;
      pr.push_back(head_out(m_pi, P, hPI, 0, B));
      pr.push_back(head_out(m_pi, P, hPI, 1, B));
      pr.push_back(head_out(m_vf, P, hVF, 0, B));
      pr.push_back(head_out(m_qf1, P, hQF1, 0, B));
      pr.push_back(head_out(m_qf2, P, hQF2, 0, B));
      pr.push_back(head_out(m_tgt, P, hTGT, 0, B));
      add_launch(ops_grads, "heads_fwd", 0, pr);
    }
    {
      SampleArgs sa{hPI.out[0], hPI.out[1], eps_buf, B, A, pi_a, nullptr, logp, ent};
      Op op; op.tag = "sample";
      op.run = [sa](hipStream_t s) {
        hipLaunchKernelGGL(sample_kernel, dim3((sa.B + 255) / 256), dim3(256), 0, s, sa);
      };
      ops_grads.push_back(op);
    }
    for (int l = 0; l < L; ++l) {
      std::vector<IgemmProb> pr;
      pr.push_back(head_layer(m_qf1, P, hQF1PI, l, feat[1], ldf, F, pi_a, A, A, B));
      pr.push_back(head_layer(m_qf2, P, hQF2PI, l, feat[1], ldf, F, pi_a, A, A, B));
      add_launch(ops_grads, "heads_fwd", 0, pr);
    }
    {
      std::vector<IgemmProb> pr;
      pr.push_back(head_out(m_qf1, P, hQF1PI, 0, B));
      pr.push_back(head_out(m_qf2, P, hQF2PI, 0, B));
      add_launch(ops_grads, "heads_fwd", 0, pr);
    }
  }
  {
    LossArgs la;
    la.B = B; la.gamma = c.gamma; la.target_entropy = c.target_entropy; la.lr = c.lr;
    la.rew = rew; la.done = done; la.v_tgt = hTGT.out[0]; la.qf1 = hQF1.out[0]; la.qf2 = hQF2.out[0];
    la.v = hVF.out[0]; la.qf1_pi = hQF1PI.out[0]; la.qf2_pi = hQF2PI.out[0]; la.logp = logp; la.entropy = ent;
    la.log_ent_coef = params + ent_off;
    la.d_qf1 = d_qf1; la.d_qf2 = d_qf2; la.d_v = d_v; la.d_qf1_pi = d_qf1pi; la.ld_d = ld_d;
    la.g_log_ent_coef = grads + ent_off; la.sc = sc;
    la.write_d = fused_heads ? 0 : 1;
    la.adam_ticked = fused_heads ? 1 : 0;
    la.ent_param = params + ent_off; la.ent_m = adam_m + ent_off; la.ent_v = adam_v + ent_off;
    loss_args = la;
    if (!fused_heads) {   // fused heads: output gradients are formed in heads_bwd_kernel, reductions ride on reduce_slabs
      Op op; op.tag = "sac_loss";
      op.run = [la](hipStream_t s) { hipLaunchKernelGGL(sac_loss_kernel, dim3(1), dim3(256), 0, s, la); };
      ops_grads.push_back(op);
    }
  }

  // =============================================================== backward through the heads
  // g[l] = gradient w.r.t. the pre-activation of layer l (ReLU mask already applied)
  if (fused_heads) {
    HeadsBwdArgs ba;
    memset(&ba, 0, sizeof(ba));
    ba.h[0] = mk_head(m_pi, hPI, &gPI, u_l0[0], nullptr, 0, 0);
    ba.h[1] = mk_head(m_vf, hVF, &gVF, u_l0[1], nullptr, 0, 0);
    ba.h[2] = mk_head(m_qf1, hQF1, &gQF1, u_l0[2], act, A, A);
    ba.h[3] = mk_head(m_qf2, hQF2, &gQF2, u_l0[3], act, A, A);
    ba.h[4] = mk_head(m_qf1, hQF1PI, &gQF1PI, u_l0[2], pi_a, A, A);
    ba.h[1].dout[0] = d_v; ba.h[2].dout[0] = d_qf1; ba.h[3].dout[0] = d_qf2; ba.h[4].dout[0] = d_qf1pi;
    for (int k = 1; k < 5; ++k) ba.h[k].ld_dout = ld_d;
    ba.ld_dm = ld_dm;
    ba.rew = rew; ba.done = done; ba.v_tgt = hTGT.out[0]; ba.qf1 = hQF1.out[0]; ba.qf2 = hQF2.out[0]; ba.v = hVF.out[0];
    ba.qf1_pi = hQF1PI.out[0]; ba.qf2_pi = hQF2PI.out[0]; ba.logp = logp; ba.gamma = c.gamma;
    ba.d_out[1] = d_v; ba.d_out[2] = d_qf1; ba.d_out[3] = d_qf2; ba.d_out[4] = d_qf1pi; ba.ld_d = ld_d;
    ba.B = B; ba.A = A; ba.mu = hPI.out[0]; ba.ls_raw = hPI.out[1]; ba.eps = eps_buf; ba.pi_a = pi_a;
    ba.log_ent_coef = params + ent_off; ba.da_pi = da_pi; ba.dmu = dmu; ba.dls = dls;
    if (!heads_mfma) {
    Op op; op.tag = "heads_bwd";
    op.run = [ba](hipStream_t s) { launch_heads_bwd(ba, s); };
    ops_grads.push_back(op);
    }
    if (cnn) {
      // d feat = g0 . W0[0:Fc]^T, masked by feat > 0.  Critic net: the three layer-0 gradients sit side
      // by side in g0cat and the three kernels are reached through a table (K = 3*H0 in one pass).
      const int H0 = hid[0];
      std::vector<int32_t> qt3(3 * H0), qt1(H0);
      const int64_t offs[3] = {m_vf.w[0], m_qf1.w[0], m_qf2.w[0]};
      for (int k = 0; k < 3; ++k)
        for (int n = 0; n < H0; ++n) qt3[k * H0 + n] = (int32_t)(offs[k] - offs[0]) + n;
      for (int n = 0; n < H0; ++n) qt1[n] = n;
      auto tab_bwd = [&](const float* g, int ldg, int K, const float* wbase, const int32_t* dtab, float* dx,
                         const float* mask) {
        IgemmProb p = blank();
        p.M = B; p.N = Fc; p.K = K;
        p.p_base[0] = g; p.p_ld_i[0] = ldg; p.p_ld_r[0] = 1; single_part(p);
        p.q_base[0] = wbase; p.q_tab_r = dtab; p.q_ld_j[0] = H0;
        p.c = dx; p.ldc = ldf; p.relu_mask = mask;
        p.vflags = VF_Q_TAB;
        set_split(p, 1);
        return p;
      };
      std::vector<IgemmProb> pr;
      pr.push_back(tab_bwd(g0cat, 3 * H0, 3 * H0, P + m_vf.w[0], upload_vec(wk, qt3), dfeat[1], feat[1]));
      pr.push_back(tab_bwd(gPI.g[0], H0, H0, P + m_pi.w[0], upload_vec(wk, qt1), dfeat[0], feat[0]));
      add_launch(ops_grads, "heads_dfeat", 1, pr);
    }
  } else {
    {
      std::vector<IgemmProb> pr;   // output layer -> g[L-1]
      pr.push_back(dense_bwd({{d_v, 1, 1, P + m_vf.ow[0]}}, B, 0, hid[L - 1], gVF.g[L - 1], hid[L - 1], hVF.z[L - 1]));
      pr.push_back(dense_bwd({{d_qf1, 1, 1, P + m_qf1.ow[0]}}, B, 0, hid[L - 1], gQF1.g[L - 1], hid[L - 1], hQF1.z[L - 1]));
      pr.push_back(dense_bwd({{d_qf2, 1, 1, P + m_qf2.ow[0]}}, B, 0, hid[L - 1], gQF2.g[L - 1], hid[L - 1], hQF2.z[L - 1]));
      pr.push_back(dense_bwd({{d_qf1pi, 1, 1, P + m_qf1.ow[0]}}, B, 0, hid[L - 1], gQF1PI.g[L - 1], hid[L - 1], hQF1PI.z[L - 1]));
      add_launch(ops_grads, "heads_bwd", 1, pr);
    }
    for (int l = L - 1; l >= 1; --l) {
      std::vector<IgemmProb> pr;
      pr.push_back(dense_bwd({{gVF.g[l], hid[l], hid[l], P + m_vf.w[l]}}, B, 0, hid[l - 1], gVF.g[l - 1], hid[l - 1], hVF.z[l - 1]));
      pr.push_back(dense_bwd({{gQF1.g[l], hid[l], hid[l], P + m_qf1.w[l]}}, B, 0, hid[l - 1], gQF1.g[l - 1], hid[l - 1], hQF1.z[l - 1]));
      pr.push_back(dense_bwd({{gQF2.g[l], hid[l], hid[l], P + m_qf2.w[l]}}, B, 0, hid[l - 1], gQF2.g[l - 1], hid[l - 1], hQF2.z[l - 1]));
      pr.push_back(dense_bwd({{gQF1PI.g[l], hid[l], hid[l], P + m_qf1.w[l]}}, B, 0, hid[l - 1], gQF1PI.g[l - 1], hid[l - 1], hQF1PI.z[l - 1]));
      add_launch(ops_grads, "heads_bwd", 1, pr);
    }
    {
      std::vector<IgemmProb> pr;   // first layer: d a_pi (policy path) and d feat (critic CNN path)
      pr.push_back(dense_bwd({{gQF1PI.g[0], hid[0], hid[0], P + m_qf1.w[0]}}, B, F, A, da_pi, A, nullptr));
      if (cnn)
        pr.push_back(dense_bwd({{gVF.g[0], hid[0], hid[0], P + m_vf.w[0]},
                                {gQF1.g[0], hid[0], hid[0], P + m_qf1.w[0]},
                                {gQF2.g[0], hid[0], hid[0], P + m_qf2.w[0]}},
                               B, 0, Fc, dfeat[1], ldf, feat[1]));
      add_launch(ops_grads, "heads_bwd", 1, pr);
    }
    {
      SampleBwdArgs sb{hPI.out[0], hPI.out[1], eps_buf, pi_a, da_pi, A, params + ent_off, B, A, dmu, dls};
      Op op; op.tag = "sample_bwd";
      op.run = [sb](hipStream_t s) {
        hipLaunchKernelGGL(sample_bwd_kernel, dim3((sb.B + 255) / 256), dim3(256), 0, s, sb);
      };
      ops_grads.push_back(op);
    }
    {
      std::vector<IgemmProb> pr;
      pr.push_back(dense_bwd({{dmu, A, A, P + m_pi.ow[0]}, {dls, A, A, P + m_pi.ow[1]}}, B, 0, hid[L - 1],
                             gPI.g[L - 1], hid[L - 1], hPI.z[L - 1]));
      add_launch(ops_grads, "heads_bwd", 1, pr);
    }
    for (int l = L - 1; l >= 1; --l) {
      std::vector<IgemmProb> pr;
      pr.push_back(dense_bwd({{gPI.g[l], hid[l], hid[l], P + m_pi.w[l]}}, B, 0, hid[l - 1], gPI.g[l - 1], hid[l - 1], hPI.z[l - 1]));
      add_launch(ops_grads, "heads_bwd", 1, pr);
    }
    if (cnn) {
      std::vector<IgemmProb> pr;
      pr.push_back(dense_bwd({{gPI.g[0], hid[0], hid[0], P + m_pi.w[0]}}, B, 0, Fc, dfeat[0], ldf, feat[0]));
      add_launch(ops_grads, "heads_bwd", 1, pr);
    }

  }
  // =============================================================== backward through the two CNNs
  bool only_vec_dense = false;
  int rider_budget = 0;                            // empty slots of conv3_bwd's last dispatch round (see below)
  std::vector<Op> conv3_bwd_plain;                 // conv3_bwd without riders, for the staged data-parallel plan
  std::vector<IgemmProb> dense_affine, conv_all;   // dense / conv weight-gradient problems as first built (staged plan)
  std::vector<IgemmProb> wg, wgc[3];   // weight gradients: dense layers / conv layers 1..3
  std::vector<IgemmProb> bwd_pr[3];    // backward-data stages fc, conv3, conv2 (launched below, once their fillers are known)
  // Measured and rejected on top of the merged weight-gradient launch (MI355X, B = 256, updates/s; conv2_bwd needs 46 KB of
  // LDS, so a CU holds three of its 1024 tiles and the launch runs as a wave of 768 plus a third-full wave of 256):
  //  * every weight gradient except conv1's riding on conv2_bwd's launch, longest tiles first, conv1 (the only consumer
  //    of conv2_bwd's result) as a launch of its own: the pair launch takes 57.7 us against 31.8 + 36.5, but conv1 alone
  //    costs 13.2 us (one short tile per CU) and its 225 slabs another 8 us of reduction: 4 395 against 4 600;
  //  * only the dense layers' weight gradients (303 tiles of 8 slabs, the length of a conv2_bwd tile) riding behind
  //    conv2_bwd's tiles: they start when the first wave drains (18 us) and end at 36 us instead of 28: conv2_bwd
  //    31.8 -> 41.2 us, the weight-gradient launch 36.6 -> 28.2 us, reduction +2.5 us: 4 519 against 4 580.
  if (cnn) {
    // backward-data with exact taps (conv_bwd_tabs_exact; the masked parity-class form of conv_bwd_tabs gives bit-identical
    // sums with 1.8-2.25x the MACs -- the auto-encoder's padded convolutions still use it)
    std::vector<int32_t> untouched;
    std::vector<ConvBwdClass> bc3 = conv_bwd_tabs_exact(cg[2], B, nullptr);
    std::vector<ConvBwdClass> bc2 = conv_bwd_tabs_exact(cg[1], B, &untouched);
    if (!untouched.empty())   // input pixels of conv2 that no output window covers (row / column 14): their gradient is zero
      for (int n = 0; n < 2; ++n) zero_once.push_back({g1[n], (size_t)(((int64_t)B * 225 - 1) * ld1 + 32) * 4});
    for (int n = 0; n < 2; ++n)
      bwd_pr[0].push_back(dense_bwd({{dfeat[n], ldf, 512, P + ex[n].fw}}, B, 0, 1024, g3[n], 1024, a3[n]));
    for (int n = 0; n < 2; ++n)
      for (auto& cl : bc3) bwd_pr[1].push_back(conv_bwd(g3[n], cl, cg[2], P + ex[n].w[2], g2[n], a2[n]));
    for (int n = 0; n < 2; ++n)
      for (auto& cl : bc2) bwd_pr[2].push_back(conv_bwd(g2[n], cl, cg[1], P + ex[n].w[1], g1[n], a1[n]));
    // conv / fc weight gradients (split reductions land in slabs, summed by reduce_slabs)
    // conv3_bwd (576 tiles of the 32x64 shape at B = 256, three per CU) fills two dispatch rounds and a quarter of the third:
    // 192 CUs hold two tiles, 64 hold three, and the launch lasts as long as those 64.  Dense-layer weight gradients
    // whose operands are complete by then (d feat, head gradients) and whose tiles are as long as conv3_bwd's (reduction
    // = the batch, 8 slabs) ride in the empty slots of that round -- list positions 576.. land exactly on the CUs that
    // hold two -- and leave the merged weight-gradient launch.
    // backward-data of conv3 and conv2 as ONE sample-local launch (conv_stack.h: conv_stack_bwd_kernel, scatter form with the
    // forward's MACs) -- OPT-IN, GRL_TUNE conv_stack_bwd=1.  Measured on the MI355X at B = 256 (same box, scripts/ab_env.sh):
    // first cut 5 660 against 5 689 updates/s; with the masks prefetched, wave-local epilogues, batched LDS read-modify-writes
    // and falling issue priority 5 741 - 5 762 against 5 755 - 5 757: NEUTRAL.  The launch itself (33 - 35 us) replaces 17.0 +
    // 20.3 us, but the dense weight gradients that ride in conv3_bwd's empty slots go back into the weight-gradient launch
    // (30.5 -> 35 us), and the conv3 half is bound by its kernel stream: 16 output pixels per sample reuse every 4 KB of
    // kernel for 16 MFMAs, where the batched exact-tap launch reuses it over 128 rows (profiles/r05_ab_conv_stack_bwd.txt)
    conv_stack_bwd = conv_stack && tune_int("conv_stack_bwd", 0) != 0;
    if (!conv_stack_bwd) {
      int cfg3 = -1;
      const int t3 = planned_tiles(bwd_pr[1], 1, "conv3_bwd", &cfg3);
      rider_budget = cfg3 == 3 ? free_slots(t3, 3) : 0;
    }
    int wsplit[3] = {72, 12, 6};   // reduction splits of conv1..3 (GRL_TUNE wg_split=a/b/c overrides the model's choice)
    pick_wgrad_splits(cg, ft, 2, wsplit, rider_budget);
    tune_int3("wg_split", wsplit);
    {   // conv1 of both networks: one problem over the side-by-side gradient buffer, columns 32n.. -> net n
      IgemmProb p = conv_wgrad(x_obs, ft[0], cg[0], g1[0], nullptr, wsplit[0], 2);
      p.c = wk.f32(p.slab_stride * p.split);
      wgc[0].push_back(p);
      for (int n = 0; n < 2; ++n) {
        ReduceDesc r;
        memset(&r, 0, sizeof(r));
        r.src = p.c + 32 * n; r.splits = p.split; r.slab_stride = p.slab_stride; r.row_len = 32; r.src_ld = 64;
        r.dst = grads + ex[n].w[0]; r.n = cg[0].K() * 32;
        reduces.push_back(r);
        ReduceDesc rb = r;
        rb.src = p.c + (int64_t)p.p_ones_i * 64 + 32 * n; rb.dst = grads + ex[n].b[0]; rb.n = 32;
        reduces.push_back(rb);
      }
    }
    for (int n = 0; n < 2; ++n) {
      {
        IgemmProb p = conv_wgrad(a1[n], ft[1], cg[1], g2[n], nullptr, wsplit[1]);
        p.c = wk.f32(p.slab_stride * p.split);
        add_wgrad(wgc[1], p, ex[n].w[1], 0, cg[1].K(), ex[n].b[1]);
      }
      {
        IgemmProb p = conv_wgrad(a2[n], ft[2], cg[2], g3[n], nullptr, wsplit[2]);
        p.c = wk.f32(p.slab_stride * p.split);
        add_wgrad(wgc[2], p, ex[n].w[2], 0, cg[2].K(), ex[n].b[2]);
      }
      {
        IgemmProb p = dense_wgrad(a3[n], 1024, 1024, true, dfeat[n], ldf, 512, B, nullptr, 1);
        p.c = wk.f32(p.slab_stride * p.split);
        add_wgrad(wg, p, ex[n].fw, 0, 1024, ex[n].fb);
      }
    }
  }
  // head weight gradients.  Fused-heads layout: every operand is row-padded to a multiple of 4 floats, and
  // problems without a bias still carry the ones row (its slab row is simply not reduced), so that all
  // dense weight gradients form ONE uniform launch of the vectorised kernel.
  auto head_wgrads = [&](const MlpP& m, const HeadAct& h, const HeadGrad& g, const float* x0, int ld0, int K0,
                         const float* x1, int ld1, int K1, std::vector<const float*> douts, int ld_dout) {
    const int K0p = ld0 >= (int)rup(K0, 4) ? (int)rup(K0, 4) : K0;   // feat rows are padded to ldf (zeros)
    const int K1p = (K1 > 0 && ld1 >= (int)rup(K1, 4)) ? (int)rup(K1, 4) : K1;
    for (int l = 0; l < L; ++l) {
      if (l == 0) {
        if (K1 > 0) {
          IgemmProb p0 = dense_wgrad(x0, ld0, K0, fused_heads, g.g[0], g.ld0, hid[0], B, nullptr, 1, K0p);
          p0.c = wk.f32(p0.slab_stride * p0.split);
          add_wgrad(wg, p0, m.w[0], 0, K0, -1);
          IgemmProb p1 = dense_wgrad(x1, ld1, K1, true, g.g[0], g.ld0, hid[0], B, nullptr, 1, K1p);
          p1.c = wk.f32(p1.slab_stride * p1.split);
          add_wgrad(wg, p1, m.w[0], K0, K1, m.b[0]);
        } else {
          IgemmProb p0 = dense_wgrad(x0, ld0, K0, true, g.g[0], g.ld0, hid[0], B, nullptr, 1, K0p);
          p0.c = wk.f32(p0.slab_stride * p0.split);
          add_wgrad(wg, p0, m.w[0], 0, K0, m.b[0]);
        }
      } else {
        IgemmProb p = dense_wgrad(h.z[l - 1], hid[l - 1], hid[l - 1], true, g.g[l], hid[l], hid[l], B, nullptr, 1);
        p.c = wk.f32(p.slab_stride * p.split);
        add_wgrad(wg, p, m.w[l], 0, hid[l - 1], m.b[l]);
      }
    }
    for (int k = 0; k < m.n_out; ++k) {
      IgemmProb p = dense_wgrad(h.z[L - 1], hid[L - 1], hid[L - 1], true, douts[k], ld_dout, m.out_dim, B, nullptr, 1);
      p.c = wk.f32(p.slab_stride * p.split);
      add_wgrad(wg, p, m.ow[k], 0, hid[L - 1], m.ob[k]);
    }
  };
  const float* act_w = fused_heads ? act_p : act;
  const int ld_act_w = fused_heads ? Ap : A;
  head_wgrads(m_pi, hPI, gPI, feat[0], ldf, F, nullptr, 0, 0, {dmu, dls}, ld_dm);
  head_wgrads(m_vf, hVF, gVF, feat[1], ldf, F, nullptr, 0, 0, {d_v}, ld_d);
  head_wgrads(m_qf1, hQF1, gQF1, feat[1], ldf, F, act_w, ld_act_w, A, {d_qf1}, ld_d);
  head_wgrads(m_qf2, hQF2, gQF2, feat[1], ldf, F, act_w, ld_act_w, A, {d_qf2}, ld_d);
  {
    // the vectorised kernel needs uniform launches: with / without bias row; whatever it cannot take
    // goes to igemm_kernel.  (Fused heads: everything lands in the first group.)
    std::vector<IgemmProb> wg_ones, wg_plain, wg_rest;
    for (auto& p : wg) {
      if (!v2_prob_ok(p, 2) || (p.K % 4)) wg_rest.push_back(p);
      else if (p.p_ones_i >= 0) wg_ones.push_back(p);
      else wg_plain.push_back(p);
    }
    wgrad_ops.clear();
    // One launch for all weight gradients: the dense problems (x^T read with affine addresses) are re-expressed
    // with the table addressing of the convolution ones and appended to that launch -- one kernel boundary
    // less, and their short reductions (K = B) fill the tail of the long convolution tiles.
    dense_affine = wg_ones;                                   // (kept for the staged data-parallel plan below)
    for (int l = 2; l >= 0; --l) conv_all.insert(conv_all.end(), wgc[l].begin(), wgc[l].end());
    only_vec_dense = wg_plain.empty() && wg_rest.empty();
    std::vector<IgemmProb> wg_merged;
    if (cnn && wg_plain.empty() && !wgc[0].empty() && v2_prob_ok(wgc[0][0], 2)) {
      for (auto p : wg_ones) {
        std::vector<int32_t> ti(p.M), tr(p.K);
        for (int i = 0; i < p.M; ++i) ti[i] = i * p.p_ld_i[0];
        for (int r = 0; r < p.K; ++r) tr[r] = r * p.p_ld_r[0];
        p.p_tab_i = upload_vec(wk, ti);
        p.p_tab_r = upload_vec(wk, tr);
        p.vflags |= VF_P_TABS;                    // v2_prob_ok held for the affine form: unit stride along i, ld_r % 4 == 0
        wg_merged.push_back(p);
      }
      wg_ones.clear();
    }
    if (cnn && conv_stack_bwd) {
      add_launch(ops_grads, "fc_bwd", 1, bwd_pr[0]);
      ConvStackBwdArgs ba;
      memset(&ba, 0, sizeof(ba));
      for (int n = 0; n < 2; ++n) {
        ConvStackBwdNet& bn = ba.nets[n];
        bn.g3 = g3[n]; bn.a2 = a2[n]; bn.a1 = a1[n]; bn.w2 = P + ex[n].w[1]; bn.w3 = P + ex[n].w[2];
        bn.g2 = g2[n]; bn.g1 = g1[n]; bn.ld1 = ld1;
      }
      ba.B = B; ba.n_nets = 2;
      Op op; op.tag = "conv_stack_bwd";
      op.flops = op.flops_exec = 2.0 * B * 2 * (36.0 * 64 * 512 + 16.0 * 64 * 576);
      op.run = [ba](hipStream_t s) { launch_conv_stack_bwd(ba, s); };
      ops_grads.push_back(op);
    } else if (cnn) {
      add_launch(ops_grads, "fc_bwd", 1, bwd_pr[0]);
      std::vector<IgemmProb> riders = take_riders(wg_merged, rider_budget);
      if (!riders.empty()) add_launch(conv3_bwd_plain, "conv3_bwd", 1, bwd_pr[1]);
      add_launch(ops_grads, "conv3_bwd", 1, bwd_pr[1], "wgrad_dense", 2, riders);
      add_launch(ops_grads, "conv2_bwd", 1, bwd_pr[2]);
    }
    add_launch(wgrad_ops, "wgrad_dense", 2, wg_ones);
    add_launch(wgrad_ops, "wgrad_dense", 2, wg_plain);
    add_launch(wgrad_ops, "wgrad_small", 2, wg_rest);
    {
      std::vector<IgemmProb> all;
      for (int l = 2; l >= 0; --l) all.insert(all.end(), wgc[l].begin(), wgc[l].end());
      all.insert(all.end(), wg_merged.begin(), wg_merged.end());
      add_launch(wgrad_ops, "wgrad_conv", 2, all);
      if (x_obs_b && cnn) {      // the same launch reading conv1's input from the second image buffer (flavour 1 of "gather_ride")
        bool patched = false;
        for (auto& p : all)
          if (p.p_base[0] == x_obs) { p.p_base[0] = x_obs_b; patched = true; }
        std::vector<Op> tmp;
        if (patched) add_launch(tmp, "wgrad_conv", 2, all);
        if (tmp.size() == 1) { wgrad_conv_alt = tmp[0]; have_wgrad_conv_alt = true; }
      }
    }
  }
  // ---- schedule: every weight-gradient launch directly behind the last producer of its operands
  {
    std::vector<Op> sched;
    auto take = [&](const char* tag) {
      for (auto& o : wgrad_ops)
        if (o.tag == tag) sched.push_back(o);
    };
    int dense_after = -1;   // the dense weight gradients need every head gradient and (CNN) d feat
    for (size_t k = 0; k < ops_grads.size(); ++k)
      if (ops_grads[k].tag == "heads_dfeat" || ops_grads[k].tag == "heads_bwd" || ops_grads[k].tag == "heads") dense_after = (int)k;
    for (size_t k = 0; k < ops_grads.size(); ++k) {
      const Op& o = ops_grads[k];
      sched.push_back(o);
      if ((int)k == dense_after) { take("wgrad_dense"); take("wgrad_small"); }
      if (o.tag == "conv2_bwd" || o.tag == "conv_stack_bwd") take("wgrad_conv");
    }
    ops_grads.swap(sched);
  }
  std::vector<Op> st0_ops, st1_ops;     // staged plan without its two reductions (added below)
  {
    // ---- staged gradient computation (data parallel): the dense weight gradients -- 90 % of the bucket's bytes --
    // get a launch of their own right after the feature gradients, so that their all-reduce can travel while the
    // convolution backward and the convolution weight gradients run (grasp_rl/parallel.py).  Costs two launches
    // more than the single-exchange plan; same tiles, same arithmetic.
    staged_ok = cnn && fused_heads && only_vec_dense && !dense_affine.empty() && !conv_all.empty();
    if (staged_ok) {
      int cut = -1;
      for (size_t k = 0; k < ops_grads.size(); ++k)
        if (ops_grads[k].tag == "heads_dfeat") cut = (int)k;
      staged_ok = cut >= 0;
      if (staged_ok) {
        for (int k = 0; k <= cut; ++k) st0_ops.push_back(ops_grads[k]);
        add_launch(st0_ops, "wgrad_dense", 2, dense_affine, "", 0, {}, 0);
        for (size_t k = cut + 1; k < ops_grads.size(); ++k) {
          if (ops_grads[k].tag.compare(0, 5, "wgrad") == 0) continue;
          if (ops_grads[k].tag == "conv3_bwd" && !conv3_bwd_plain.empty()) st1_ops.push_back(conv3_bwd_plain[0]);   // (riders belong to stage 0 here)
          else st1_ops.push_back(ops_grads[k]);
        }
        add_launch(st1_ops, "wgrad_conv", 2, conv_all, "", 0, {}, 0);
      }
    }
  }
  {
    std::vector<int2> rt = reduce_tiles();          // (marks the 16-byte-eligible descriptors: before the upload)
    d_reduces = upload_vec(wk, reduces);
    int2* d_rt = upload_vec(wk, rt);
    const int ntiles = (int)rt.size();
    ReduceDesc* dr = d_reduces;
    Op op; op.tag = "reduce_slabs";
    op.join = true;
    const LossArgs la = loss_args;
    const int has_loss = fused_heads ? 1 : 0;
    AdamArgs aa;
    memset(&aa, 0, sizeof(aa));
    aa.params = params; aa.grads = grads; aa.m = adam_m; aa.v = adam_v; aa.n_train = n_train; aa.sc = sc;
    aa.grad_scale = 1.f; aa.tau = c.tau; aa.eps = 1e-8f;
    aa.src_ofs = vf_off; aa.n_polyak = n_polyak; aa.target = params + tgt_off;
    op.run = [dr, d_rt, ntiles, la, has_loss, aa](hipStream_t s) {
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ntiles + has_loss), dim3(256), 0, s, dr, d_rt, ntiles, la, has_loss, aa, 0);
    };
    ops_grads.push_back(op);
    adam_base = aa;
    red_all.tiles = d_rt; red_all.n = ntiles; red_all.has_loss = has_loss;
    if (staged_ok) {
      // reductions of the two stages: convolution descriptors are those that land in the conv variables of a net
      auto is_conv = [&](const ReduceDesc& r) {
        for (int n = 0; n < 2; ++n)
          if (r.dst >= grads + ex[n].w[0] && r.dst < grads + ex[n].fw) return true;
        return false;
      };
      std::vector<int2> rt0 = reduce_tiles([&](const ReduceDesc& r) { return !is_conv(r); });
      std::vector<int2> rt1 = reduce_tiles(is_conv);
      int2* d_rt0 = upload_vec(wk, rt0);
      int2* d_rt1 = upload_vec(wk, rt1);
      const int n0 = (int)rt0.size(), n1 = (int)rt1.size();
      Op r0; r0.tag = "reduce_dense";
      r0.run = [dr, d_rt0, n0, la, aa](hipStream_t s) {
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(n0), dim3(256), 0, s, dr, d_rt0, n0, la, 0, aa, 0);
      };
      Op r1; r1.tag = "reduce_conv";
      r1.run = [dr, d_rt1, n1, la, has_loss, aa](hipStream_t s) {
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(n1 + has_loss), dim3(256), 0, s, dr, d_rt1, n1, la, has_loss, aa, 0);
      };
      red_dense.tiles = d_rt0; red_dense.n = n0; red_dense.has_loss = 0;
      red_conv.tiles = d_rt1; red_conv.n = n1; red_conv.has_loss = has_loss;
      ops_stage0 = st0_ops; ops_stage0.push_back(r0);
      ops_stage1 = st1_ops; ops_stage1.push_back(r1);
    }
    // Full updates (no gradient exchange in between): every trainable element is the sum of one slab
    // column, so Adam + Polyak are applied where the sum is formed -- one launch and one pass over the
    // gradient bucket less.  log_ent_coef, whose gradient comes from the loss workgroup, is applied there.
    if (has_loss && tune_int("fused_adam", 1)) {
      ops_grads_apply.assign(ops_grads.begin(), ops_grads.end() - 1);
      Op fo; fo.tag = "reduce_adam";
      fo.join = true;
      fo.bytes = (double)n_train * 4 * 7 + (double)n_polyak * 4 * 2;
      fo.run = [dr, d_rt, ntiles, la, has_loss, aa](hipStream_t s) {
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ntiles + has_loss), dim3(256), 0, s, dr, d_rt, ntiles, la, has_loss, aa, 1);
      };
      ops_grads_apply.push_back(fo);
      // ---- "prefetch": a call of n >= 2 updates on the device RNG gathers the minibatch of update t+1 inside the LAST
      // launch of update t (reduce_slabs_gather_kernel): nothing enters the replay between the updates of one call, the
      // Philox counter makes the draw independent of when it happens, and every reader of update t's minibatch tensors
      // has finished when that launch starts.  Same kernels, same arithmetic, one launch (and one dependent latency
      // chain) less per update.  The head launch opens the update instead of the gather (see HeadsFusedArgs).
      //   first : gather (does not touch the Adam step size) | body, heads[tick] | reduce + Adam + gather(t+1, counter + 1)
      //   middle:                                              body, heads[tick, counter += 1] | reduce + Adam + gather(t+1)
      //   last  :                                              body, heads[tick, counter += 1] | reduce + Adam (counter += 1)
      if (heads_mfma && tune_int("gather_prefetch", 1)) {
        GatherArgs g1 = pf_ga;
        g1.use_rng = 1; g1.adam_tick = 0; g1.quiet = 0; g1.rng_ahead = 0;
        const int gx = pf_gx;
        {
          Op op; op.tag = "gather_norm";
          op.bytes = ops_rng[0].bytes;
          op.run = [g1, gx](hipStream_t s) {
            if (g1.rows > 1) hipLaunchKernelGGL(gather_norm_lin_kernel, dim3(gather_blocks(g1, gx)), dim3(256), 0, s, g1, gx);
            else hipLaunchKernelGGL(gather_norm_kernel, dim3(gx, g1.B, 2), dim3(256), 0, s, g1);
          };
          ops_pf_first.push_back(op);
        }
        GatherArgs g2 = g1;
        g2.quiet = 1; g2.rng_ahead = 1;
        LossArgs lk = la;
        lk.keep_rng = 1;
        Op ro; ro.tag = "reduce_adam";
        ro.join = true;
        ro.bytes = fo.bytes + ops_rng[0].bytes;
        AdamArgs aq = aa;
        aq.skip_bucket = 1;                 // (the call's last update -- `fo` -- leaves its gradients in the bucket)
        ro.bytes -= (double)n_train * 4;
        ro.run = [dr, d_rt, ntiles, lk, has_loss, aq, g2, gx](hipStream_t s) {
          const dim3 grid(ntiles + has_loss + gather_blocks(g2, gx));
          if (g2.rows > 1) hipLaunchKernelGGL(reduce_slabs_gather_kernel<true>, grid, dim3(256), 0, s, dr, d_rt, ntiles, lk, has_loss, aq, 1, g2, gx);
          else hipLaunchKernelGGL(reduce_slabs_gather_kernel<false>, grid, dim3(256), 0, s, dr, d_rt, ntiles, lk, has_loss, aq, 1, g2, gx);
        };
        for (int v = 0; v < 3; ++v) {     // 0 first, 1 middle, 2 last
          std::vector<Op>& dst = v == 0 ? ops_pf_first : (v == 1 ? ops_pf_mid : ops_pf_last);
          for (size_t k = 0; k + 1 < ops_grads_apply.size(); ++k) {
            if (ops_grads_apply[k].tag == "heads") dst.push_back(pf_heads[v == 0 ? 0 : 1]);
            else dst.push_back(ops_grads_apply[k]);
          }
          dst.push_back(v == 2 ? fo : ro);
        }
        pf_lk = lk; pf_g2 = g2;     // (the data-parallel update builds its own final launches from these at connect)
        prefetch_ok = true;
        // ---- "gather_ride": the IMAGES of update t+1 gathered by extra workgroups of update t's HEAD launch -- 64 workgroups of
        // row-local chains on 256 CUs for 16 us -- instead of the reduction launch, which is memory bound itself.  What the
        // gather writes and update t still reads after its head launch: x_obs (conv1's weight gradient, in the last GEMM launch)
        // -- double buffered, flavour f reads buffer f and gathers into the other; x_next (read by the forward stack only) needs
        // no second copy; the per-row extras (direct features, action, reward, done, index, noise: read by the head launch and
        // the dense weight gradients) stay with the reduction launch that ends the update, as a gather of parts = 2.  Counters:
        // the riders draw with DevScalars.rng_img (the head launch itself advances rng_step), which the call's first gather
        // sets and every riding reduction advances.
        //   first : gather[rng_img = c + 1] | body(A), heads[tick; images(t+1) -> B] | reduce + Adam + extras(t+1) [rng_img += 1]
        //   middle: body(f), heads[tick, counter += 1; images(t+1) -> other] | reduce + Adam + extras(t+1) [rng_img += 1]
        //   last  : body(f), heads[tick, counter += 1]                        | reduce + Adam (counter += 1)
        if (x_obs_b && conv_stack && have_wgrad_conv_alt && !conv_stack_bwd && (g2.vec4 || !elem_vec4_built()) && B % GATHER_RIDE_ROWS == 0 &&
            ride_heads_args) {      // (the emulation build has no 16-byte form: its riders walk the rows element by element)
          Op conv_alt;                                 // the forward stack reading the observations from x_obs_b
          bool have_conv_alt = false;
          for (auto& o : ops_grads_apply)
            if (o.tag == "conv_stack_fwd") {
              ConvStackArgs cb = ride_conv_args;
              cb.nets[0].x = x_obs_b; cb.nets[1].x = x_obs_b;
              const int Ci = C_img;
              conv_alt = o;
              conv_alt.run = [cb, Ci](hipStream_t s) { launch_conv_stack_fwd(Ci, cb, s); };
              have_conv_alt = true;
            }
          int ride_rows = tune_int("ride_rows", GATHER_RIDE_ROWS);
          if (ride_rows != 4 && ride_rows != 8 && ride_rows != 16) ride_rows = GATHER_RIDE_ROWS;
          if (have_conv_alt) {
            GatherArgs gx2 = g2;                         // extras of update t+1, carried by update t's reduction launch
            gx2.parts = 2;
            LossArgs lr3 = lk;
            lr3.bump_img = 1;
            Op rx; rx.tag = "reduce_adam";
            rx.join = true;
            rx.bytes = ro.bytes - ops_rng[0].bytes;
            rx.run = [dr, d_rt, ntiles, lr3, has_loss, aq, gx2](hipStream_t s) {
              const dim3 grid(ntiles + has_loss + gather_blocks(gx2, 1));
              if (gx2.rows > 1) hipLaunchKernelGGL(reduce_slabs_gather_kernel<true>, grid, dim3(256), 0, s, dr, d_rt, ntiles, lr3, has_loss, aq, 1, gx2, 1);
              else hipLaunchKernelGGL(reduce_slabs_gather_kernel<false>, grid, dim3(256), 0, s, dr, d_rt, ntiles, lr3, has_loss, aq, 1, gx2, 1);
            };
            ride_lk = lr3; ride_g2 = gx2;
            Op first_g = ops_pf_first[0];                // the call's own gather, leaving rng_img behind
            {
              GatherArgs g0 = g1;
              g0.set_img = 1;
              first_g.run = [g0, gx](hipStream_t s) {
                if (g0.rows > 1) hipLaunchKernelGGL(gather_norm_lin_kernel, dim3(gather_blocks(g0, gx)), dim3(256), 0, s, g0, gx);
                else hipLaunchKernelGGL(gather_norm_kernel, dim3(gx, g0.B, 2), dim3(256), 0, s, g0);
              };
            }
            const HeadsFusedArgs* d_ha = ride_heads_args;
            const int hshape = ride_heads_shape, hnblk = ride_heads_nblk;
            for (int f = 0; f < 2; ++f) {
              GatherArgs gi = g2;                        // images of update t+1 into the buffer flavour f does NOT read
              gi.parts = 1; gi.img_ctr = 1; gi.rng_ahead = 0; gi.quiet = 1;
              gi.rows = ride_rows;
              gi.x_obs = f ? x_obs : x_obs_b;
              Op heads_ride[2];                          // [0] the call's first update, [1] later ones (see pf_heads)
              for (int v = 0; v < 2; ++v) {
                heads_ride[v] = pf_heads[v];
                const HeadsFusedArgs* dv = d_ha + 1 + v;
                heads_ride[v].bytes = ops_rng[0].bytes;
                heads_ride[v].run = [dv, hnblk, hshape, gi, gx](hipStream_t s) { launch_heads_fused(hshape, hnblk, s, dv, &gi, gx); };
              }
              for (int v = (f ? 1 : 0); v < 3; ++v) {    // (the call's first update is always flavour 0)
                const std::vector<Op>& src = v == 0 ? ops_pf_first : (v == 1 ? ops_pf_mid : ops_pf_last);
                std::vector<Op>& dst = v == 0 ? ops_ride_first : (v == 1 ? ops_ride_mid[f] : ops_ride_last[f]);
                for (size_t k = 0; k < src.size(); ++k) {
                  const bool last_op = k + 1 == src.size();
                  if (v == 0 && k == 0) dst.push_back(first_g);
                  else if (src[k].tag == "conv_stack_fwd") dst.push_back(f ? conv_alt : src[k]);
                  else if (src[k].tag == "wgrad_conv") dst.push_back(f ? wgrad_conv_alt : src[k]);
                  else if (src[k].tag == "heads" && v != 2) dst.push_back(heads_ride[v == 0 ? 0 : 1]);
                  else if (last_op && v != 2) dst.push_back(rx);
                  else dst.push_back(src[k]);
                }
              }
            }
            ride_ok = true;
            if (getenv("GRL_PLAN_DUMP"))
              fprintf(stderr, "grl plan: gather_ride  %d image-gather workgroups of the next update ride on the head launch (%d tiles per row, %d rows each); extras on the reduction launch\n",
                      gx * (2 * B / ride_rows), gx, ride_rows);
          }
        }
      }
    }
  }

  // =============================================================== apply
  {
    Op op; op.tag = "adam_polyak";
    op.bytes = (double)n_train * 4 * 7 + (double)n_polyak * 4 * 2;
    grl_ctx* self = this;
    op.run = [self](hipStream_t s) {
      AdamArgs aa;
      aa.params = self->params; aa.grads = self->grads; aa.m = self->adam_m; aa.v = self->adam_v;
      aa.n_train = self->n_train; aa.sc = self->sc; aa.grad_scale = self->grad_scale; aa.tau = self->cfg.tau; aa.eps = 1e-8f;
      aa.src_ofs = self->vf_off; aa.n_polyak = self->n_polyak; aa.target = self->params + self->tgt_off;
      const int blocks = (int)std::min<int64_t>(2048, (self->n_train + 255) / 256);
      hipLaunchKernelGGL(adam_polyak_kernel, dim3(blocks), dim3(256), 0, s, aa);
    };
    ops_apply.push_back(op);
  }

  // =============================================================== act path (batch NA, pi net only)
  {
    ax = cnn ? wk.f32((int64_t)NA * img_elems) : nullptr;
    afeat = wk.f32((int64_t)NA * ldf);
    // noise in, actions out: page-locked HOST memory that the last launch of the act path reads and writes itself -- 320 bytes
    // each way for 16 environments, for which a copy-engine transfer in the stream costs more than the whole policy head
    if (!dry) {    // (grl_query_sizes plans without a device)
      if (hipHostMalloc((void**)&act_io_host, (size_t)2 * NA * A * 4, 0) != hipSuccess) return fail(GRL_ERR_HIP, "hipHostMalloc failed");
      memset(act_io_host, 0, (size_t)2 * NA * A * 4);
      // (GRL_TUNE act_poll=0 keeps the stream synchronisation at the end of grl_act)
      if (tune_int("act_poll", 1) != 0 && hipHostMalloc((void**)&act_done_host, 64, hipHostMallocCoherent) == hipSuccess) *act_done_host = 0u;
      else act_done_host = nullptr;
    }
    a_eps = act_io_host;
    a_out = act_io_host + (size_t)NA * A;
    alloc_head(ahPI, NA, 2, A);
    ActIngestArgs ia;
    memset(&ia, 0, sizeof(ia));
    ia.obs = stg_obs; ia.n = NA; ia.hw = hw * hw; ia.c_obs = c.obs_channels; ia.c_img = C_img; ia.n_direct = nd;
    ia.vec_dim = cnn ? 0 : c.obs_dim; ia.scale_div = cnn ? 255.f : 1.f;
    ia.x = cnn ? ax : afeat; ia.ldx = cnn ? img_elems : ldf; ia.d = afeat + 512; ia.ldd = ldf;
    ActIngestArgs ian = ia;
    ian.normalize = 1; ian.clip_obs = c.clip_obs;
    ian.mean = s_mean; ian.stdv = s_std; ian.dmean = s_dmean; ian.dstd = s_dstd;
    {   // entry launch: [observed by grl_observe | handed to grl_act] x [raw: VecNormalize applied here | already normalised]
      const int elems = cnn ? img_elems : c.obs_dim;
      for (int v = 0; v < 4; ++v) {
        ActIngestArgs iv = (v & 1) ? ian : ia;
        if (v & 2) iv.obs = ob_latest;
        Op op; op.tag = "act_ingest";
        op.run = [iv, elems](hipStream_t s) {
          hipLaunchKernelGGL(act_ingest_kernel, dim3((elems + 255) / 256, iv.n), dim3(256), 0, s, iv);
        };
        ops_act_in[v].push_back(op);
      }
    }
    // the policy head on the matrix cores (act_mfma.h) for the shapes it covers (GRL_TUNE act_mfma=0: the VALU kernel): the
    // extractor's dense layer then runs as a split-K GEMM whose partial sums the head kernel adds up while it stages its input
    const bool mfma_heads = act_mfma_built() && am_shape_ok(F, L, hid, A) && tune_int("act_mfma", 1) != 0;
    float* fc_parts = nullptr;
    int fc_split = 0;
    if (cnn) {
      aa1 = wk.f32((int64_t)NA * 225 * 32); aa2 = wk.f32((int64_t)NA * 36 * 64); aa3 = wk.f32((int64_t)NA * 16 * 64);
      float* io[4] = {ax, aa1, aa2, aa3};
      if (conv_stack) {     // one workgroup per observation walks conv1 -> conv2 -> conv3 (round 4: three launches, 4.9 + 9.2 + 9.2 us)
        ConvStackArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.nets[0].x = ax;
        for (int l = 0; l < 3; ++l) { ca.nets[0].w[l] = P + ex[0].w[l]; ca.nets[0].b[l] = P + ex[0].b[l]; }
        ca.nets[0].a3 = aa3;
        ca.B = NA; ca.n_nets = 1;
        const int Ci = C_img;
        Op op; op.tag = "act_conv";
        op.run = [ca, Ci](hipStream_t s) { launch_conv_stack_fwd(Ci, ca, s); };
        ops_act.push_back(op);
      } else
      for (int l = 0; l < 3; ++l) {
        ConvGeom ag = cg[l];
        ag.ldx = ag.ldy = 0;                      // the acting pass keeps dense buffers of its own
        ConvFwdTabs t = conv_fwd_tabs(ag, NA);
        add_launch(ops_act, "act_conv", 0,
                   {conv_fwd(io[l], t, ag, P + ex[0].w[l], P + ex[0].b[l], io[l + 1], ACT_RELU, 0.f)});
      }
      if (mfma_heads) {
        IgemmProb p = dense_fwd(aa3, 1024, 1024, nullptr, 0, 0, NA, P + ex[0].fw, 512, nullptr, nullptr, 512, ACT_NONE);
        set_split(p, AM_MAXP);       // one tile walking 32 slabs (13.6 us for 16 rows) -> 4 x the tiles, 8 slabs each
        fc_split = p.split;
        fc_parts = wk.f32((int64_t)NA * 512 * p.split);
        p.c = fc_parts;
        add_launch(ops_act, "act_fc", 0, {p});
      } else {
        add_launch(ops_act, "act_fc", 0,
                   {dense_fwd(aa3, 1024, 1024, nullptr, 0, 0, NA, P + ex[0].fw, 512, P + ex[0].fb, afeat, ldf, ACT_RELU)});
      }
    }
    // the policy head: one launch (act_heads_kernel) for layer widths it covers, else a launch per layer + the output launch.
    // Two variants (deterministic / sampled) so that each is a static graph.
    bool one_launch = F <= ACT_HEADS_MAX_IN && A <= 64;
    for (int l = 0; l < L; ++l) one_launch = one_launch && hid[l] <= ACT_HEADS_MAX_HID;
    for (int det = 0; det < 2; ++det) {
      std::vector<Op>& tail = det ? ops_act_det : ops_act_sto;
      if (one_launch) {
        ActHeadsArgs ha;
        memset(&ha, 0, sizeof(ha));
        ha.x = afeat; ha.ldx = ldf; ha.K0 = F; ha.L = L;
        for (int l = 0; l < L; ++l) { ha.w[l] = P + m_pi.w[l]; ha.b[l] = P + m_pi.b[l]; ha.hid[l] = hid[l]; }
        for (int k = 0; k < 2; ++k) { ha.ow[k] = P + m_pi.ow[k]; ha.ob[k] = P + m_pi.ob[k]; }
        ha.A = A; ha.eps = a_eps; ha.mu = ahPI.out[0]; ha.ls = ahPI.out[1]; ha.out = a_out; ha.rows = NA; ha.deterministic = det;
        Op op; op.tag = "act_heads";
        if (mfma_heads) {
          if (fc_split > 0) {
            ha.x_parts = fc_parts; ha.n_parts = fc_split; ha.part_stride = (long)NA * 512; ha.ld_parts = 512;
            ha.x_bias = P + ex[0].fb; ha.n_sum = 512;
          }
          ha.done = act_done_host;
          act_done_wgs = act_done_host ? (unsigned)((NA + HT_RB - 1) / HT_RB) : 0u;
          op.run = [ha](hipStream_t s) { launch_act_heads_mfma(ha, s); };
        } else {
          op.run = [ha](hipStream_t s) { hipLaunchKernelGGL(act_heads_kernel, dim3(ha.rows), dim3(256), 0, s, ha); };
        }
        tail.push_back(op);
        continue;
      }
      for (int l = 0; l < L; ++l)
        add_launch(tail, "act_head", 0, {head_layer(m_pi, P, ahPI, l, afeat, ldf, F, nullptr, 0, 0, NA)});
      add_launch(tail, "act_head", 0, {head_out(m_pi, P, ahPI, 0, NA), head_out(m_pi, P, ahPI, 1, NA)});
      const float* mu = ahPI.out[0]; const float* ls = ahPI.out[1]; const float* ep = a_eps; float* ao = a_out;
      const int rows = NA, Ad = A;
      Op op; op.tag = "act_out";
      op.run = [mu, ls, ep, ao, rows, Ad, det](hipStream_t s) {
        hipLaunchKernelGGL(act_out_kernel, dim3((rows * Ad + 255) / 256), dim3(256), 0, s, mu, ls, ep, rows, Ad, det, ao);
      };
      tail.push_back(op);
    }
  }

  // =============================================================== Keras auto-encoder (A.9), batch NA
  {
    const ConvGeom eg[3] = {{64, 64, 1, 7, 7, 2, 2, 32, 32, 32}, {32, 32, 32, 5, 5, 2, 1, 16, 16, 32},
                            {16, 16, 32, 3, 3, 2, 0, 8, 8, 32}};
    const int64_t wn[8] = {7 * 7 * 32, 32, 5 * 5 * 32 * 32, 32, 3 * 3 * 32 * 32, 32, 2048 * 100, 100};
    for (int k = 0; k < 8; ++k) enc_w[k] = wk.f32(wn[k]);
    ex_in = wk.f32((int64_t)NA * 4096);
    ec1 = wk.f32((int64_t)NA * 32 * 32 * 32); ec2 = wk.f32((int64_t)NA * 16 * 16 * 32);
    ec3 = wk.f32((int64_t)NA * 8 * 8 * 32); eout = wk.f32((int64_t)NA * 100);
    float* io[4] = {ex_in, ec1, ec2, ec3};
    for (int l = 0; l < 3; ++l) {
      ConvFwdTabs t = conv_fwd_tabs(eg[l], NA);
      add_launch(ops_enc, "enc_conv", 0,
                 {conv_fwd(io[l], t, eg[l], enc_w[2 * l], enc_w[2 * l + 1], io[l + 1], ACT_LEAKY, 0.1f)});
    }
    IgemmProb p = dense_fwd(ec3, 2048, 2048, nullptr, 0, 0, NA, enc_w[6], 100, enc_w[7], eout, 100, ACT_LEAKY);
    p.act_alpha = 0.1f;
    add_launch(ops_enc, "enc_dense", 0, {p});
  }

  // debug taps
  dbg["feat_pi"] = {feat[0], (int64_t)B * ldf};
  dbg["feat_vf"] = {feat[1], (int64_t)B * ldf};
  dbg["feat_tgt"] = {feat[2], (int64_t)B * ldf};
  if (cnn) {
    dbg["x_obs"] = {x_obs, (int64_t)B * img_elems};
    dbg["x_next"] = {x_next, (int64_t)B * img_elems};
    dbg["a1_pi"] = {a1[0], (int64_t)B * 225 * 32};   // (side-by-side layout: the first half of the pair buffer, stride 64)
    dbg["a2_pi"] = {a2[0], (int64_t)B * 36 * 64};
    dbg["a3_pi"] = {a3[0], (int64_t)B * 1024};
    dbg["a1_pair"] = {a1[0], (int64_t)B * 225 * 64};   // [B * 225][pi 0..31 | values_fn 32..63]
    dbg["a2_vf"] = {a2[1], (int64_t)B * 36 * 64};
    dbg["a3_vf"] = {a3[1], (int64_t)B * 1024};
    dbg["g1_vf"] = {g1[1], (int64_t)B * 225 * 32};
    dbg["g2_vf"] = {g2[1], (int64_t)B * 36 * 64};
    dbg["g3_vf"] = {g3[1], (int64_t)B * 1024};
    dbg["dfeat_pi"] = {dfeat[0], (int64_t)B * ldf};
    dbg["dfeat_vf"] = {dfeat[1], (int64_t)B * ldf};
  }
  dbg["mu"] = {hPI.out[0], (int64_t)B * A};
  dbg["log_std"] = {hPI.out[1], (int64_t)B * A};
  dbg["pi"] = {pi_a, (int64_t)B * A};
  dbg["logp"] = {logp, B};
  dbg["qf1"] = {hQF1.out[0], B}; dbg["qf2"] = {hQF2.out[0], B};
  dbg["v"] = {hVF.out[0], B}; dbg["v_tgt"] = {hTGT.out[0], B};
  dbg["qf1_pi"] = {hQF1PI.out[0], B}; dbg["qf2_pi"] = {hQF2PI.out[0], B};
  dbg["rew"] = {rew, B}; dbg["done"] = {done, B}; dbg["act"] = {act, (int64_t)B * A};
  dbg["dmu"] = {dmu, (int64_t)B * A}; dbg["dls"] = {dls, (int64_t)B * A}; dbg["da_pi"] = {da_pi, (int64_t)B * A};
  dbg["grads"] = {grads, n_train};
  dbg["adam_m"] = {adam_m, n_train};
  dbg["adam_v"] = {adam_v, n_train};
  dbg["rp_obs"] = {rp_obs, cap * obs_store}; dbg["rp_next"] = {rp_next, cap * obs_store};
  dbg["rp_dobs"] = {rp_dobs, cap * std::max(nd, 1)}; dbg["rp_dnext"] = {rp_dnext, cap * std::max(nd, 1)};
  dbg["rp_act"] = {rp_act, cap * A}; dbg["rp_rew"] = {rp_rew, cap}; dbg["rp_done"] = {rp_done, cap};
  dbg["idx_raw"] = {(const float*)idx_buf, (int64_t)2 * B};     // replay indices of the last minibatch: int64 viewed as float pairs
  return GRL_OK;
}
