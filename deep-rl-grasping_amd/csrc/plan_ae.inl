// plan_ae.inl: launch plan of the depth auto-encoder training step (grl_ctx::plan_ae) -- part of engine.hip (included there, same translation unit: the plans are methods of grl_ctx).

// --------------------------------------------------------------------------------------------------
// Depth auto-encoder training: encoders.py:90-124 network, :127 MSE, :130 Adam,
// config/encoder.yaml (7/5/3 kernels, 32 filters, stride 2, encoding 100, lr 2e-4, batch 128).
// Parameters are the 16 Keras tensors in model.h5 order; one step = forward, MSE, backward, Keras-Adam.
int grl_ctx::plan_ae() {
  const grl_config& c = cfg;
  cnn = false;
  B = c.batch_size; NA = std::max(1, c.act_batch); A = 1; L = 0;
  img_elems = 4096; F = 100; Fc = 0; ldf = 100; C_img = 1; hw = 64;
  const float LA = 0.1f;   // LeakyReLU alpha (encoders.py:87)
  // ---------------- parameter layout (Keras creation order == HDF5 order of the shipped model.h5)
  struct CL { const char* name; int kh, cin, cout; };
  const CL enc[3] = {{"encoder/conv2d_1", 7, 1, 32}, {"encoder/conv2d_2", 5, 32, 32}, {"encoder/conv2d_3", 3, 32, 32}};
  const CL dec[3] = {{"decoder/conv2d_4", 3, 32, 32}, {"decoder/conv2d_5", 5, 32, 32}, {"decoder/conv2d_6", 7, 32, 1}};
  int64_t ew[3], eb[3], dw[3], db[3];
  for (int l = 0; l < 3; ++l) {
    ew[l] = add_var(std::string(enc[l].name) + "/kernel", {enc[l].kh, enc[l].kh, enc[l].cin, enc[l].cout}, true);
    eb[l] = add_var(std::string(enc[l].name) + "/bias", {enc[l].cout}, true);
  }
  const int64_t edw = add_var("encoder/dense_1/kernel", {2048, 100}, true), edb = add_var("encoder/dense_1/bias", {100}, true);
  const int64_t ddw = add_var("decoder/dense_2/kernel", {100, 2048}, true), ddb = add_var("decoder/dense_2/bias", {2048}, true);
  for (int l = 0; l < 3; ++l) {
    dw[l] = add_var(std::string(dec[l].name) + "/kernel", {dec[l].kh, dec[l].kh, dec[l].cin, dec[l].cout}, true);
    db[l] = add_var(std::string(dec[l].name) + "/bias", {dec[l].cout}, true);
  }
  n_train = n_params; tgt_off = n_params; vf_off = 0; n_polyak = 0; ent_off = 0;
  // ---------------- arenas
  params = st.f32(n_params);
  adam_m = st.f32(n_train);
  adam_v = st.f32(n_train);
  sc = (DevScalars*)st.take(sizeof(DevScalars));
  s_mean = (double*)st.take(8); s_std = (double*)st.take(8); s_dmean = (double*)st.take(8); s_dstd = (double*)st.take(8);
  s_ret = (double*)st.take(8);
  grads = gr.f32(n_train);
  rp_obs = rp_next = rp_dobs = rp_dnext = rp_act = rp_rew = rp_done = rp.f32(4);   // no replay on this path
  stg_n = std::max(NA, 64);
  stg_obs = stg_next = stg_act = stg_rew = stg_done = wk.f32(4);
  idx_buf = (int64_t*)wk.take(8); eps_buf = wk.f32(4);
  for (int n = 0; n < 3; ++n) feat[n] = wk.f32(4);
  const float* P = params;
  // activations [B, H, W, C] (NHWC) and their gradients
  auto T = [&](int h, int ch) { return wk.f32((int64_t)B * h * h * ch); };
  ae_x = T(64, 1);
  float *e3 = T(8, 32), *z = wk.f32((int64_t)B * 100), *dh = T(8, 32);      // (e1 / e2 exist only in their bordered form, below)
  float *d4 = T(16, 32), *d5 = T(32, 32), *out = T(64, 1);
  // inputs of the padded convolutions live in zero-bordered buffers (border written once): the 'same'
  // convolution becomes a 'valid' one over the bordered image, so neither the forward GEMM nor the weight
  // gradient needs per-tap bounds masks and both run on the vectorised kernel
  auto TP = [&](int h, int lo, int hi, int ch) {
    const int64_t n = (int64_t)B * (h + lo + hi) * (h + lo + hi) * ch;
    float* b = wk.f32(n);
    zero_once.push_back({b, (size_t)n * 4});
    return b;
  };
  float *x_p = TP(64, 2, 3, 1), *e1_p = TP(32, 1, 2, 32), *e2_p = TP(16, 0, 1, 32);
  float *u4 = TP(16, 1, 1, 32), *u5 = TP(32, 2, 2, 32);
  float* g_pad = nullptr;
  float *g_out = T(64, 1), *g_d5 = T(32, 32), *g_u5 = T(32, 32), *g_d4 = T(16, 32), *g_u4 = T(16, 32);
  float *g_dh = T(8, 32), *g_z = wk.f32((int64_t)B * 100), *g_e3 = T(8, 32), *g_e2 = T(16, 32), *g_e1 = T(32, 32);
  // geometry: encoder convs 'SAME' stride 2 (TF asymmetric padding: low pad 2 / 1 / 0), decoder convs 'SAME' stride 1
  const ConvGeom ge[3] = {{64, 64, 1, 7, 7, 2, 2, 32, 32, 32}, {32, 32, 32, 5, 5, 2, 1, 16, 16, 32}, {16, 16, 32, 3, 3, 2, 0, 8, 8, 32}};
  const ConvGeom gd[3] = {{16, 16, 32, 3, 3, 1, 1, 16, 16, 32}, {32, 32, 32, 5, 5, 1, 2, 32, 32, 32}, {64, 64, 32, 7, 7, 1, 3, 64, 64, 1}};
  // the same convolutions as 'valid' ones over the bordered inputs (forward + weight gradient)
  const ConvGeom gev[3] = {{69, 69, 1, 7, 7, 2, 0, 32, 32, 32}, {35, 35, 32, 5, 5, 2, 0, 16, 16, 32}, {17, 17, 32, 3, 3, 2, 0, 8, 8, 32}};
  const ConvGeom gdv[2] = {{18, 18, 32, 3, 3, 1, 0, 16, 16, 32}, {36, 36, 32, 5, 5, 1, 0, 32, 32, 32}};
  ConvFwdTabs fte[3], ftd[2];
  for (int l = 0; l < 3; ++l) fte[l] = conv_fwd_tabs(gev[l], B);
  for (int l = 0; l < 2; ++l) ftd[l] = conv_fwd_tabs(gdv[l], B);
  auto elem = [&](const char* tag, std::function<void(hipStream_t)> f) {
    Op op; op.tag = tag; op.run = std::move(f);
    ops_ae.push_back(op);
  };
  auto up = [&](const float* h, float* u, int H, int border) {
    const int Bn = B;
    elem("ae_upsample", [=](hipStream_t s) {
      const long quads = (long)Bn * 2 * H * 2 * H * 8;
      hipLaunchKernelGGL(upsample2_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, h, u, Bn, H, H, 32,
                         border, border);
    });
  };
  auto up_bwd = [&](const float* gu, const float* h, float* gh, int H) {
    const int Bn = B;
    elem("ae_upsample_bwd", [=](hipStream_t s) {
      const long n = (long)Bn * H * H * 32;
      hipLaunchKernelGGL(upsample2_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gu, h, gh, Bn, H, H, 32, LA);
    });
  };
  // =============================================================== forward
  float* W6p = wk.f32(4 * 56 * 32);   // output kernel flipped + padded, four copies (backward-data of the output convolution, below)
  float* W1p = wk.f32(56 * 32);       // first encoder kernel, rows padded to 8 taps (its forward, here)
  {
    const float *W6 = P + dw[2], *W1 = P + ew[0];
    // (one launch with the copy of the minibatch into its bordered buffer: ae_prep_pad_kernel)
    const int n_prep = (56 * 32 + 255) / 256;
    const long total = (long)B * 4096;
    const float* xin = ae_x;
    float* xpad = x_p;
    elem("ae_kernel_prep", [=](hipStream_t s) {
      hipLaunchKernelGGL(ae_prep_pad_kernel, dim3((unsigned)(n_prep + (total + 255) / 256)), dim3(256), 0, s, W6, W6p, 32, W1, W1p, 32,
                         n_prep, xin, xpad, total, 64, 64, 1, 2, 3);
    });
  }
  // Encoder activations e1 / e2 are consumed in their zero-bordered form (next convolution, weight gradient) and as the
  // LeakyReLU mask of the backward-data launches: the convolution that produces them writes the bordered layout directly (output
  // row table, c_tab_i) and the backward-data launches read their mask through a row table of its own (IgemmProb.m_tab_i) -- no
  // plain copies, no pad-copy launches (round 5: two launches, 21 MB read + written at B = 128).
  auto bordered_rows = [&](int H, int lo, int hi, int C) {     // output pixel (n, oh, ow) -> its offset in [N, H+lo+hi, W+lo+hi, C]
    std::vector<int32_t> ct((size_t)B * H * H);
    const int Hp = H + lo + hi;
    for (int n = 0; n < B; ++n)
      for (int oh = 0; oh < H; ++oh)
        for (int ow = 0; ow < H; ++ow) ct[((size_t)n * H + oh) * H + ow] = ((n * Hp + oh + lo) * Hp + ow + lo) * C;
    return ct;
  };
  {
    const float* in[3] = {x_p, e1_p, e2_p};
    float* o[3] = {nullptr, nullptr, e3};
    float* ob[3] = {e1_p, e2_p, nullptr};
    const int oH[2] = {32, 16}, olo[2] = {1, 0}, ohi[2] = {2, 1};
    auto to_bordered = [&](IgemmProb& p, int l) {
      if (l > 1) return;
      p.c = ob[l];
      p.c_tab_i = upload_vec(wk, bordered_rows(oH[l], olo[l], ohi[l], 32));
      p.vflags |= VF_CT4;
    };
    for (int l = 0; l < 3; ++l) {      // (x_p was filled by the step's first launch)
      if (l == 0) {
        // 7 x 8 taps over the bordered 69 x 69 image (see ae_kernel_prep): the reduction index is (kh, j), j = 0 .. 7
        std::vector<int32_t> tr(57, 0);
        for (int kh = 0; kh < 7; ++kh)
          for (int j = 0; j < 8; ++j) tr[kh * 8 + j] = kh * 69 + j;
        IgemmProb p = conv_fwd(in[0], fte[0], gev[0], W1p, P + eb[0], o[0], ACT_LEAKY, LA);
        p.K = 56;
        p.p_tab_r = upload_vec(wk, tr);
        p.vflags |= VF_P_TABS;      // 4-runs along the taps at dword-aligned offsets (16-byte buffer loads take them on gfx950)
        to_bordered(p, 0);
        add_launch(ops_ae, "ae_enc_conv", 0, {p});
        continue;
      }
      IgemmProb p = conv_fwd(in[l], fte[l], gev[l], P + ew[l], P + eb[l], o[l], ACT_LEAKY, LA);
      to_bordered(p, l);
      add_launch(ops_ae, "ae_enc_conv", 0, {p});
    }
  }
  {
    IgemmProb p = dense_fwd(e3, 2048, 2048, nullptr, 0, 0, B, P + edw, 100, P + edb, z, 100, ACT_LEAKY);
    p.act_alpha = LA;
    add_launch(ops_ae, "ae_dense", 0, {p});
    IgemmProb q = dense_fwd(z, 100, 100, nullptr, 0, 0, B, P + ddw, 2048, P + ddb, dh, 2048, ACT_LEAKY);
    q.act_alpha = LA;
    add_launch(ops_ae, "ae_dense", 0, {q});
  }
  up(dh, u4, 8, 1);
  add_launch(ops_ae, "ae_dec_conv", 0, {conv_fwd(u4, ftd[0], gdv[0], P + dw[0], P + db[0], d4, ACT_LEAKY, LA)});
  up(d4, u5, 16, 2);
  add_launch(ops_ae, "ae_dec_conv", 0, {conv_fwd(u5, ftd[1], gdv[1], P + dw[1], P + db[1], d5, ACT_LEAKY, LA)});
  // (no up(d5): nothing reads the 64 x 64 x 32 up-sampled image any more -- 67 MB at B = 128, round 5)
  // output conv (7x7 'same', 32 -> 1): N = 1 wastes the matrix cores, so T[tap, p] = W[tap, :] . u[p, :] as a GEMM with
  // M = 49, then a 49-tap gather-sum (ae_kernels.h: ae_tapsum_kernel).  u6 is d5 with every pixel repeated 2 x 2
  // (UpSampling2D), so T is formed over the pixels of d5 -- a quarter of the columns, the same products -- and the gather-sum
  // reads T at (ih / 2, iw / 2): 51.9 -> 13.5 us for the GEMM (round 5: over the 4096 pixels of u6)
  const long ldT = (long)B * 1024;
  float* Tt = wk.f32(49 * ldT);
  add_launch(ops_ae, "ae_out_conv", 1, {dense_bwd({{P + dw[2], 32, 32, d5}}, 49, 0, (int)ldT, Tt, (int)ldT, nullptr)});
  // The forward pass ends with the 49-tap gather-sum; a TRAINING step forms the loss and the output gradient in the same launch
  // (ae_tapsum_mse_kernel: one launch and 6 us less than gather-sum + MSE), the forward-only path
  // (Model.predict / evaluate) ends with the plain gather-sum.
  const long npix = (long)B * 4096;
  const int n_part = (int)((npix + 255) / 256);
  {
    const float* b6 = P + db[2];
    Op op; op.tag = "ae_out_tapsum";
    op.run = [=](hipStream_t s) {
      hipLaunchKernelGGL(ae_tapsum_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, (const float*)Tt, ldT, b6, out, npix);
    };
    ae_out = out;
    ops_ae_fwd = ops_ae;            // everything so far + the gather-sum: the forward pass
    ops_ae_fwd.push_back(op);
  }
  // =============================================================== loss
  {
    g_pad = wk.f32((int64_t)B * 4900);
    zero_once.push_back({g_pad, (size_t)B * 4900 * 4});   // the 3-pixel border stays zero
    float* part = wk.f32(n_part);
    float* partial_g = wk.f32(n_part);
    float* gp4 = wk.f32((int64_t)4 * B * 1296);
    zero_once.push_back({gp4, (size_t)4 * B * 1296 * 4});   // the 2-pixel borders stay zero
    ae_gp4 = gp4;
    MseArgs ma{out, ae_x, g_out, part, npix, g_pad, partial_g, gp4, (long)B * 1296};
    const float lr = c.lr;
    DevScalars* scp = sc;
    float* gb6 = grads + db[2];
    TapMseArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.T = Tt; ta.ldT = ldT; ta.bias = P + db[2]; ta.n_pix = npix; ta.m = ma;
    elem("ae_out_tapsum_mse", [=](hipStream_t s) {
      hipLaunchKernelGGL(ae_tapsum_mse_kernel, dim3(n_part), dim3(256), 0, s, ta);
      hipLaunchKernelGGL(ae_finish_kernel, dim3(1), dim3(256), 0, s, (const float*)part, (const float*)partial_g, n_part,
                         ma.n_total, lr, scp, gb6);
    });
  }
  // =============================================================== backward
  std::vector<IgemmProb> wgc, wgd;
  auto cw = [&](const float* x, const ConvFwdTabs& t, const ConvGeom& g, const float* gy, int64_t w_off, int64_t b_off, int split) {
    IgemmProb p = conv_wgrad(x, t, g, gy, nullptr, split);
    p.c = wk.f32(p.slab_stride * p.split);
    add_wgrad(wgc, p, w_off, 0, g.K(), b_off);
  };
  // mask_lo / mask_hi >= 0: `mask` is the zero-bordered form of the activations ([N, H+lo+hi, W+lo+hi, C]); its rows are found
  // through a table derived from the class's output row table (offsets in the plain [N, H, W, C] layout of dx)
  auto cb = [&](const char* tag, const float* gy, const ConvGeom& g, const float* w, float* dx, const float* mask,
                int mask_lo = -1, int mask_hi = -1) {
    std::vector<IgemmProb> pr;
    for (auto& cl : conv_bwd_tabs(g, B)) {
      IgemmProb p = conv_bwd(gy, cl, g, w, dx, mask);
      p.act_alpha = LA;                                  // LeakyReLU gradient where a mask is given
      if (mask && mask_lo >= 0) {
        std::vector<int32_t> mt(cl.ct_host.size());
        const int Hp = g.H + mask_lo + mask_hi, Wp = g.W + mask_lo + mask_hi;
        for (size_t i = 0; i < mt.size(); ++i) {
          const int ct = cl.ct_host[i];
          if (ct < 0) { mt[i] = 0; continue; }            // (row not stored: its mask is never read)
          const int pix = ct / g.C, n = pix / (g.H * g.W), y = (pix / g.W) % g.H, x = pix % g.W;
          mt[i] = ((n * Hp + y + mask_lo) * Wp + x + mask_lo) * g.C;
        }
        p.m_tab_i = upload_vec(wk, mt);
      }
      pr.push_back(p);
    }
    add_launch(ops_ae, tag, 1, pr);
  };
  {
    // output conv (7x7, 32 -> 1), weight gradient: dW[tap, c] = sum_p g[p - shift(tap)] u6[p, c] with u6[p] = d5[p / 2]:
    //   dW[tap, c] = sum over the four sub-positions s of a 2 x 2 block, sum over pixels q of d5:  g[2 q + s - shift(tap)] d5[q, c]
    // -- four GEMMs (M = 49 taps, N = 32 channels, K = pixels of d5) whose P operand walks ONE sub-position plane of the
    // de-interleaved output gradient (MseArgs.gp4) with unit stride: for (s, tap), t = s - shift lands in plane (t mod 2) at
    // offset floor(t / 2).  Same products as over u6, which is no longer formed; the 4 x 64 slabs are summed as one list.
    std::vector<int32_t> tr((size_t)B * 1024);
    for (int n = 0; n < B; ++n)
      for (int qy = 0; qy < 32; ++qy)
        for (int qx = 0; qx < 32; ++qx) tr[((size_t)n * 32 + qy) * 32 + qx] = n * 1296 + (qy + 2) * 36 + (qx + 2);
    const int32_t* d_tr = upload_vec(wk, tr);
    const long plane = (long)B * 1296;
    std::vector<IgemmProb> four;
    float* slabs = nullptr;
    int per = 0;
    int64_t sstride = 0;
    for (int sp = 0; sp < 4; ++sp) {
      std::vector<int32_t> ti(49);
      for (int kh = 0; kh < 7; ++kh)
        for (int kw = 0; kw < 7; ++kw) {
          const int ty = (sp >> 1) - (kh - 3), tx = (sp & 1) - (kw - 3);
          const int py = ty & 1, px = tx & 1, oy = (ty - py) / 2, ox = (tx - px) / 2;       // t = 2 o + p, p in {0, 1}
          ti[kh * 7 + kw] = (int32_t)((py * 2 + px) * plane + oy * 36 + ox);
        }
      IgemmProb p = blank();
      p.M = 49; p.N = 32; p.K = B * 1024;
      p.p_base[0] = ae_gp4; p.p_tab_i = upload_vec(wk, ti); p.p_tab_r = d_tr; single_part(p);
      p.q_base[0] = d5; p.q_ld_r[0] = 32; p.q_ld_j[0] = 1;
      p.ldc = 32;
      p.vflags |= VF_P_TABS;       // 4-runs along the pixel index: rows of 32 pixels, quads never straddle one
      set_split(p, 64);
      if (sp == 0) { per = p.split; sstride = p.slab_stride; slabs = wk.f32(sstride * per * 4); }
      p.c = slabs + (int64_t)sp * per * sstride;
      four.push_back(p);
    }
    {
      ReduceDesc r;
      memset(&r, 0, sizeof(r));
      r.src = slabs; r.splits = 4 * per; r.slab_stride = sstride;
      r.dst = grads + dw[2]; r.n = 49 * 32;
      reduces.push_back(r);
    }
    add_launch(ops_ae, "ae_out_wgrad", 0, four, "", 0, {}, 3);      // (32 x 64 tiles, 2-way k split: 47.6 against 53.4 us for the N <= 32 default)
  }
  {
    // backward-data of the output conv AND of the up-sampling in front of it, in one product: the gradient of a pixel q of d5 is
    // the sum over its 2 x 2 pixels of u6, each the 7 x 8-tap product of the round-5 launch --
    //   g_d5[q, c] = LeakyReLU'(d5[q, c]) * sum_{s in 2x2} sum_{kh, j} g_pad[pix(2 q + s) - shift(kh, j)] Wp[(kh, j), c]
    // a GEMM with M = pixels of d5, N = 32, K = 4 x 56 (the flipped, padded kernel repeated for the four sub-positions:
    // ae_kernel_prep), the LeakyReLU gradient in its epilogue.  g_u6 -- 67 MB written by one launch and read back by the next at
    // B = 128 -- no longer exists, and neither does that up-sampling backward launch.
    float* Wp = W6p;                  // (written by ae_kernel_prep at the start of the step: four copies)
    std::vector<int32_t> ti((size_t)B * 1024), tr(224);
    for (int n = 0; n < B; ++n)
      for (int qy = 0; qy < 32; ++qy)
        for (int qx = 0; qx < 32; ++qx) ti[((size_t)n * 32 + qy) * 32 + qx] = n * 4900 + (2 * qy + 6) * 70 + (2 * qx + 6);
    for (int sp = 0; sp < 4; ++sp)
      for (int kh = 0; kh < 7; ++kh)
        for (int j = 0; j < 8; ++j) tr[sp * 56 + kh * 8 + j] = (sp >> 1) * 70 + (sp & 1) - kh * 70 - 7 + j;
    IgemmProb p = blank();
    p.M = B * 1024; p.N = 32; p.K = 224;
    p.p_base[0] = g_pad; p.p_tab_i = upload_vec(wk, ti); p.p_tab_r = upload_vec(wk, tr); single_part(p);
    p.vflags |= VF_P_TABS;                       // 4-runs along the taps, dword-aligned offsets
    p.q_base[0] = Wp; p.q_ld_r[0] = 32; p.q_ld_j[0] = 1;
    p.c = g_d5; p.ldc = 32;
    p.relu_mask = d5; p.act_alpha = LA;
    set_split(p, 1);
    add_launch(ops_ae, "ae_out_conv_bwd", 0, {p});
  }
  cw(u5, ftd[1], gdv[1], g_d5, dw[1], db[1], 32);
  cb("ae_dec_conv_bwd", g_d5, gd[1], P + dw[1], g_u5, nullptr);
  up_bwd(g_u5, d4, g_d4, 16);
  cw(u4, ftd[0], gdv[0], g_d4, dw[0], db[0], 8);
  cb("ae_dec_conv_bwd", g_d4, gd[0], P + dw[0], g_u4, nullptr);
  up_bwd(g_u4, dh, g_dh, 8);
  {
    IgemmProb p = dense_wgrad(z, 100, 100, true, g_dh, 2048, 2048, B, nullptr, 1);
    p.c = wk.f32(p.slab_stride * p.split);
    add_wgrad(wgd, p, ddw, 0, 100, ddb);
    IgemmProb b = dense_bwd({{g_dh, 2048, 2048, P + ddw}}, B, 0, 100, g_z, 100, z);
    b.act_alpha = LA;
    add_launch(ops_ae, "ae_dense_bwd", 1, {b});
    IgemmProb p2 = dense_wgrad(e3, 2048, 2048, true, g_z, 100, 100, B, nullptr, 1);
    p2.c = wk.f32(p2.slab_stride * p2.split);
    add_wgrad(wgd, p2, edw, 0, 2048, edb);
    IgemmProb b2 = dense_bwd({{g_z, 100, 100, P + edw}}, B, 0, 2048, g_e3, 2048, e3);
    b2.act_alpha = LA;
    add_launch(ops_ae, "ae_dense_bwd", 1, {b2});
  }
  cw(e2_p, fte[2], gev[2], g_e3, ew[2], eb[2], 4);
  cb("ae_enc_conv_bwd", g_e3, ge[2], P + ew[2], g_e2, e2_p, 0, 1);
  cw(e1_p, fte[1], gev[1], g_e2, ew[1], eb[1], 16);
  cb("ae_enc_conv_bwd", g_e2, ge[1], P + ew[1], g_e1, e1_p, 1, 2);
  {
    // weight gradient of the 1-channel first convolution (7x7, stride 2): rows = taps, each kernel row padded to 8
    // so that a quad of rows is 4 neighbouring pixels of the bordered image (16-byte loads at dword alignment);
    // the padding rows (kw = 7) are computed and never reduced
    std::vector<int32_t> ti(57), tr((size_t)B * 1024);
    for (int kh = 0; kh < 7; ++kh)
      for (int j = 0; j < 8; ++j) ti[kh * 8 + j] = kh * 69 + j;
    ti[56] = 0;
    for (int n = 0; n < B; ++n)
      for (int oh = 0; oh < 32; ++oh)
        for (int ow = 0; ow < 32; ++ow) tr[((size_t)n * 32 + oh) * 32 + ow] = n * 69 * 69 + 2 * oh * 69 + 2 * ow;
    IgemmProb p = blank();
    p.M = 57; p.N = 32; p.K = B * 1024;
    p.p_base[0] = x_p; p.p_tab_i = upload_vec(wk, ti); p.p_tab_r = upload_vec(wk, tr); single_part(p);
    p.p_ones_i = 56;
    p.vflags |= VF_P_TABS;
    p.q_base[0] = g_e1; p.q_ld_r[0] = 32; p.q_ld_j[0] = 1;
    p.ldc = 32;
    set_split(p, 64);
    p.c = wk.f32(p.slab_stride * p.split);
    wgc.push_back(p);
    for (int kh = 0; kh < 7; ++kh) {
      ReduceDesc r;
      memset(&r, 0, sizeof(r));
      r.src = p.c + (int64_t)kh * 8 * 32; r.splits = p.split; r.slab_stride = p.slab_stride;
      r.dst = grads + ew[0] + (int64_t)kh * 7 * 32; r.n = 7 * 32;
      reduces.push_back(r);
    }
    ReduceDesc rb;
    memset(&rb, 0, sizeof(rb));
    rb.src = p.c + (int64_t)56 * 32; rb.splits = p.split; rb.slab_stride = p.slab_stride; rb.dst = grads + eb[0]; rb.n = 32;
    reduces.push_back(rb);
  }
  {
    // uniform launches for the vectorised kernel; whatever it cannot take goes to igemm_kernel
    std::vector<IgemmProb> ok_c, rest;
    for (auto& p : wgc) (v2_prob_ok(p, 2) && (p.K % 4) == 0 ? ok_c : rest).push_back(p);
    add_launch(ops_ae, "ae_wgrad_conv", 2, ok_c);
    add_launch(ops_ae, "ae_wgrad_small", 2, rest);
    add_launch(ops_ae, "ae_wgrad_dense", 2, wgd);
  }
  {
    std::vector<int2> rt = reduce_tiles();
    d_reduces = upload_vec(wk, reduces);
    int2* d_rt = upload_vec(wk, rt);
    const int ntiles = (int)rt.size();
    ReduceDesc* dr = d_reduces;
    LossArgs none;
    memset(&none, 0, sizeof(none));
    elem("reduce_slabs", [=](hipStream_t s) {
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ntiles), dim3(256), 0, s, dr, d_rt, ntiles, none, 0, AdamArgs{}, 0);
    });
  }
  {
    grl_ctx* self = this;
    elem("adam", [self](hipStream_t s) {
      AdamArgs aa;
      aa.params = self->params; aa.grads = self->grads; aa.m = self->adam_m; aa.v = self->adam_v;
      aa.n_train = self->n_train; aa.sc = self->sc; aa.grad_scale = 1.f; aa.tau = 0.f; aa.eps = 1e-7f;   // Keras epsilon
      aa.src_ofs = 0; aa.n_polyak = 0; aa.target = self->params;
      const int blocks = (int)std::min<int64_t>(2048, (self->n_train + 255) / 256);
      hipLaunchKernelGGL(adam_polyak_kernel, dim3(blocks), dim3(256), 0, s, aa);
    });
  }
  // =============================================================== encode path (batch NA) on the trained weights
  {
    ex_in = wk.f32((int64_t)NA * 4096);
    ec1 = wk.f32((int64_t)NA * 32 * 32 * 32); ec2 = wk.f32((int64_t)NA * 16 * 16 * 32);
    ec3 = wk.f32((int64_t)NA * 8 * 8 * 32); eout = wk.f32((int64_t)NA * 100);
    float* io[4] = {ex_in, ec1, ec2, ec3};
    for (int l = 0; l < 3; ++l) {
      ConvFwdTabs t = conv_fwd_tabs(ge[l], NA);
      add_launch(ops_enc, "enc_conv", 0, {conv_fwd(io[l], t, ge[l], P + ew[l], P + eb[l], io[l + 1], ACT_LEAKY, LA)});
    }
    IgemmProb p = dense_fwd(ec3, 2048, 2048, nullptr, 0, 0, NA, P + edw, 100, P + edb, eout, 100, ACT_LEAKY);
    p.act_alpha = LA;
    add_launch(ops_enc, "enc_dense", 0, {p});
    for (int k = 0; k < 8; ++k) enc_w[k] = nullptr;
    enc_loaded = true;
  }
  dbg["out"] = {out, (int64_t)B * 4096};
  dbg["z"] = {z, (int64_t)B * 100};
  dbg["e3"] = {e3, (int64_t)B * 2048};
  dbg["d5"] = {d5, (int64_t)B * 32 * 32 * 32};
  dbg["grads"] = {grads, n_train};
  dbg["adam_m"] = {adam_m, n_train};
  dbg["adam_v"] = {adam_v, n_train};
  return GRL_OK;
}
