// capi.inl: the C ABI of include/grl.h (extern "C" entry points over grl_ctx) -- part of engine.hip (included there, same translation unit: the plans are methods of grl_ctx).

// ==================================================================================================
// C ABI
// ==================================================================================================
static int check_cfg(const grl_config* c) {
  if (!c) return fail(GRL_ERR_INVALID, "null config");
  if (c->algo == GRL_ALGO_AE) {
    if (c->batch_size < 1 || c->batch_size > 4096) return fail(GRL_ERR_INVALID, "batch_size out of range");
    if (c->replay_capacity < 1) return fail(GRL_ERR_INVALID, "replay_capacity must be >= 1");
    return GRL_OK;
  }
  if (c->extractor < 0 || c->extractor > 2) return fail(GRL_ERR_INVALID, "extractor must be 0..2");
  if (c->n_layers < 1 || c->n_layers > GRL_MAX_LAYERS) return fail(GRL_ERR_INVALID, "n_layers out of range");
  for (int l = 0; l < c->n_layers; ++l)
    if (c->layers[l] < 1 || c->layers[l] > 4096) return fail(GRL_ERR_INVALID, "layer width out of range");
  if (c->batch_size < 1 || c->batch_size > 65536) return fail(GRL_ERR_INVALID, "batch_size out of range");
  if (c->act_dim < 1 || c->act_dim > 64) return fail(GRL_ERR_INVALID, "act_dim out of range");
  if (c->replay_capacity < 1) return fail(GRL_ERR_INVALID, "replay_capacity must be >= 1");
  if (c->algo < 0 || c->algo > 3) return fail(GRL_ERR_INVALID, "algo must be 0..3");
  if (c->algo != GRL_ALGO_SAC) {
    if (c->extractor != GRL_EXTRACTOR_MLP) return fail(GRL_ERR_INVALID, "DQN/BDQ run on vector observations (MLP extractor)");
    if (c->q_branches < 1 || c->q_branches > 16 || c->q_branches != c->act_dim)
      return fail(GRL_ERR_INVALID, "q_branches must be 1..16 and equal act_dim");
    if (c->q_bins < 2 || c->q_bins > 1024) return fail(GRL_ERR_INVALID, "q_bins out of range");
    if (c->q_n_common < 0 || c->q_n_common > GRL_MAX_LAYERS || c->q_n_branch < 1 || c->q_n_branch > GRL_MAX_LAYERS ||
        c->q_n_value < 1 || c->q_n_value > GRL_MAX_LAYERS)
      return fail(GRL_ERR_INVALID, "tower depths out of range (branch and value towers need >= 1 hidden layer)");
    if (c->q_per && c->batch_size > 1024) return fail(GRL_ERR_INVALID, "prioritised replay supports batch_size <= 1024");
    if (c->q_per && c->replay_capacity > (int64_t)PER_BLK * PER_BLK)
      return fail(GRL_ERR_INVALID, "prioritised replay supports up to 1024 x 1024 transitions (two-level segment tree)");
  }
  if (c->replay_rgb_u8) {
    const int c_img = c->obs_channels - ((c->extractor == GRL_EXTRACTOR_AUGMENTED && c->n_direct > 0) ? 1 : 0);
    if (c->algo != GRL_ALGO_SAC || c->extractor == GRL_EXTRACTOR_MLP || c_img != 4)
      return fail(GRL_ERR_INVALID, "replay_rgb_u8 needs SAC on RGB-D images with 4 image channels (R, G, B, depth)");
  }
  if (c->extractor == GRL_EXTRACTOR_MLP) {
    if (c->obs_dim < 1) return fail(GRL_ERR_INVALID, "obs_dim must be >= 1 for the MLP extractor");
  } else {
    if (c->img_hw != 64) return fail(GRL_ERR_INVALID, "only 64x64 images (camera_info.yaml) are supported");
    if (c->obs_channels < 1 || c->obs_channels > 8) return fail(GRL_ERR_INVALID, "obs_channels out of range");
    if (c->extractor == GRL_EXTRACTOR_AUGMENTED && (c->n_direct < 0 || c->n_direct > 64))
      return fail(GRL_ERR_INVALID, "n_direct out of range");
    if (c->extractor == GRL_EXTRACTOR_AUGMENTED && c->n_direct > 0 && c->obs_channels < 2)
      return fail(GRL_ERR_INVALID, "augmented extractor needs >= 2 observation channels");
  }
  return GRL_OK;
}

extern "C" {

const char* grl_last_error(void) { return g_err.c_str(); }
int grl_version(void) { return 1; }

int grl_query_sizes(const grl_config* cfg, grl_sizes* out) {
  if (int e = check_cfg(cfg)) return e;
  if (!out) return fail(GRL_ERR_INVALID, "null out");
  grl_ctx ctx;
  ctx.cfg = *cfg;
  ctx.dry = true;
  if (int e = ctx.plan()) return e;
  out->state_bytes = ctx.st.off + 256;
  out->grads_bytes = ctx.gr.off + 256;
  out->work_bytes = ctx.wk.off + 256;
  out->replay_bytes = ctx.rp.off + 256;
  out->n_params = ctx.n_params;
  out->n_trainable = ctx.n_train;
  return GRL_OK;
}

int grl_create(const grl_config* cfg, const grl_buffers* bufs, grl_handle* out) {
  if (int e = check_cfg(cfg)) return e;
  if (!bufs || !out || !bufs->state || !bufs->grads || !bufs->work || !bufs->replay)
    return fail(GRL_ERR_INVALID, "null buffers");
  grl_ctx* h = new grl_ctx();
  h->cfg = *cfg;
  h->st.base = (char*)bufs->state; h->gr.base = (char*)bufs->grads;
  h->wk.base = (char*)bufs->work; h->rp.base = (char*)bufs->replay;
  if (int e = h->plan()) { delete h; return e; }
  const char* ng = getenv("GRL_NO_GRAPH");
  h->use_graph = !(ng && ng[0] == '1');
  for (auto& u : h->uploads) {
    hipError_t e = hipMemcpy(u.dst, u.bytes.data(), u.bytes.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete h; return fail(GRL_ERR_HIP, std::string("table upload: ") + hipGetErrorString(e)); }
  }
  h->uploads.clear();
  for (auto& z : h->zero_once) hipMemset(z.first, 0, z.second);
  // state: zero Adam moments, scalars; stats = identity
  hipMemset(h->adam_m, 0, (size_t)h->n_train * 4);
  hipMemset(h->adam_v, 0, (size_t)h->n_train * 4);
  hipMemset(h->grads, 0, (size_t)h->n_train * 4);
  DevScalars s0;
  memset(&s0, 0, sizeof(s0));
  s0.beta1_power = 0.9f; s0.beta2_power = 0.999f;
  s0.lr = cfg->lr;
  hipError_t e = hipMemcpy(h->sc, &s0, sizeof(s0), hipMemcpyHostToDevice);
  if (e != hipSuccess) { delete h; return fail(GRL_ERR_HIP, std::string("scalar init: ") + hipGetErrorString(e)); }
  if (h->n_mean) {   // RunningMeanStd(): mean 0, var 1, count 1e-4
    std::vector<double> ones((size_t)h->n_elems, 1.0);
    const double c2[2] = {1e-4, 1e-4};
    hipMemset(h->n_mean, 0, (size_t)h->n_elems * 8);
    hipMemcpy(h->n_var, ones.data(), ones.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(h->n_count, c2, 16, hipMemcpyHostToDevice);
  }
  if (h->per_on) {
    hipMemset(h->per.p, 0, (size_t)cfg->replay_capacity * 8);
    PerState ps;
    memset(&ps, 0, sizeof(ps));
    ps.max_priority = 1.f; ps.p_min = 1.0; ps.beta = 1.0;
    e = hipMemcpy(h->per.st, &ps, sizeof(ps), hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete h; return fail(GRL_ERR_HIP, std::string("per init: ") + hipGetErrorString(e)); }
  }
  *out = h;
  return GRL_OK;
}

int grl_destroy(grl_handle h) {
  if (!h) return GRL_OK;
  hipStreamSynchronize(h->stream);
  delete h;
  return GRL_OK;
}

int grl_set_stream(grl_handle h, void* s) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if ((hipStream_t)s != h->stream) h->drop_graphs();
  h->stream = (hipStream_t)s;
  return GRL_OK;
}

int grl_param_count(grl_handle h) { return h ? (int)h->vars.size() : fail(GRL_ERR_INVALID, "null handle"); }

int grl_param_info(grl_handle h, int i, char* name, int cap, int64_t* off, int64_t* numel, int32_t* ndim,
                   int64_t shape[4], int32_t* trainable) {
  if (!h || i < 0 || i >= (int)h->vars.size()) return fail(GRL_ERR_INVALID, "bad parameter index");
  const Var& v = h->vars[i];
  if (name && cap > 0) { strncpy(name, v.name.c_str(), cap - 1); name[cap - 1] = 0; }
  if (off) *off = v.off;
  if (numel) *numel = v.numel;
  if (ndim) *ndim = v.ndim;
  if (shape) for (int k = 0; k < 4; ++k) shape[k] = v.shape[k];
  if (trainable) *trainable = v.trainable ? 1 : 0;
  return GRL_OK;
}

int grl_reset_optimizer(grl_handle h) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  HIPCHK(hipMemsetAsync(h->adam_m, 0, (size_t)h->n_train * 4, h->stream));
  HIPCHK(hipMemsetAsync(h->adam_v, 0, (size_t)h->n_train * 4, h->stream));
  DevScalars s0;
  HIPCHK(hipMemcpy(&s0, h->sc, sizeof(s0), hipMemcpyDeviceToHost));
  s0.beta1_power = 0.9f; s0.beta2_power = 0.999f;
  HIPCHK(hipMemcpy(h->sc, &s0, sizeof(s0), hipMemcpyHostToDevice));
  return GRL_OK;
}

int grl_set_learning_rate(grl_handle h, float lr) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if (!(lr >= 0.f)) return fail(GRL_ERR_INVALID, "learning rate must be >= 0");
  // the step size lives in device memory (the captured graphs read it), set in stream order
  hipLaunchKernelGGL(set_f32_kernel, dim3(1), dim3(1), 0, h->stream, &h->sc->lr, lr);
  return GRL_OK;
}

int grl_set_obs_stats(grl_handle h, const double* mean, const double* var, double ret_var) {
  if (!h || !mean || !var) return fail(GRL_ERR_INVALID, "null argument");
  const grl_config& c = h->cfg;
  const double eps = c.norm_eps;
  const int nd = h->cnn ? h->F - 512 : 0;
  // The five statistic blocks sit one after the other in the state arena (alignment gaps in between are unused):
  // they are written into a page-locked mirror of that span and leave as ONE asynchronous copy in stream order.
  // Two mirrors alternate, each guarded by an event, so the host never waits for the GPU here (the learn loop calls
  // this before every update: a stream synchronisation plus five blocking copies serialised host and device).
  char* base = (char*)h->s_mean;
  const size_t span = (size_t)((char*)h->s_ret + 8 - base);
  if (!h->pin_stats[0]) {
    for (int k = 0; k < 2; ++k) {
      HIPCHK(hipHostMalloc((void**)&h->pin_stats[k], span, 0));
      memset(h->pin_stats[k], 0, span);
      HIPCHK(hipEventCreateWithFlags(&h->pin_stats_ev[k], hipEventDisableTiming));
    }
  }
  const int k = h->pin_stats_next;
  h->pin_stats_next ^= 1;
  if (h->pin_stats_used[k]) HIPCHK(hipEventSynchronize(h->pin_stats_ev[k]));   // the copy issued two calls ago
  char* pm = h->pin_stats[k];
  double* m = (double*)pm;
  double* s = (double*)(pm + ((char*)h->s_std - base));
  double* dm = (double*)(pm + ((char*)h->s_dmean - base));
  double* ds = (double*)(pm + ((char*)h->s_dstd - base));
  double* rs = (double*)(pm + ((char*)h->s_ret - base));
  if (h->cnn) {
    const int co = c.obs_channels, ci = h->C_img;
    for (int px = 0; px < h->hw * h->hw; ++px)
      for (int ch = 0; ch < ci; ++ch) {
        m[px * ci + ch] = mean[px * co + ch];
        s[px * ci + ch] = std::sqrt(var[px * co + ch] + eps);
      }
    for (int q = 0; q < nd; ++q) {
      dm[q] = mean[q * co + (co - 1)];
      ds[q] = std::sqrt(var[q * co + (co - 1)] + eps);
    }
  } else {
    for (int q = 0; q < h->img_elems; ++q) { m[q] = mean[q]; s[q] = std::sqrt(var[q] + eps); }
  }
  *rs = std::sqrt(ret_var + eps);
  HIPCHK(hipMemcpyAsync(base, pm, span, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipEventRecord(h->pin_stats_ev[k], h->stream));
  h->pin_stats_used[k] = true;
  return GRL_OK;
}

// Host-to-device copy of a CALLER-owned buffer that the caller may reuse the moment the entry point returns (include/grl.h:
// "copied before the call returns").  For PAGEABLE memory (NumPy arrays, malloc) hipMemcpyAsync stages the source before it
// returns -- the documented behaviour of the asynchronous copies for pageable host memory -- so nothing more is needed.  A
// page-locked or registered source is read by the DMA engine LATER, in stream order: wait for that copy.  GRL_TUNE
// host_copy_wait=1 waits in every case (a runtime whose pageable path cannot be trusted); tests/test_gpu_api.py mutates
// the source right after each of these calls.
static int copy_from_caller(grl_handle h, void* dst, const void* src, size_t bytes) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
  static const int always = tune_int("host_copy_wait", 0);
  bool wait = always != 0;
  if (!wait) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, src) == hipSuccess) wait = at.type == hipMemoryTypeHost || at.type == hipMemoryTypeManaged;
    else (void)hipGetLastError();      // unknown to the runtime: pageable
  }
  if (wait) {
    if (!h->copy_ev) HIPCHK(hipEventCreateWithFlags(&h->copy_ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->copy_ev, h->stream));
    HIPCHK(hipEventSynchronize(h->copy_ev));
  }
  return GRL_OK;
}

int grl_set_ret_var(grl_handle h, double ret_var) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  const double sd = std::sqrt(ret_var + (double)h->cfg.norm_eps);
  HIPCHK(hipMemcpyAsync(h->s_ret, &sd, 8, hipMemcpyHostToDevice, h->stream));   // (pageable source: staged before returning)
  return GRL_OK;
}

int grl_set_running_stats(grl_handle h, const double* mean, const double* var, double count) {
  if (!h || !mean || !var) return fail(GRL_ERR_INVALID, "null argument");
  if (!h->n_mean) return fail(GRL_ERR_STATE, "this handle keeps no running statistics");
  // (rare: once per attach / load; pageable sources are staged before hipMemcpyAsync returns)
  if (int e = copy_from_caller(h, h->n_mean, mean, (size_t)h->n_elems * 8)) return e;
  if (int e = copy_from_caller(h, h->n_var, var, (size_t)h->n_elems * 8)) return e;
  const double c2[2] = {count, count};
  HIPCHK(hipMemcpyAsync(h->n_count, c2, 16, hipMemcpyHostToDevice, h->stream));
  return GRL_OK;
}

// RunningMeanStd.update over `n` raw observations that are already on the device
static int norm_update_from(grl_handle h, const float* dev_obs, int n) {
  const grl_config& c = h->cfg;
  NormUpdateArgs a;
  memset(&a, 0, sizeof(a));
  a.obs = dev_obs; a.n = n; a.elems = (int)h->n_elems;
  a.mean = h->n_mean; a.var = h->n_var; a.count = h->n_count; a.parity = h->n_parity; a.eps = c.norm_eps;
  a.hw = h->hw * h->hw; a.c_obs = c.obs_channels; a.c_img = h->C_img; a.n_direct = h->cnn ? h->F - 512 : 0; a.vec = h->cnn ? 0 : 1;
  a.s_mean = h->s_mean; a.s_std = h->s_std; a.s_dmean = h->s_dmean; a.s_dstd = h->s_dstd;
  const dim3 grid((unsigned)((h->n_elems + 255) / 256));
  if (h->dp_on) {   // data parallel: the batch moments of all ranks, merged in rank order by every replica (dp_kernels.h)
    if (h->dp_err_host && *h->dp_err_host) return fail(GRL_ERR_STATE, "an exchange timed out waiting for a peer (the replicas are no longer in step)");
    DpNormArgs na = h->dp_norm;
    na.nu = a;
    hipLaunchKernelGGL(dp_norm_moments_kernel, grid, dim3(256), 0, h->stream, na);
    hipLaunchKernelGGL(dp_wait_kernel, dim3(1), dim3(64), 0, h->stream, na.d, 0);
    hipLaunchKernelGGL(dp_norm_merge_kernel, grid, dim3(256), 0, h->stream, na);
  } else {
    hipLaunchKernelGGL(norm_update_kernel, grid, dim3(256), 0, h->stream, a);
  }
  h->n_parity ^= 1;
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_norm_update(grl_handle h, const float* obs, int n) {
  if (!h || !obs || n < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (!h->n_mean) return fail(GRL_ERR_STATE, "this handle keeps no running statistics");
  if (n > h->stg_n) return fail(GRL_ERR_INVALID, "more observations than one env step of act_batch environments");
  if (int e = copy_from_caller(h, h->n_stage, obs, (size_t)n * h->n_elems * 4)) return e;
  return norm_update_from(h, h->n_stage, n);
}

int grl_get_obs_stats(grl_handle h, double* mean, double* var, double* count) {
  if (!h || !mean || !var || !count) return fail(GRL_ERR_INVALID, "null argument");
  if (!h->n_mean) return fail(GRL_ERR_STATE, "this handle keeps no running statistics");
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(mean, h->n_mean, (size_t)h->n_elems * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(var, h->n_var, (size_t)h->n_elems * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(count, h->n_count + h->n_parity, 8, hipMemcpyDeviceToHost));
  return GRL_OK;
}

static int replay_add_dev(grl_handle h, const float* obs, const float* act, const float* rew, const float* nxt,
                          const float* done, int n, const int* next_row = nullptr, const float* next_alt = nullptr) {
  const grl_config& c = h->cfg;
  IngestArgs ia;
  memset(&ia, 0, sizeof(ia));
  ia.obs = obs; ia.next_obs = nxt; ia.act = act; ia.rew = rew; ia.done = done;
  ia.n = n; ia.hw = h->hw * h->hw; ia.c_obs = c.obs_channels; ia.c_img = h->C_img;
  ia.n_direct = h->cnn ? h->F - 512 : 0; ia.act_dim = h->A; ia.vec_dim = h->cnn ? 0 : c.obs_dim;
  ia.pos = h->rp_pos; ia.cap = c.replay_capacity;
  ia.rp_obs = h->rp_obs; ia.rp_next = h->rp_next; ia.rp_dobs = h->rp_dobs; ia.rp_dnext = h->rp_dnext;
  ia.rp_act = h->rp_act; ia.rp_rew = h->rp_rew; ia.rp_done = h->rp_done;
  ia.rgb_u8 = (c.algo == GRL_ALGO_SAC) ? c.replay_rgb_u8 : 0;
  ia.next_row = next_row; ia.next_alt = next_alt;
  const int elems = h->cnn ? (ia.rgb_u8 ? h->hw * h->hw : h->img_elems) : c.obs_dim;
  ia.size_out = &h->sc->replay_size;
  ia.new_size = std::min<int64_t>(c.replay_capacity, h->rp_size + n);
  hipLaunchKernelGGL(ingest_kernel, dim3((elems + 255) / 256, n, 2), dim3(256), 0, h->stream, ia);
  if (h->per_on)   // new transitions enter with max_priority ** alpha
    hipLaunchKernelGGL(per_add_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->per, h->rp_pos, n,
                       (int64_t)c.replay_capacity);
  h->rp_pos = (h->rp_pos + n) % c.replay_capacity;
  h->rp_size = ia.new_size;
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_replay_add_device(grl_handle h, const float* obs, const float* act, const float* rew,
                          const float* next_obs, const float* done, int n) {
  if (!h || !obs || !act || !rew || !next_obs || !done || n < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (n > h->cfg.replay_capacity) return fail(GRL_ERR_INVALID, "n exceeds replay capacity");
  return replay_add_dev(h, obs, act, rew, next_obs, done, n);
}

// host buffer -> pinned staging -> device (and back): hipMemcpyAsync from pageable memory is a synchronous
// staged copy; through page-locked buffers the per-call overhead is a host memcpy plus a true async DMA
static int pin_reserve(grl_handle h, size_t in_floats, size_t out_floats) {
  if (in_floats > h->pin_in_n) {
    if (h->pin_in) hipHostFree(h->pin_in);
    h->pin_in = nullptr; h->pin_in_n = 0;
    HIPCHK(hipHostMalloc((void**)&h->pin_in, in_floats * 4, 0));
    h->pin_in_n = in_floats;
  }
  if (out_floats > h->pin_out_n) {
    if (h->pin_out) hipHostFree(h->pin_out);
    h->pin_out = nullptr; h->pin_out_n = 0;
    HIPCHK(hipHostMalloc((void**)&h->pin_out, out_floats * 4, 0));
    h->pin_out_n = out_floats;
  }
  return GRL_OK;
}

int grl_replay_add(grl_handle h, const float* obs, const float* act, const float* rew, const float* next_obs,
                   const float* done, int n) {
  if (!h || !obs || !act || !rew || !next_obs || !done || n < 1) return fail(GRL_ERR_INVALID, "bad argument");
  const int64_t oe = h->cnn ? (int64_t)h->hw * h->hw * h->cfg.obs_channels : h->cfg.obs_dim;
  // (pageable copies: for these sizes -- 64 KB per transition -- an extra host copy into pinned staging costs
  // more than it saves: 77 -> 90 us for 16 transitions, 150 -> 214 us for 64; measured with scripts/act_bench.py)
  for (int k0 = 0; k0 < n; k0 += h->stg_n) {
    const int m = std::min(h->stg_n, n - k0);
    if (int e = copy_from_caller(h, h->stg_obs, obs + k0 * oe, (size_t)m * oe * 4)) return e;
    if (int e = copy_from_caller(h, h->stg_next, next_obs + k0 * oe, (size_t)m * oe * 4)) return e;
    if (int e = copy_from_caller(h, h->stg_act, act + (int64_t)k0 * h->A, (size_t)m * h->A * 4)) return e;
    if (int e = copy_from_caller(h, h->stg_rew, rew + k0, (size_t)m * 4)) return e;
    if (int e = copy_from_caller(h, h->stg_done, done + k0, (size_t)m * 4)) return e;
    // (no wait between chunks for pageable sources: the copies are stream-ordered behind the ingest launch that reads the
    // staging buffers, and the runtime has taken its copy of the source by the time copy_from_caller returns)
    if (int e = replay_add_dev(h, h->stg_obs, h->stg_act, h->stg_rew, h->stg_next, h->stg_done, m)) return e;
  }
  return GRL_OK;
}

// ---------------------------------------------------------------------------------------------- observations uploaded once
// One env step's observations serve three consumers -- the running statistics, the next action and two replay rows
// (next_obs of this step, obs of the next).  grl_observe uploads them ONCE; grl_act (GRL_ACT_OBSERVED) and
// grl_replay_add_observed read the device copy.  The small per-step arrays of the latter (actions, rewards, done flags,
// terminal rows) pass through page-locked staging that the ingest launch reads directly: two buffers used in turn, each
// guarded by an event.
static int ob_stage_acquire(grl_handle h, float** pin) {
  const size_t need = (size_t)h->stg_n * (size_t)(h->ob_elems + h->A + 4);
  if (h->pin_ob_n < need) {
    for (int k = 0; k < 2; ++k) {
      if (h->pin_ob[k]) { HIPCHK(hipStreamSynchronize(h->stream)); hipHostFree(h->pin_ob[k]); h->pin_ob[k] = nullptr; }
      HIPCHK(hipHostMalloc((void**)&h->pin_ob[k], need * 4, 0));
      if (!h->pin_ob_ev[k]) HIPCHK(hipEventCreateWithFlags(&h->pin_ob_ev[k], hipEventDisableTiming));
      h->pin_ob_used[k] = false;
    }
    h->pin_ob_n = need;
  }
  const int k = h->pin_ob_next;
  if (h->pin_ob_used[k]) HIPCHK(hipEventSynchronize(h->pin_ob_ev[k]));   // the copies that read it two calls ago
  *pin = h->pin_ob[k];
  return GRL_OK;
}
static int ob_stage_release(grl_handle h) {
  const int k = h->pin_ob_next;
  HIPCHK(hipEventRecord(h->pin_ob_ev[k], h->stream));
  h->pin_ob_used[k] = true;
  h->pin_ob_next = k ^ 1;
  return GRL_OK;
}

int grl_observe(grl_handle h, const float* obs, int n, int flags) {
  if (!h || !obs || n < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (!h->ob_latest) return fail(GRL_ERR_STATE, "this handle keeps no observed observations (SAC handles do)");
  if (n > h->stg_n) return fail(GRL_ERR_INVALID, "more observations than one env step of act_batch environments");
  if (flags & ~GRL_OBSERVE_UPDATE_STATS) return fail(GRL_ERR_INVALID, "unknown flag bits");
  if ((flags & GRL_OBSERVE_UPDATE_STATS) && !h->n_mean) return fail(GRL_ERR_STATE, "this handle keeps no running statistics");
  const size_t bytes = (size_t)n * h->ob_elems * 4;
  if (h->ob_n > 0)   // the previous step's observations become the `obs` side of the next replay rows
    HIPCHK(hipMemcpyAsync(h->ob_prev, h->ob_latest, (size_t)h->ob_n * h->ob_elems * 4, hipMemcpyDeviceToDevice, h->stream));
  h->ob_n_prev = h->ob_n;
  // (the caller's buffer goes to hipMemcpyAsync as it is, copy_from_caller; a bounce through page-locked staging of our own
  //  measured slower: 140 vs 126 us per env step of 16 observations, round 4)
  if (int e = copy_from_caller(h, h->ob_latest, obs, bytes)) return e;
  h->ob_n = n;
  if (flags & GRL_OBSERVE_UPDATE_STATS) return norm_update_from(h, h->ob_latest, n);
  return GRL_OK;
}

int grl_replay_add_observed(grl_handle h, const float* act, const float* rew, const float* done, int n,
                            const int32_t* term_rows, const float* term_obs, int n_term) {
  if (!h || !act || !rew || !done || n < 1 || n_term < 0 || n_term > n) return fail(GRL_ERR_INVALID, "bad argument");
  if (n_term > 0 && (!term_rows || !term_obs)) return fail(GRL_ERR_INVALID, "terminal rows without their observations");
  if (!h->ob_latest) return fail(GRL_ERR_STATE, "this handle keeps no observed observations (SAC handles do)");
  if (h->ob_n != n || h->ob_n_prev != n)
    return fail(GRL_ERR_STATE, "grl_replay_add_observed needs two consecutive grl_observe calls of n observations each");
  float* pin = nullptr;
  if (int e = ob_stage_acquire(h, &pin)) return e;
  const int A = h->A;
  memcpy(pin, act, (size_t)n * A * 4);
  memcpy(pin + (size_t)n * A, rew, (size_t)n * 4);
  memcpy(pin + (size_t)n * (A + 1), done, (size_t)n * 4);
  int32_t* rowmap = (int32_t*)(pin + (size_t)n * (A + 2));
  for (int k = 0; k < n; ++k) rowmap[k] = -1;
  for (int j = 0; j < n_term; ++j) {
    if (term_rows[j] < 0 || term_rows[j] >= n) { (void)ob_stage_release(h); return fail(GRL_ERR_INVALID, "terminal row out of range"); }
    rowmap[term_rows[j]] = j;
  }
  // (the ingest launch reads actions / rewards / done flags / row map straight from the page-locked buffer: 0.5 KB)
  const size_t small = (size_t)n * (A + 3);
  if (n_term > 0) {
    memcpy(pin + small, term_obs, (size_t)n_term * h->ob_elems * 4);
    HIPCHK(hipMemcpyAsync(h->ob_term, pin + small, (size_t)n_term * h->ob_elems * 4, hipMemcpyHostToDevice, h->stream));
  }
  const int e = replay_add_dev(h, h->ob_prev, pin, pin + (size_t)n * A, h->ob_latest, pin + (size_t)n * (A + 1), n,
                               n_term > 0 ? (const int*)(pin + (size_t)n * (A + 2)) : nullptr, h->ob_term);
  if (int e2 = ob_stage_release(h)) return e2;    // (the event follows the launch that reads the buffer)
  return e;
}

int64_t grl_replay_size(grl_handle h) { return h ? h->rp_size : -1; }

static int stage_noise(grl_handle h, const int64_t* idx, const float* eps, int step) {
  HIPCHK(hipMemcpyAsync(h->idx_buf, idx + (int64_t)step * h->B, (size_t)h->B * 8, hipMemcpyDeviceToDevice, h->stream));
  const int64_t per = h->cfg.algo == GRL_ALGO_SAC ? (int64_t)h->B * h->A : (int64_t)h->B;   // Q: importance weights
  HIPCHK(hipMemcpyAsync(h->eps_buf, eps + (int64_t)step * per, (size_t)per * 4, hipMemcpyDeviceToDevice, h->stream));
  return GRL_OK;
}

int grl_compute_grads(grl_handle h, const int64_t* idx, const float* eps) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if ((idx == nullptr) != (eps == nullptr)) return fail(GRL_ERR_INVALID, "idx and eps must both be given or both be NULL");
  if (h->rp_size < 1) return fail(GRL_ERR_STATE, "replay buffer is empty");
  if (idx) {
    if (int e = stage_noise(h, idx, eps, 0)) return e;
    if (int e = h->run_seq("grads_explicit", {&h->ops_gather, &h->ops_grads})) return e;
  } else {
    if (int e = h->run_seq("grads_rng", {&h->ops_rng, &h->ops_grads})) return e;
  }
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_compute_grads_staged(grl_handle h, int stage, const int64_t* idx, const float* eps) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if (stage != 0 && stage != 1) return fail(GRL_ERR_INVALID, "stage must be 0 or 1");
  if (!h->staged_ok) {          // plans without a staged form: everything in stage 0, one bucket (grl_grad_ranges)
    if (stage == 1) return GRL_OK;
    return grl_compute_grads(h, idx, eps);
  }
  if (stage == 1) {
    if (int e = h->run_seq("grads_stage1", {&h->ops_stage1})) return e;
    HIPCHK(hipGetLastError());
    return GRL_OK;
  }
  if ((idx == nullptr) != (eps == nullptr)) return fail(GRL_ERR_INVALID, "idx and eps must both be given or both be NULL");
  if (h->rp_size < 1) return fail(GRL_ERR_STATE, "replay buffer is empty");
  if (idx) {
    if (int e = stage_noise(h, idx, eps, 0)) return e;
    if (int e = h->run_seq("grads_stage0_explicit", {&h->ops_gather, &h->ops_stage0})) return e;
  } else {
    if (int e = h->run_seq("grads_stage0_rng", {&h->ops_rng, &h->ops_stage0})) return e;
  }
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_grad_ranges(grl_handle h, int bucket, int cap, int64_t* offsets, int64_t* numels) {
  if (!h || !offsets || !numels) return fail(GRL_ERR_INVALID, "null argument");
  if (bucket != 0 && bucket != 1) return fail(GRL_ERR_INVALID, "bucket must be 0 or 1");
  std::vector<std::pair<int64_t, int64_t>> r;
  if (!h->staged_ok) {
    if (bucket == 0) r.push_back({0, h->n_train});
  } else if (bucket == 0) {      // dense: fc + heads of the policy net, fc + vf / qf1 / qf2 heads of the value net
    r.push_back({h->ex[0].fw, h->vf_off - h->ex[0].fw});
    r.push_back({h->ex[1].fw, h->ent_off - h->ex[1].fw});
  } else {                       // convolutions of both nets + the entropy coefficient (its gradient comes with the loss sums)
    r.push_back({h->ex[0].w[0], h->ex[0].fw - h->ex[0].w[0]});
    r.push_back({h->ex[1].w[0], h->ex[1].fw - h->ex[1].w[0]});
    r.push_back({h->ent_off, h->n_train - h->ent_off});
  }
  if ((int)r.size() > cap) return fail(GRL_ERR_INVALID, "range buffer too small");
  for (size_t k = 0; k < r.size(); ++k) { offsets[k] = r[k].first; numels[k] = r[k].second; }
  return (int)r.size();
}

int grl_apply_grads(grl_handle h, float grad_scale) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if (grad_scale != h->apply_graph_scale) {   // the scale is baked into the captured kernel arguments
    auto it = h->graphs.find("apply");
    if (it != h->graphs.end()) { (void)hipGraphExecDestroy(it->second); h->graphs.erase(it); }
    h->apply_graph_scale = grad_scale;
  }
  h->grad_scale = grad_scale;
  if (int e = h->run_seq("apply", {&h->ops_apply})) return e;
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_train_step(grl_handle h, int n_steps, const int64_t* idx, const float* eps) {
  if (!h || n_steps < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (h->cfg.algo == GRL_ALGO_AE) return fail(GRL_ERR_STATE, "auto-encoder handles train with grl_ae_train_step");
  if ((idx == nullptr) != (eps == nullptr)) return fail(GRL_ERR_INVALID, "idx and eps must both be given or both be NULL");
  if (h->rp_size < 1) return fail(GRL_ERR_STATE, "replay buffer is empty");
  h->grad_scale = 1.f;
  if (!idx && n_steps >= 2 && h->ride_ok && !h->prof) {     // device RNG, several updates, double-buffered images (plan_sac "gather_ride")
    if (int e = h->run_ride("ride", &h->ops_ride_first, h->ops_ride_mid, h->ops_ride_last, n_steps)) return e;
    HIPCHK(hipGetLastError());
    return GRL_OK;
  }
  if (!idx && n_steps >= 2 && h->prefetch_ok && !h->prof) {   // device RNG, several updates: prefetching sequences (plan_sac)
    // short calls (what SAC.learn issues: n = number of environments) are ONE graph, cached per n
    if (n_steps <= 32 && tune_int("graph_updates", 16) != 1) {
      std::vector<std::vector<Op>*> seq;
      seq.push_back(&h->ops_pf_first);
      for (int s = 0; s < n_steps - 2; ++s) seq.push_back(&h->ops_pf_mid);
      seq.push_back(&h->ops_pf_last);
      if (int e = h->run_seq("pf_call_" + std::to_string(n_steps), seq)) return e;
      HIPCHK(hipGetLastError());
      return GRL_OK;
    }
    // first | middle updates, grouped several to a graph (run_repeated) | last
    if (int e = h->run_seq("pf_first", {&h->ops_pf_first})) return e;
    if (n_steps > 2)
      if (int e = h->run_repeated("pf_mid", {&h->ops_pf_mid}, n_steps - 2)) return e;
    if (int e = h->run_seq("pf_last", {&h->ops_pf_last})) return e;
    HIPCHK(hipGetLastError());
    return GRL_OK;
  }
  if (!idx && n_steps >= 2 && h->q_pf_ok && !h->prof) {      // DQN / BDQ, uniform replay: prefetching sequences (plan_q "q_pf")
    if (n_steps <= 32 && tune_int("graph_updates", 16) != 1) {
      std::vector<std::vector<Op>*> seq;
      seq.push_back(&h->ops_q_pf_first);
      for (int s = 0; s < n_steps - 2; ++s) seq.push_back(&h->ops_q_pf_mid);
      seq.push_back(&h->ops_q_pf_last);
      if (int e = h->run_seq("q_pf_call_" + std::to_string(n_steps), seq)) return e;
      HIPCHK(hipGetLastError());
      return GRL_OK;
    }
    if (int e = h->run_seq("q_pf_first", {&h->ops_q_pf_first})) return e;
    if (n_steps > 2)
      if (int e = h->run_repeated("q_pf_mid", {&h->ops_q_pf_mid}, n_steps - 2)) return e;
    if (int e = h->run_seq("q_pf_last", {&h->ops_q_pf_last})) return e;
    HIPCHK(hipGetLastError());
    return GRL_OK;
  }
  if (!idx) {      // device RNG: identical updates, several to a graph
    if (!h->ops_grads_apply.empty()) {
      if (int e = h->run_repeated("full_rng", {&h->ops_rng, &h->ops_grads_apply}, n_steps)) return e;
    } else if (int e = h->run_repeated("full_rng", {&h->ops_rng, &h->ops_grads, &h->ops_apply}, n_steps)) return e;
    HIPCHK(hipGetLastError());
    return GRL_OK;
  }
  for (int s = 0; s < n_steps; ++s) {
    if (int e = stage_noise(h, idx, eps, s)) return e;
    if (!h->ops_grads_apply.empty()) {
      if (int e = h->run_seq("full_explicit", {&h->ops_gather, &h->ops_grads_apply})) return e;
    } else if (int e = h->run_seq("full_explicit", {&h->ops_gather, &h->ops_grads, &h->ops_apply})) return e;
  }
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_train_step_per(grl_handle h, int n_steps, double beta, const double* u) {
  if (!h || n_steps < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (!h->per_on) return fail(GRL_ERR_STATE, "prioritised replay is not enabled (grl_config.q_per)");
  if (h->rp_size < 1) return fail(GRL_ERR_STATE, "replay buffer is empty");
  if (!(beta > 0.0)) return fail(GRL_ERR_INVALID, "beta must be positive (PrioritizedReplayBuffer.sample asserts beta > 0)");
  if (h->rp_size < 2) return fail(GRL_ERR_STATE, "prioritised sampling needs at least two stored transitions (sum(0, len - 1))");
  HIPCHK(hipMemcpyAsync(&h->per.st->beta, &beta, 8, hipMemcpyHostToDevice, h->stream));
  h->grad_scale = 1.f;
  if (h->dp_on) {
    // connected handle (grl_allreduce_connect): every rank draws from ITS priority tree (importance weights against its own
    // total and minimum), the gradient sums are exchanged inside the graph, the plan's clip + Adam apply the mean, every rank
    // writes the priorities of its own rows back.  All ranks must call alike, as with grl_train_step_allreduce.
    if (*h->dp_err_host) return fail(GRL_ERR_STATE, "an exchange timed out waiting for a peer (the replicas are no longer in step)");
    h->grad_scale = 1.f / (float)h->dp.world;
    if (!u) {
      if (int e = h->run_repeated("dp_per_rng", {&h->ops_per_rng_g, &h->dp_body_per, &h->ops_dp, &h->ops_per_update}, n_steps)) return e;
    } else {
      for (int s = 0; s < n_steps; ++s) {
        HIPCHK(hipMemcpyAsync(h->per_u, u + (int64_t)s * h->B, (size_t)h->B * 8, hipMemcpyDeviceToDevice, h->stream));
        if (int e = h->run_seq("dp_per_u", {&h->ops_per_u_g, &h->dp_body_per, &h->ops_dp, &h->ops_per_update})) return e;
      }
    }
    HIPCHK(hipGetLastError());
    return GRL_OK;
  }
  if (!u) {        // device Philox: identical updates, several to a graph
    if (n_steps >= 2 && h->per_pf_ok && !h->prof) {
      // four launches per update (plan_q "per_pf"): the sampler of update t + 1 rides on the launch that ends update t
      if (int e = h->run_seq("per_pf_first", {&h->ops_per_pf_first})) return e;
      if (n_steps > 2)
        if (int e = h->run_repeated("per_pf_mid", {&h->ops_per_pf_mid}, n_steps - 2)) return e;
      if (int e = h->run_seq("per_pf_last", {&h->ops_per_pf_last})) return e;
    } else if (n_steps >= 2 && !h->ops_grads_apply_per_r.empty() && !h->prof) {
      // nothing but the updates themselves touches the leaves inside one call: the first update sums every block of the
      // ring, each apply launch refreshes the blocks its write-back touched, the later samplers start from those
      if (int e = h->run_seq("per_rng_first", {&h->ops_per_rng_g, &h->ops_grads_apply_per_r})) return e;
      if (int e = h->run_repeated("per_rng_inc", {&h->ops_per_rng_g_inc, &h->ops_grads_apply_per_r}, n_steps - 1)) return e;
    } else if (!h->ops_grads_apply_per.empty()) {
      if (int e = h->run_repeated("per_rng", {&h->ops_per_rng_g, &h->ops_grads_apply_per}, n_steps)) return e;
    } else if (!h->ops_grads_apply.empty()) {
      if (int e = h->run_repeated("per_rng", {&h->ops_per_rng, &h->ops_gather, &h->ops_grads_apply, &h->ops_per_update}, n_steps)) return e;
    } else if (int e = h->run_repeated("per_rng", {&h->ops_per_rng, &h->ops_gather, &h->ops_grads, &h->ops_apply, &h->ops_per_update}, n_steps)) return e;
    HIPCHK(hipGetLastError());
    return GRL_OK;
  }
  for (int s = 0; s < n_steps; ++s) {
    if (u) {
      HIPCHK(hipMemcpyAsync(h->per_u, u + (int64_t)s * h->B, (size_t)h->B * 8, hipMemcpyDeviceToDevice, h->stream));
      if (!h->ops_grads_apply_per.empty()) {
        if (int e = h->run_seq("per_u", {&h->ops_per_u_g, &h->ops_grads_apply_per})) return e;
      } else if (!h->ops_grads_apply.empty()) {
        if (int e = h->run_seq("per_u", {&h->ops_per_u, &h->ops_gather, &h->ops_grads_apply, &h->ops_per_update})) return e;
      } else if (int e = h->run_seq("per_u", {&h->ops_per_u, &h->ops_gather, &h->ops_grads, &h->ops_apply, &h->ops_per_update})) return e;
    } else {
      if (!h->ops_grads_apply_per.empty()) {
        if (int e = h->run_seq("per_rng", {&h->ops_per_rng_g, &h->ops_grads_apply_per})) return e;
      } else if (!h->ops_grads_apply.empty()) {
        if (int e = h->run_seq("per_rng", {&h->ops_per_rng, &h->ops_gather, &h->ops_grads_apply, &h->ops_per_update})) return e;
      } else if (int e = h->run_seq("per_rng", {&h->ops_per_rng, &h->ops_gather, &h->ops_grads, &h->ops_apply, &h->ops_per_update})) return e;
    }
  }
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_ae_train_step(grl_handle h, const float* imgs, int n_steps) {
  if (!h || !imgs || n_steps < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (h->cfg.algo != GRL_ALGO_AE) return fail(GRL_ERR_STATE, "not an auto-encoder handle (grl_config.algo)");
  const size_t per = (size_t)h->B * 4096;
  for (int s = 0; s < n_steps; ++s) {
    HIPCHK(hipMemcpyAsync(h->ae_x, imgs + per * s, per * 4, hipMemcpyDeviceToDevice, h->stream));
    if (int e = h->run_seq("ae_step", {&h->ops_ae})) return e;
  }
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_ae_reconstruct(grl_handle h, const float* imgs, float* out) {
  if (!h || !imgs || !out) return fail(GRL_ERR_INVALID, "null argument");
  if (h->cfg.algo != GRL_ALGO_AE) return fail(GRL_ERR_STATE, "not an auto-encoder handle (grl_config.algo)");
  const size_t per = (size_t)h->B * 4096;
  HIPCHK(hipMemcpyAsync(h->ae_x, imgs, per * 4, hipMemcpyDeviceToDevice, h->stream));
  if (int e = h->run_seq("ae_fwd", {&h->ops_ae_fwd})) return e;
  HIPCHK(hipMemcpyAsync(out, h->ae_out, per * 4, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_get_metrics(grl_handle h, grl_metrics* out) {
  if (!h || !out) return fail(GRL_ERR_INVALID, "null argument");
  DevScalars s;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(&s, h->sc, sizeof(s), hipMemcpyDeviceToHost));
  out->policy_loss = s.policy_loss; out->qf1_loss = s.qf1_loss; out->qf2_loss = s.qf2_loss;
  out->value_loss = s.value_loss; out->ent_coef_loss = s.ent_loss; out->ent_coef = s.ent_coef;
  out->entropy = s.entropy; out->mean_qf1 = s.mean_qf1; out->mean_v = s.mean_v;
  return GRL_OK;
}

int grl_act(grl_handle h, const float* obs, int n, int flags, const float* eps, float* out) {
  if (!h || !out || n < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (flags & ~(GRL_ACT_DETERMINISTIC | GRL_ACT_RAW_OBS | GRL_ACT_OBSERVED)) return fail(GRL_ERR_INVALID, "unknown flag bits");
  const int deterministic = flags & GRL_ACT_DETERMINISTIC;
  const bool raw = (flags & GRL_ACT_RAW_OBS) != 0;        // raw observations: VecNormalize applied on the device (grl_norm_update statistics)
  const bool observed = (flags & GRL_ACT_OBSERVED) != 0;  // act on what grl_observe uploaded: no second upload
  const bool q = h->cfg.algo != GRL_ALGO_SAC;   // DQN / BDQ: Q-values [n, D*bins]
  if (!observed && !obs) return fail(GRL_ERR_INVALID, "bad argument");
  if (q && (raw || observed)) return fail(GRL_ERR_STATE, "the Q handles take normalised observations handed to grl_act");
  if (raw && !h->n_mean) return fail(GRL_ERR_STATE, "this handle has no normalising act path");
  if (observed && h->ob_n != n) return fail(GRL_ERR_STATE, "n differs from the number of observations grl_observe holds");
  if (n > h->NA) return fail(GRL_ERR_INVALID, "n exceeds act_batch");
  if (!q && !deterministic && !eps) return fail(GRL_ERR_INVALID, "stochastic action needs eps");
  const int64_t oe = (!q && h->cnn) ? (int64_t)h->hw * h->hw * h->cfg.obs_channels : h->cfg.obs_dim;
  const size_t n_in = observed ? 0 : (size_t)n * oe, n_eps = (!q && !deterministic) ? (size_t)n * h->A : 0;
  const size_t n_out = q ? (size_t)n * h->qD * h->qN : (size_t)n * h->A;
  if (int e = pin_reserve(h, n_in, q ? n_out : 0)) return e;
  if (n_in) memcpy(h->pin_in, obs, n_in * 4);
  if (n_eps) memcpy(h->a_eps, eps, n_eps * 4);      // (SAC: a_eps / a_out are host memory the launches read and write; the
  if (n_in) HIPCHK(hipMemcpyAsync(h->stg_obs, h->pin_in, n_in * 4, hipMemcpyHostToDevice, h->stream));   // previous call synchronised)
  if (q) {
    if (int e = h->run_seq("act", {&h->ops_act})) return e;
  } else {
    // the launches cover act_batch rows whatever n is (rows beyond n hold stale observations: computed, not returned)
    const int v = (observed ? 2 : 0) | (raw ? 1 : 0);
    if (int e = h->run_seq(std::string(deterministic ? "act_det" : "act_sto") + char('0' + v),
                           {&h->ops_act_in[v], &h->ops_act, deterministic ? &h->ops_act_det : &h->ops_act_sto}))
      return e;
  }
  const bool poll = h->act_done_wgs && !h->prof;
  if (q && !poll) HIPCHK(hipMemcpyAsync(h->pin_out, h->q_aout, n_out * 4, hipMemcpyDeviceToHost, h->stream));
  bool seen = false;
  // the last launch counts its workgroups into coherent host memory once their actions are out (act_mfma.h) -- on EVERY call,
  // polled or not (a call under grl_profile_enable synchronises the stream instead): the expectation advances with it
  if (h->act_done_wgs) h->act_done_seen += h->act_done_wgs;
  if (poll) {
    // poll the counter instead of the stream's completion signal; bounded -- after 2 ms (or on any doubt) the stream is
    // synchronised as before.  Wrap-safe and tolerant of a counter that is ahead: (int)(cur - want) >= 0
    const unsigned want = h->act_done_seen;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
      if ((int)(__atomic_load_n(h->act_done_host, __ATOMIC_ACQUIRE) - want) >= 0) { seen = true; break; }
      if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
  }
  if (!seen) {
    HIPCHK(hipStreamSynchronize(h->stream));
    // the stream is idle: whatever the counter says now is what the next call starts from
    if (h->act_done_wgs) h->act_done_seen = __atomic_load_n(h->act_done_host, __ATOMIC_ACQUIRE);
  }
  HIPCHK(hipGetLastError());
  memcpy(out, q ? (poll ? h->q_act_host : h->pin_out) : h->a_out, n_out * 4);
  return GRL_OK;
}

// ---------------------------------------------------------------------------------------------- data parallel, in-graph
// this rank's two exchange allocations: flags = [DP_CHANNELS x DpCtl] (fine-grained), data = [src | red | gathered | 2 moment blocks]
static size_t dp_ctl_stride() { return (size_t)rup((int64_t)sizeof(DpCtl), 256); }
static size_t dp_arr_bytes(int64_t n) { return (size_t)rup(n * 4, 256); }
static size_t dp_mom_bytes(int64_t elems) { return (size_t)rup((2 * elems + 4) * 4, 256); }

int grl_allreduce_init(grl_handle h, int rank, int world, void* handle_out) {
  if (!h || !handle_out) return fail(GRL_ERR_INVALID, "null argument");
  if (h->cfg.algo == GRL_ALGO_AE) return fail(GRL_ERR_STATE, "the exchange step is defined for SAC / DQN / BDQ handles");
  if (world < 1 || world > DP_MAX_WORLD || rank < 0 || rank >= world) return fail(GRL_ERR_INVALID, "bad rank / world size");
  if (h->dp_buf) return fail(GRL_ERR_STATE, "grl_allreduce_init was already called on this handle");
  if (h->n_train % 4 || h->n_train * 4 >= (int64_t)1 << 31) return fail(GRL_ERR_STATE, "the gradient bucket is not a whole number of 16-byte groups below 2 GB");
  static_assert(sizeof(hipIpcMemHandle_t) == 64 && GRL_ALLREDUCE_HANDLE_BYTES == 128, "grl.h documents two 64-byte handles");
  const size_t fbytes = DP_CHANNELS * dp_ctl_stride();
  const size_t dbytes = 3 * dp_arr_bytes(h->n_train) + 2 * dp_mom_bytes(h->n_elems);
  // flags fine-grained: stores of a peer become visible to a kernel that is already running (and polling)
  hipError_t e = hipExtMallocWithFlags(&h->dp_flags, fbytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { h->dp_flags = nullptr; return fail(GRL_ERR_HIP, std::string("exchange flags: ") + hipGetErrorString(e)); }
  // the data as well unless GRL_TUNE dp_coarse=1: the exchange kernels store it write-through and load it at system scope,
  // so its caching policy costs nothing measurable on one GPU and fine-grained is the conservative choice between GPUs
  e = tune_int("dp_coarse", 0) ? hipMalloc(&h->dp_buf, dbytes) : hipExtMallocWithFlags(&h->dp_buf, dbytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipFree(h->dp_flags); h->dp_flags = nullptr; h->dp_buf = nullptr;
    return fail(GRL_ERR_HIP, std::string("exchange buffer: ") + hipGetErrorString(e));
  }
  HIPCHK(hipMemset(h->dp_flags, 0, fbytes));
  HIPCHK(hipMemset(h->dp_buf, 0, dbytes));
  for (int k = 0; k < DP_CHANNELS; ++k) {     // the first exchange (epoch 1) uses source buffer 1
    const uint32_t one = 1u;
    HIPCHK(hipMemcpy((char*)h->dp_flags + k * dp_ctl_stride() + offsetof(DpCtl, next_buf), &one, 4, hipMemcpyHostToDevice));
  }
  if (!h->dp_err_host) {     // page-locked mailbox: a kernel that gives up waiting tells the host without a device read
    HIPCHK(hipHostMalloc((void**)&h->dp_err_host, 64, 0));
    memset(h->dp_err_host, 0, 64);      // word 0: error, word 1: wait bound in ms (0 = the captured default; grl_allreduce_set_timeout)
  }
  HIPCHK(hipDeviceSynchronize());
  hipIpcMemHandle_t mh[2];
  e = hipIpcGetMemHandle(&mh[0], h->dp_flags);
  if (e == hipSuccess) e = hipIpcGetMemHandle(&mh[1], h->dp_buf);
  if (e != hipSuccess) {
    (void)hipFree(h->dp_buf); (void)hipFree(h->dp_flags); h->dp_buf = h->dp_flags = nullptr;
    return fail(GRL_ERR_HIP, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e));
  }
  memcpy(handle_out, mh, GRL_ALLREDUCE_HANDLE_BYTES);
  memset(&h->dp, 0, sizeof(h->dp));
  h->dp.rank = rank; h->dp.world = world;
  return GRL_OK;
}

// arguments of one channel over the given pieces of the bucket
static DpArgs dp_channel(grl_ctx* h, int channel, const std::vector<std::pair<int64_t, int64_t>>& pieces, char* const* flags,
                         char* const* data) {
  DpArgs d;
  memset(&d, 0, sizeof(d));
  d.rank = h->dp.rank; d.world = h->dp.world;
  d.n_ranges = (int)pieces.size();
  int64_t v = 0;
  for (size_t k = 0; k < pieces.size(); ++k) { d.start[k] = pieces[k].first; d.vstart[k] = v; v += pieces[k].second; }
  for (size_t k = pieces.size(); k <= DP_MAX_RANGES; ++k) d.vstart[k] = v;
  d.n = v;
  d.chunk = rup((v + d.world - 1) / d.world, 4);      // (the last ranks' chunks may be short or empty)
  d.timeout_ticks = (uint64_t)std::max(1, tune_int("dp_timeout_ms", 120000)) * 100000ull;     // 100 MHz wall clock
  d.host_err = h->dp_err_host;
  for (int p = 0; p < d.world; ++p) {
    d.ctl[p] = (DpCtl*)(flags[p] + (size_t)channel * dp_ctl_stride());
    d.src[p] = (float*)data[p];
    d.red[p] = (float*)(data[p] + dp_arr_bytes(h->n_train));
  }
  return d;
}
// grids of the exchange kernels (GRL_TUNE dp_blocks=reduce/gather/apply overrides: tuning aid).  Default: one quad per
// thread, no grid-stride loop -- each thread issues its loads in one batch (measured: the looping 512-block form took
// 19 us for the 5.4 MB bucket where the slab reduction with fused Adam, one quad per thread, takes 12)
static int dp_blocks(int which, int64_t quads) {
  int v[3] = {0, 0, 0};
  tune_int3("dp_blocks", v);
  const int64_t all = std::max<int64_t>(1, (quads + 255) / 256);
  return (int)(v[which] > 0 ? std::min<int64_t>(v[which], all) : std::min<int64_t>(all, 4096));
}
// K1: the reduction that ends a gradient computation, publishing its sums on channel `da` as it forms them; with `ga`
// it also carries the replay gather of the next update (multi-update calls on the device RNG)
static Op dp_k1_op(grl_ctx* self, const grl_ctx::ReducePlan& rp, const DpArgs& da, const LossArgs& la, const GatherArgs* ga, int gx,
                   const char* tag) {
  Op op; op.tag = tag; op.bytes = 8.0 * (double)da.n;
  op.join = true;                                       // (takes the place of reduce_slabs, which joins the side lane)
  const ReduceDesc* dr = self->d_reduces;
  const AdamArgs aa = self->adam_base;
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  if (ga) g = *ga;
  const int gxx = ga ? gx : 0;
  op.run = [dr, rp, la, aa, da, g, gxx](hipStream_t s) {
    const int nb = rp.n + rp.has_loss + (gxx > 0 ? gather_blocks(g, gxx) : 0);
    if (gxx > 0 && g.rows > 1) hipLaunchKernelGGL(dp_reduce_slabs_kernel<true>, dim3(nb), dim3(256), 0, s, dr, rp.tiles, rp.n, la, rp.has_loss, aa, da, g, gxx);
    else hipLaunchKernelGGL(dp_reduce_slabs_kernel<false>, dim3(nb), dim3(256), 0, s, dr, rp.tiles, rp.n, la, rp.has_loss, aa, da, g, gxx);
  };
  return op;
}
// W: one wave announces (`which` 0: ready, 1: done) and waits for every rank (the only kernel of the exchange that waits)
static Op dp_wait_op(const DpArgs& da, int which) {
  Op op; op.tag = which == 0 ? "dp_wait_ready" : "dp_wait_done";
  op.run = [da, which](hipStream_t s) { hipLaunchKernelGGL(dp_wait_kernel, dim3(1), dim3(64), 0, s, da, which); };
  return op;
}
static Op dp_reduce_op(const DpArgs& da) {
  Op op; op.tag = "dp_reduce"; op.bytes = 4.0 * (double)da.chunk * (da.world + 1);
  const int blocks = dp_blocks(0, da.chunk / 4);
  op.run = [da, blocks](hipStream_t s) { hipLaunchKernelGGL(dp_reduce_kernel, dim3(blocks), dim3(256), 0, s, da); };
  return op;
}
static Op dp_gather_op(const DpArgs& da) {
  Op op; op.tag = "dp_gather"; op.bytes = 8.0 * (double)da.n;
  const int blocks = dp_blocks(1, da.n / 4);
  op.run = [da, blocks](hipStream_t s) { hipLaunchKernelGGL(dp_gather_kernel, dim3(blocks), dim3(256), 0, s, da); };
  return op;
}
static AdamArgs dp_adam_args(grl_ctx* self, int world) {
  AdamArgs aa;
  memset(&aa, 0, sizeof(aa));
  aa.params = self->params; aa.grads = nullptr; aa.m = self->adam_m; aa.v = self->adam_v;
  aa.n_train = self->n_train; aa.sc = self->sc; aa.grad_scale = 1.f / (float)world; aa.tau = self->cfg.tau; aa.eps = 1e-8f;
  aa.src_ofs = self->vf_off; aa.n_polyak = self->n_polyak; aa.target = self->params + self->tgt_off;
  return aa;
}
static Op dp_apply_op(grl_ctx* self, const DpArgs& da, const DpArgs& db) {
  Op op; op.tag = "dp_apply"; op.bytes = (double)self->n_train * 4 * 7 + (double)self->n_polyak * 4 * 2;
  op.join = true;
  const AdamArgs aa = dp_adam_args(self, da.world);
  const int blocks = dp_blocks(2, std::max(da.chunk, db.chunk) / 4);     // (a block walks the owners' chunks one after the other)
  op.run = [aa, da, db, blocks](hipStream_t s) { hipLaunchKernelGGL(dp_apply_kernel, dim3(blocks), dim3(256), 0, s, aa, da, db); };
  return op;
}
static Op dp_apply_oneshot_op(grl_ctx* self, const DpArgs& da) {
  Op op; op.tag = "dp_apply1"; op.bytes = (double)self->n_train * 4 * (6 + da.world) + (double)self->n_polyak * 4 * 2;
  op.join = true;
  const AdamArgs aa = dp_adam_args(self, da.world);
  const int blocks = dp_blocks(2, da.n / 4);
  op.run = [aa, da, blocks](hipStream_t s) { hipLaunchKernelGGL(dp_apply_oneshot_kernel, dim3(blocks), dim3(256), 0, s, aa, da); };
  return op;
}

int grl_allreduce_connect(grl_handle h, const void* handles) {
  if (!h || !handles) return fail(GRL_ERR_INVALID, "null argument");
  if (!h->dp_buf) return fail(GRL_ERR_STATE, "call grl_allreduce_init first");
  if (h->dp_on) return fail(GRL_ERR_STATE, "already connected");
  const bool qh = h->cfg.algo != GRL_ALGO_SAC;
  // (DQN / BDQ: the batch means of the loss launch follow the reduction as a launch of their own in the split plan)
  size_t n_body = h->ops_grads.size();
  while (n_body > 0 && h->ops_grads[n_body - 1].tag == "q_finish") --n_body;
  if (n_body == 0 || h->ops_grads[n_body - 1].tag != "reduce_slabs" || !h->red_all.tiles)
    return fail(GRL_ERR_STATE, "this plan does not end in the slab reduction the exchange publishes from");
  char* flags[DP_MAX_WORLD];
  char* data[DP_MAX_WORLD];
  for (int p = 0; p < h->dp.world; ++p) {
    flags[p] = (char*)h->dp_flags;
    data[p] = (char*)h->dp_buf;
    if (p != h->dp.rank) {
      hipIpcMemHandle_t mh[2];
      memcpy(mh, (const char*)handles + (size_t)GRL_ALLREDUCE_HANDLE_BYTES * p, GRL_ALLREDUCE_HANDLE_BYTES);
      void* ptr[2] = {nullptr, nullptr};
      for (int k = 0; k < 2; ++k) {
        hipError_t e = hipIpcOpenMemHandle(&ptr[k], mh[k], hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return fail(GRL_ERR_HIP, "hipIpcOpenMemHandle (rank " + std::to_string(p) + "): " + hipGetErrorString(e));
        h->dp_peer[2 * p + k] = ptr[k];
      }
      flags[p] = (char*)ptr[0];
      data[p] = (char*)ptr[1];
    }
  }
  DpArgs none;
  memset(&none, 0, sizeof(none));
  // ---- the plain update: one exchange of the whole bucket; the gradient computation's last launch publishes (K1)
  h->dp = dp_channel(h, 0, {{0, h->n_train}}, flags, data);
  DpArgs d1s = h->dp;
  d1s.oneshot = 1;
  h->dp_body.assign(h->ops_grads.begin(), h->ops_grads.begin() + (n_body - 1));
  if (qh) {
    // DQN / BDQ: per-variable clip_by_norm sits between the all-reduce and Adam, so the sums are pulled into the gradient
    // bucket (two-shot + gather) and the plan's own apply launches follow with grad_scale = 1 / W (clip threshold W c on the
    // sum, plan_q.inl).  One form only: grl_allreduce_set_mode accepts "auto" / two-shot on these handles.
    DpArgs d = h->dp;
    d.gathered = h->grads;
    std::vector<Op>& tail = h->ops_dp;
    tail.clear();
    tail.push_back(dp_k1_op(h, h->red_all, d, h->loss_args, nullptr, 0, "reduce_publish"));
    tail.push_back(dp_wait_op(d, 0));
    tail.push_back(dp_reduce_op(d));
    tail.push_back(dp_wait_op(d, 1));
    tail.push_back(dp_gather_op(d));
    for (size_t k = n_body; k < h->ops_grads.size(); ++k) tail.push_back(h->ops_grads[k]);      // q_finish
    for (auto& o : h->ops_apply) tail.push_back(o);
    h->dp_body_per.clear();
    for (auto& o : h->dp_body)
      if (o.tag != "gather_norm") h->dp_body_per.push_back(o);
    h->ops_dp1.clear();
    for (auto* v : {&h->ops_pfdp_first, &h->ops_pfdp_mid, &h->ops_pfdp_last, &h->ops_pfdp1_first, &h->ops_pfdp1_mid, &h->ops_pfdp1_last}) v->clear();
  }
  for (int one = 0; one < 2 && !qh; ++one) {
    const DpArgs& d = one ? d1s : h->dp;
    std::vector<Op>& tail = one ? h->ops_dp1 : h->ops_dp;
    tail.clear();
    tail.push_back(dp_k1_op(h, h->red_all, d, h->loss_args, nullptr, 0, "reduce_publish"));
    tail.push_back(dp_wait_op(d, 0));
    if (one) tail.push_back(dp_apply_oneshot_op(h, d));
    else { tail.push_back(dp_reduce_op(d)); tail.push_back(dp_wait_op(d, 1)); tail.push_back(dp_apply_op(h, d, none)); }
    // multi-update calls on the device RNG: the same with the gather of the next update riding on K1 (plan_sac "prefetch")
    std::vector<Op>* pf[3] = {one ? &h->ops_pfdp1_first : &h->ops_pfdp_first, one ? &h->ops_pfdp1_mid : &h->ops_pfdp_mid,
                              one ? &h->ops_pfdp1_last : &h->ops_pfdp_last};
    for (auto* v : pf) v->clear();
    if (h->prefetch_ok) {
      const std::vector<Op>* src[3] = {&h->ops_pf_first, &h->ops_pf_mid, &h->ops_pf_last};
      for (int v = 0; v < 3; ++v) {
        pf[v]->assign(src[v]->begin(), src[v]->end() - 1);
        if (v == 2) pf[v]->push_back(dp_k1_op(h, h->red_all, d, h->loss_args, nullptr, 0, "reduce_publish"));
        else pf[v]->push_back(dp_k1_op(h, h->red_all, d, h->pf_lk, &h->pf_g2, h->pf_gx, "reduce_publish"));
        for (size_t k = 1; k < tail.size(); ++k) pf[v]->push_back(tail[k]);
      }
    }
    // ... or, with double-buffered images, the image gather riding on the head launch and the extras on K1 (plan_sac "gather_ride")
    for (int f = 0; f < 2; ++f) { h->ops_ridedp_mid[one][f].clear(); h->ops_ridedp_last[one][f].clear(); }
    h->ops_ridedp_first[one].clear();
    if (h->ride_ok) {
      auto close = [&](const std::vector<Op>& src, std::vector<Op>& dst, bool last) {
        dst.assign(src.begin(), src.end() - 1);
        if (last) dst.push_back(dp_k1_op(h, h->red_all, d, h->loss_args, nullptr, 0, "reduce_publish"));
        else dst.push_back(dp_k1_op(h, h->red_all, d, h->ride_lk, &h->ride_g2, 1, "reduce_publish"));
        for (size_t k = 1; k < tail.size(); ++k) dst.push_back(tail[k]);
      };
      close(h->ops_ride_first, h->ops_ridedp_first[one], false);
      for (int f = 0; f < 2; ++f) {
        close(h->ops_ride_mid[f], h->ops_ridedp_mid[one][f], false);
        close(h->ops_ride_last[f], h->ops_ridedp_last[one][f], true);
      }
    }
  }
  // ---- the overlapped update (grl_allreduce_set_overlap): the staged plan (grl_compute_grads_staged) with both exchanges in
  // the graph.  After heads_dfeat a SIDE LANE forms the dense layers' weight gradients, reduces them (publishing) and
  // exchanges them on channel 0 -- 90 % of the bytes -- while the main lane runs the backward through the convolutions and
  // their weight gradients; the small convolution bucket follows on channel 1; Adam waits for both.  Same tiles, same
  // sums: parameters are bit-identical to the plain update's.
  h->ops_dp_overlap.clear();
  if (h->staged_ok) {
    std::vector<std::pair<int64_t, int64_t>> dense, conv;
    int64_t off[8], num[8];
    for (int b = 0; b < 2; ++b) {
      const int nr = grl_grad_ranges(h, b, 8, off, num);
      if (nr < 0) return nr;
      for (int k = 0; k < nr; ++k) (b == 0 ? dense : conv).push_back({off[k], num[k]});
    }
    bool ok = dense.size() <= DP_MAX_RANGES && conv.size() <= DP_MAX_RANGES;
    for (auto& r : dense) ok = ok && r.first % 4 == 0 && r.second % 4 == 0;
    for (auto& r : conv) ok = ok && r.first % 4 == 0 && r.second % 4 == 0;
    int cut = -1;
    for (size_t k = 0; k < h->ops_stage0.size(); ++k)
      if (h->ops_stage0[k].tag == "heads_dfeat") cut = (int)k;
    ok = ok && h->ops_stage0.back().tag == "reduce_dense" && h->ops_stage1.back().tag == "reduce_conv";
    if (ok && cut >= 0) {
      DpArgs d0 = dp_channel(h, 0, dense, flags, data);
      const DpArgs d1 = dp_channel(h, 1, conv, flags, data);
      std::vector<Op>& L = h->ops_dp_overlap;
      for (int k = 0; k <= cut; ++k) L.push_back(h->ops_stage0[k]);
      bool first = true;
      auto side = [&](Op o) { o.lane = 1; o.fork = first; o.join = false; first = false; L.push_back(o); };
      for (size_t k = cut + 1; k + 1 < h->ops_stage0.size(); ++k) side(h->ops_stage0[k]);     // wgrad_dense
      side(dp_k1_op(h, h->red_dense, d0, h->loss_args, nullptr, 0, "reduce_dense_publish"));
      side(dp_wait_op(d0, 0));
      side(dp_reduce_op(d0));
      side(dp_wait_op(d0, 1));
      d0.gathered = (float*)((char*)h->dp_buf + 2 * dp_arr_bytes(h->n_train));     // the side lane also pulls the dense sums
      side(dp_gather_op(d0));
      for (size_t k = 0; k + 1 < h->ops_stage1.size(); ++k) { Op o = h->ops_stage1[k]; o.join = false; L.push_back(o); }
      { Op o = dp_k1_op(h, h->red_conv, d1, h->loss_args, nullptr, 0, "reduce_conv_publish"); o.join = false; L.push_back(o); }
      L.push_back(dp_wait_op(d1, 0));
      L.push_back(dp_reduce_op(d1));
      L.push_back(dp_wait_op(d1, 1));
      L.push_back(dp_apply_op(h, d0, d1));       // (joins the side lane)
    }
  }
  // ---- the running-statistics merge (grl_norm_update on a connected handle): channel 2, moment blocks behind the three arrays
  memset(&h->dp_norm, 0, sizeof(h->dp_norm));
  h->dp_norm.d = dp_channel(h, 2, {}, flags, data);
  h->dp_norm.mom_stride = (int64_t)(dp_mom_bytes(h->n_elems) / 4);
  for (int p = 0; p < h->dp.world; ++p) h->dp_norm.mom[p] = (float*)(data[p] + 3 * dp_arr_bytes(h->n_train));
  h->dp_on = true;
  return GRL_OK;
}

int grl_allreduce_disconnect(grl_handle h) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if (!h->dp_buf && !h->dp_flags) return GRL_OK;          // never initialised (or already released): nothing to do
  HIPCHK(hipStreamSynchronize(h->stream));
  h->drop_graphs();                                        // captured sequences hold the peers' addresses
  h->dp_on = false;
  h->dp_overlap = false;
  h->dp_mode = 0;
  for (auto* v : {&h->ops_dp, &h->ops_dp1, &h->ops_pfdp_first, &h->ops_pfdp_mid, &h->ops_pfdp_last, &h->ops_pfdp1_first,
                  &h->ops_pfdp1_mid, &h->ops_pfdp1_last, &h->dp_body, &h->dp_body_per, &h->ops_dp_overlap})
    v->clear();
  for (int one = 0; one < 2; ++one) {
    h->ops_ridedp_first[one].clear();
    for (int f = 0; f < 2; ++f) { h->ops_ridedp_mid[one][f].clear(); h->ops_ridedp_last[one][f].clear(); }
  }
  for (int p = 0; p < 2 * DP_MAX_WORLD; ++p)
    if (h->dp_peer[p]) { (void)hipIpcCloseMemHandle(h->dp_peer[p]); h->dp_peer[p] = nullptr; }
  if (h->dp_buf) (void)hipFree(h->dp_buf);
  if (h->dp_flags) (void)hipFree(h->dp_flags);
  h->dp_buf = h->dp_flags = nullptr;
  memset(&h->dp, 0, sizeof(h->dp));
  memset(&h->dp_norm, 0, sizeof(h->dp_norm));
  if (h->dp_err_host) *h->dp_err_host = 0u;
  return GRL_OK;
}

int grl_allreduce_set_overlap(grl_handle h, int on) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if (!h->dp_on) return fail(GRL_ERR_STATE, "call grl_allreduce_init / grl_allreduce_connect first");
  if (on && h->ops_dp_overlap.empty()) return fail(GRL_ERR_STATE, "this configuration has no staged plan: the exchange cannot be overlapped");
  h->dp_overlap = on != 0;
  return GRL_OK;
}

int grl_allreduce_set_mode(grl_handle h, int mode) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if (!h->dp_on) return fail(GRL_ERR_STATE, "call grl_allreduce_init / grl_allreduce_connect first");
  if (mode < 0 || mode > 2) return fail(GRL_ERR_INVALID, "mode: 0 auto, 1 two-shot, 2 one-shot");
  if (mode == 2 && h->cfg.algo != GRL_ALGO_SAC) return fail(GRL_ERR_STATE, "DQN / BDQ handles exchange two-shot (the clipped apply reads the gathered sums)");
  h->dp_mode = mode;
  return GRL_OK;
}

int grl_allreduce_set_timeout(grl_handle h, int ms) {
  if (!h || ms < 0) return fail(GRL_ERR_INVALID, "bad argument");
  if (!h->dp_err_host) return fail(GRL_ERR_STATE, "call grl_allreduce_init first");
  __atomic_store_n(h->dp_err_host + 1, (uint32_t)ms, __ATOMIC_RELEASE);     // read by the waiting kernels (dp_wait_all)
  return GRL_OK;
}

int grl_train_step_allreduce(grl_handle h, int n_steps, const int64_t* idx, const float* eps) {
  if (!h || n_steps < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (!h->dp_on) return fail(GRL_ERR_STATE, "call grl_allreduce_init / grl_allreduce_connect first");
  if ((idx == nullptr) != (eps == nullptr)) return fail(GRL_ERR_INVALID, "idx and eps must both be given or both be NULL");
  if (h->rp_size < 1) return fail(GRL_ERR_STATE, "replay buffer is empty");
  if (*h->dp_err_host) return fail(GRL_ERR_STATE, "an exchange timed out waiting for a peer (the replicas are no longer in step)");
  // one-shot (every rank adds all contributions itself: one kernel and one flag round less, (W - 1) n bytes per rank
  // instead of 2 (W - 1) / W n) pays for W <= 2
  const bool qh = h->cfg.algo != GRL_ALGO_SAC;
  if (qh && h->per_on) return fail(GRL_ERR_STATE, "the exchange step draws uniform minibatches: prioritised handles train with grl_train_step_per");
  if (qh) h->grad_scale = 1.f / (float)h->dp.world;       // (baked into the captured apply launches: constant per connection)
  const bool one = !qh && !h->dp_overlap && (h->dp_mode == 2 || (h->dp_mode == 0 && h->dp.world <= 2));
  std::vector<Op> none;
  std::vector<Op>* body = h->dp_overlap ? &h->ops_dp_overlap : &h->dp_body;
  std::vector<Op>* tail = h->dp_overlap ? &none : (one ? &h->ops_dp1 : &h->ops_dp);
  const std::string sfx = one ? "1" : "";
  std::vector<Op>* pf[3] = {one ? &h->ops_pfdp1_first : &h->ops_pfdp_first, one ? &h->ops_pfdp1_mid : &h->ops_pfdp_mid,
                            one ? &h->ops_pfdp1_last : &h->ops_pfdp_last};
  if (!idx && n_steps >= 2 && !h->dp_overlap && h->ride_ok && !h->prof && !h->ops_ridedp_first[one ? 1 : 0].empty()) {
    // plain exchange on the device RNG, double-buffered images: the image gather of update t+1 rides on update t's head launch
    const int o = one ? 1 : 0;
    if (int e = h->run_ride("dpr" + sfx, &h->ops_ridedp_first[o], h->ops_ridedp_mid[o], h->ops_ridedp_last[o], n_steps)) return e;
  } else if (!idx && n_steps >= 2 && !h->dp_overlap && h->prefetch_ok && !h->prof && !pf[1]->empty() && tune_int("gather_prefetch", 1)) {
    // plain exchange on the device RNG: the prefetching sequences (the gather of update t+1 rides on the reduction of update t)
    if (int e = h->run_seq("dpp_first" + sfx, {pf[0]})) return e;
    if (n_steps > 2)
      if (int e = h->run_repeated("dpp_mid" + sfx, {pf[1]}, n_steps - 2)) return e;
    if (int e = h->run_seq("dpp_last" + sfx, {pf[2]})) return e;
  } else if (!idx) {      // device RNG: identical updates, several to a graph
    if (int e = h->run_repeated((h->dp_overlap ? "dpo_rng" : "dp_rng") + sfx, {&h->ops_rng, body, tail}, n_steps)) return e;
  } else {
    for (int s = 0; s < n_steps; ++s) {
      if (int e = stage_noise(h, idx, eps, s)) return e;
      if (int e = h->run_seq((h->dp_overlap ? "dpo_explicit" : "dp_explicit") + sfx, {&h->ops_gather, body, tail})) return e;
    }
  }
  HIPCHK(hipGetLastError());
  return GRL_OK;
}

int grl_allreduce_status(grl_handle h, int64_t* exchanges, int* error) {
  if (!h || !h->dp_flags) return fail(GRL_ERR_STATE, "no exchange buffer");
  HIPCHK(hipStreamSynchronize(h->stream));
  DpCtl c[DP_CHANNELS];
  for (int k = 0; k < DP_CHANNELS; ++k)
    HIPCHK(hipMemcpy(&c[k], (char*)h->dp_flags + k * dp_ctl_stride(), sizeof(DpCtl), hipMemcpyDeviceToHost));
  if (exchanges) *exchanges = c[0].epoch;          // (channel 0 takes part in every update, plain or overlapped)
  if (getenv("GRL_PLAN_DUMP"))
    for (int k = 0; k < DP_CHANNELS; ++k) {
      fprintf(stderr, "grl dp: rank %d channel %d epoch %u error %u next_buf %u ready", h->dp.rank, k, c[k].epoch, c[k].error, c[k].next_buf);
      for (int p = 0; p < h->dp.world; ++p) fprintf(stderr, " %u", c[k].ready[p]);
      fprintf(stderr, " done");
      for (int p = 0; p < h->dp.world; ++p) fprintf(stderr, " %u", c[k].done[p]);
      fprintf(stderr, " mailbox %u\n", h->dp_err_host ? *h->dp_err_host : 0u);
    }
  int err = h->dp_err_host ? (int)*h->dp_err_host : 0;
  for (int k = 0; k < DP_CHANNELS; ++k) err |= (int)c[k].error;
  if (error) *error = err;
  if (err) return fail(GRL_ERR_STATE, "an exchange timed out waiting for a peer (the replicas are no longer in step)");
  return GRL_OK;
}

int grl_q_update_target(grl_handle h) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  if (h->cfg.algo == GRL_ALGO_SAC) return fail(GRL_ERR_STATE, "SAC has no hard target update");
  HIPCHK(hipMemcpyAsync(h->params + h->tgt_off, h->params + h->q_online_off, (size_t)h->q_online_n * 4,
                        hipMemcpyDeviceToDevice, h->stream));
  return GRL_OK;
}

int grl_encoder_load(grl_handle h, const float* const* w, const int64_t* numels, int n_arrays) {
  if (!h || !w || !numels || n_arrays != 8) return fail(GRL_ERR_INVALID, "expected 8 weight arrays");
  if (h->cfg.algo != GRL_ALGO_SAC) return fail(GRL_ERR_STATE, "grl_encoder_load is for SAC handles (auto-encoder handles encode with their own parameters)");
  const int64_t wn[8] = {7 * 7 * 32, 32, 5 * 5 * 32 * 32, 32, 3 * 3 * 32 * 32, 32, 2048 * 100, 100};
  for (int k = 0; k < 8; ++k)
    if (numels[k] != wn[k]) return fail(GRL_ERR_INVALID, "encoder weight " + std::to_string(k) + " has the wrong size");
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int k = 0; k < 8; ++k) HIPCHK(hipMemcpy(h->enc_w[k], w[k], (size_t)wn[k] * 4, hipMemcpyHostToDevice));
  h->enc_loaded = true;
  return GRL_OK;
}

int grl_encode(grl_handle h, const float* depth, int n, float* out) {
  if (!h || !depth || !out || n < 1) return fail(GRL_ERR_INVALID, "bad argument");
  if (!h->enc_loaded) return fail(GRL_ERR_STATE, "grl_encoder_load has not been called");
  if (n > h->NA) return fail(GRL_ERR_INVALID, "n exceeds act_batch");
  if (int e = pin_reserve(h, (size_t)n * 4096, (size_t)n * 100)) return e;
  memcpy(h->pin_in, depth, (size_t)n * 4096 * 4);
  HIPCHK(hipMemcpyAsync(h->ex_in, h->pin_in, (size_t)n * 4096 * 4, hipMemcpyHostToDevice, h->stream));
  if (int e = h->run_seq("encode", {&h->ops_enc})) return e;
  HIPCHK(hipMemcpyAsync(h->pin_out, h->eout, (size_t)n * 100 * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  memcpy(out, h->pin_out, (size_t)n * 100 * 4);
  return GRL_OK;
}

int64_t grl_debug_fetch(grl_handle h, const char* name, float* out, int64_t cap) {
  if (!h || !name || !out) return fail(GRL_ERR_INVALID, "null argument");
  auto it = h->dbg.find(name);
  if (it == h->dbg.end()) return fail(GRL_ERR_INVALID, std::string("unknown tensor ") + name);
  const int64_t n = std::min(cap, it->second.second);
  if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(GRL_ERR_HIP, "sync failed");
  hipError_t e = hipMemcpy(out, it->second.first, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(GRL_ERR_HIP, hipGetErrorString(e));
  return n;
}

int64_t grl_debug_store(grl_handle h, const char* name, const float* in, int64_t n) {
  if (!h || !name || !in) return fail(GRL_ERR_INVALID, "null argument");
  auto it = h->dbg.find(name);
  if (it == h->dbg.end()) return fail(GRL_ERR_INVALID, std::string("unknown tensor ") + name);
  if (n > it->second.second) return fail(GRL_ERR_INVALID, std::string("too many values for ") + name);
  if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(GRL_ERR_HIP, "sync failed");
  hipError_t e = hipMemcpy(const_cast<float*>(it->second.first), in, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail(GRL_ERR_HIP, hipGetErrorString(e));
  return n;
}

int grl_profile_enable(grl_handle h, int on) {
  if (!h) return fail(GRL_ERR_INVALID, "null handle");
  h->prof = on != 0;
  if (on) h->prof_acc.clear();
  return GRL_OK;
}

int grl_profile_query(grl_handle h, const char* name, double* avg_ms, int64_t* launches) {
  if (!h || !name) return fail(GRL_ERR_INVALID, "null argument");
  auto it = h->prof_acc.find(name);
  if (it == h->prof_acc.end()) return fail(GRL_ERR_INVALID, std::string("no profile for ") + name);
  if (avg_ms) *avg_ms = it->second.n ? it->second.ms / it->second.n : 0.0;
  if (launches) *launches = it->second.n;
  return GRL_OK;
}

/* list profiled tags: writes "tag:avg_ms:launches:flops_per_launch:bytes_per_launch:executed_flops_per_launch\n" lines */
int grl_profile_dump(grl_handle h, char* buf, int cap) {
  if (!h || !buf || cap < 1) return fail(GRL_ERR_INVALID, "bad argument");
  std::string s;
  for (auto& kv : h->prof_acc) {
    char line[256];
    snprintf(line, sizeof(line), "%s:%.6f:%lld:%.0f:%.0f:%.0f\n", kv.first.c_str(),
             kv.second.n ? kv.second.ms / kv.second.n : 0.0, (long long)kv.second.n,
             kv.second.n ? kv.second.flops / kv.second.n : 0.0, kv.second.n ? kv.second.bytes / kv.second.n : 0.0,
             kv.second.n ? kv.second.flops_exec / kv.second.n : 0.0);
    s += line;
  }
  strncpy(buf, s.c_str(), cap - 1);
  buf[cap - 1] = 0;
  return GRL_OK;
}

}  // extern "C"
