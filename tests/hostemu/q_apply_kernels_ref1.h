// q_apply_kernels_ref1.h -- TEST-ONLY reference form of the kernels of csrc/q_apply_kernels.h: sequential loops over
// the same descriptors, included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu,
// tests/conftest.py).  Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.
inline void q_finish_ref(DevScalars* sc, const float* row_part, int rows, bool tick_rng = true) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int r = 0; r < rows; ++r) { s0 += row_part[3 * r]; s1 += row_part[3 * r + 1]; s2 += row_part[3 * r + 2]; }
  q_metrics(sc, rows, s0, s1, s2, tick_rng);
}
inline void q_finish_kernel(DevScalars* sc, const float* row_part, int rows) {
  if (threadIdx.x == 0) q_finish_ref(sc, row_part, rows);
}
inline void q_reduce_clip_adam_kernel(const ReduceDesc* descs, int n_desc, float clip, AdamArgs aa, const float* row_part, int rows,
                                      int finish, PerArgs per, const int64_t* per_idx, int n_extra, QNextArgs nx) {
  if ((int)blockIdx.x >= n_desc + n_extra) {   // the next update's sampler / uniform gather rides on this launch
    if (nx.uniform_gx > 0) {
      const int k = (int)blockIdx.x - n_desc - n_extra, gx = nx.uniform_gx;
      if (threadIdx.x < 256) gather_norm_body(nx.g, k % gx, (k / gx) % nx.g.B, k / (gx * nx.g.B));
      return;
    }
    per_sample_ref(nx.per, nx.n_blocks, nx.g, 1, (int)blockIdx.x - n_desc - n_extra);
    return;
  }
  if ((int)blockIdx.x == n_desc + 1) {   // priority write-back (prioritised replay)
    if (threadIdx.x == 0) per_update_ref(per, per_idx);
    return;
  }
  if ((int)blockIdx.x > n_desc + 1) {    // block sums of the sum tree refreshed by the apply launch (multi-update calls)
    if (threadIdx.x == 0) per_refresh_ref(per, per_idx, (int)blockIdx.x - n_desc - 2);
    return;
  }
  if (threadIdx.x != 0) return;
  if ((int)blockIdx.x == n_desc) {
    if (finish & 1) q_finish_ref(const_cast<DevScalars*>(aa.sc), row_part, rows, (finish & 2) == 0);
    return;
  }
  const ReduceDesc d = descs[blockIdx.x];
  std::vector<float> g(d.n);
  float ss = 0.f;
  for (int i = 0; i < d.n; ++i) {
    const float* src = d.src + (d.row_len > 0 ? (long)(i / d.row_len) * d.src_ld + i % d.row_len : i);
    float s = 0.f;
    for (int k = 0; k < d.splits; ++k) s += src[(long)k * d.slab_stride];
    g[i] = s;
    ss += s * s;
  }
  const float sc = clip > 0.f ? clip / fmaxf(sqrtf(ss), clip) : 1.f;
  const float alpha = aa.sc->adam_alpha;
  const int64_t e0 = d.dst - aa.grads;
  for (int i = 0; i < d.n; ++i) {
    const float gi = clip > 0.f ? g[i] * sc : g[i];
    d.dst[i] = gi;
    adam_elem(grad_scaled(gi, aa.grad_scale), aa.params[e0 + i], aa.m[e0 + i], aa.v[e0 + i], alpha, aa.eps);
  }
}
