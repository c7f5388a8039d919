// per_kernels_ref1.h -- TEST-ONLY reference form of the kernels of csrc/per_kernels.h (sequential loops over the same descriptors),
// included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).
// Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.
inline double per_tree_root(double* tr, int leaves) {   // tr[leaves + i] filled; builds tr[1 .. leaves)
  for (int i = leaves - 1; i >= 1; --i) tr[i] = tr[2 * i] + tr[2 * i + 1];
  return tr[1];
}
inline int per_tree_walk(const double* tr, int leaves, double& rem) {
  int i = 1;
  while (i < leaves) {
    const double left = tr[2 * i];
    if (left > rem) i = 2 * i;
    else { rem -= left; i = 2 * i + 1; }
  }
  return i - leaves;
}
inline void per_blocksum_kernel(PerArgs a) {
  if (threadIdx.x != 0) return;
  const int64_t size = a.sc->replay_size;
  const int64_t i0 = (int64_t)blockIdx.x * PER_BLK;
  static thread_local double tr[2 * PER_BLK];
  float m = INFINITY;
  for (int i = 0; i < PER_BLK; ++i) {
    const bool in = i0 + i < size;
    tr[PER_BLK + i] = in ? (double)a.p[i0 + i] : 0.0;
    if (in) m = fminf(m, a.p[i0 + i]);
  }
  a.bsum[blockIdx.x] = per_tree_root(tr, PER_BLK);
  a.bmin[blockIdx.x] = m;
}
inline void per_sample_kernel(PerArgs a, int n_blocks, GatherArgs g, int do_gather) {
  const int k = blockIdx.x;
  const int64_t size = a.sc->replay_size;
  static thread_local double tr[2 * PER_BLK];
  float pmin = INFINITY;
  for (int j = 0; j < PER_BLK; ++j) {
    tr[PER_BLK + j] = j < n_blocks ? a.bsum[j] : 0.0;
    if (j < n_blocks) pmin = fminf(pmin, a.bmin[j]);
  }
  const double total = per_tree_root(tr, PER_BLK);
  float u;
  if (a.u) u = a.u[k];
  else {
    const uint64_t step = a.sc->rng_step;
    uint32_t c[4] = {(uint32_t)step, (uint32_t)(step >> 32), (uint32_t)k, 0x50455221u};
    philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    u = (float)(c[0] >> 8) * (1.f / 16777216.f);
  }
  double rem = ((double)u + (double)k) * total / (double)a.B;
  const int j = per_tree_walk(tr, PER_BLK, rem);
  const int64_t b0 = (int64_t)j * PER_BLK;
  for (int i = 0; i < PER_BLK; ++i) tr[PER_BLK + i] = (j < n_blocks && b0 + i < size) ? (double)a.p[b0 + i] : 0.0;
  per_tree_root(tr, PER_BLK);
  const int64_t i = std::min<int64_t>(b0 + per_tree_walk(tr, PER_BLK, rem), size - 1);
  if (do_gather) {                       // (every emulated thread walks the tree, as on the device, then gathers its elements)
    gather_row_device(g, k, i);
    if (g.adam_tick && k == 0 && threadIdx.x == 0) adam_tick_device(g.sc);
  }
  if (threadIdx.x != 0) return;
  a.idx_out[k] = i;
  const double ps = (double)a.p[i] / total, pm = (double)pmin / total;
  a.w_out[k] = (float)(pow(ps * (double)size, -(double)a.st->beta) / pow(pm * (double)size, -(double)a.st->beta));
  if (k == 0) { a.st->total = total; a.st->p_min = pmin; if (!a.u) a.sc->rng_used = 1u; }
}
