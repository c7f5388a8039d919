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
  double m = INFINITY;
  for (int i = 0; i < PER_BLK; ++i) {
    const bool in = i0 + i < size;
    tr[PER_BLK + i] = in ? a.p[i0 + i] : 0.0;
    if (in) m = fmin(m, a.p[i0 + i]);
  }
  a.bsum[blockIdx.x] = per_tree_root(tr, PER_BLK);
  a.bmin[blockIdx.x] = m;
  const int64_t e = size - 2;
  if (e >= 0 && e / PER_BLK == (int64_t)blockIdx.x) a.st->tail_w = per_prefix_reduce(tr, e % PER_BLK, 1, [] { return 0.0; });
}
// sample k (run by every emulated thread of the workgroup that owns it; threads beyond 255 stay out)
inline void per_sample_ref(const PerArgs& a, int n_blocks, const GatherArgs& g, int do_gather, const int k) {
  if (threadIdx.x >= 256) return;
  const int64_t size = a.sc->replay_size;
  static thread_local double tr[2 * PER_BLK];
  double pmin = INFINITY;
  for (int j = 0; j < PER_BLK; ++j) {
    tr[PER_BLK + j] = j < n_blocks ? a.bsum[j] : 0.0;
    if (j < n_blocks) pmin = fmin(pmin, a.bmin[j]);
  }
  const double total = per_tree_root(tr, PER_BLK);
  const double tail = a.st->tail_w;
  const double total_s = size >= 2 ? per_prefix_reduce(tr, size - 2, PER_BLK, [tail] { return tail; }) : 0.0;
  double rem = per_mass(a, k, per_uniform(a, k), total_s);
  const int j = per_tree_walk(tr, PER_BLK, rem);
  const int64_t b0 = (int64_t)j * PER_BLK;
  for (int i = 0; i < PER_BLK; ++i) tr[PER_BLK + i] = (j < n_blocks && b0 + i < size) ? a.p[b0 + i] : 0.0;
  per_tree_root(tr, PER_BLK);
  const int64_t i = std::min<int64_t>(b0 + per_tree_walk(tr, PER_BLK, rem), size - 1);
  if (do_gather) {                       // (every emulated thread walks the tree, as on the device, then gathers its elements)
    gather_row_device(g, k, i);
    if (g.adam_tick && k == 0 && threadIdx.x == 0) adam_tick_device(g.sc);
  }
  if (threadIdx.x != 0) return;
  a.idx_out[k] = i;
  a.w_out[k] = per_weight(a.p[i], total, pmin, size, a.st->beta);
  if (k == 0) { a.st->total = total; a.st->total_s = total_s; a.st->p_min = pmin; if (!a.u) a.sc->rng_used = 1u; }
}
inline void per_sample_kernel(PerArgs a, int n_blocks, GatherArgs g, int do_gather) {
  per_sample_ref(a, n_blocks, g, do_gather, (int)blockIdx.x);
}
