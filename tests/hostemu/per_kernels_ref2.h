// per_kernels_ref2.h -- TEST-ONLY reference form of the kernels of csrc/per_kernels.h (sequential loops over the same descriptors),
// included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).
// Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.
inline void per_update_ref(const PerArgs& a, const int64_t* idx) {
  float mx = a.st->max_priority;
  for (int k = 0; k < a.B; ++k) {
    const float pr = a.prio_in[k] + a.eps;
    a.p[idx[k]] = (double)per_powf(pr, a.alpha);
    mx = fmaxf(mx, pr);
  }
  a.st->max_priority = mx;
}
inline void per_update_kernel(PerArgs a, const int64_t* idx) {
  if (threadIdx.x == 0 && blockIdx.x == 0) per_update_ref(a, idx);
}
// block sums kept current by the apply launch (per_refresh_body): the block of sample k, if k is the first sample in it
inline void per_refresh_ref(const PerArgs& a, const int64_t* idx, int k) {
  const int64_t blk = idx[k] / PER_BLK;
  for (int j = 0; j < k; ++j)
    if (idx[j] / PER_BLK == blk) return;
  const int64_t size = a.sc->replay_size, b0 = blk * PER_BLK;
  static thread_local double tr[2 * PER_BLK];
  for (int i = 0; i < PER_BLK; ++i) tr[PER_BLK + i] = b0 + i < size ? a.p[b0 + i] : 0.0;
  for (int j = 0; j < a.B; ++j)                    // ascending: a transition drawn twice keeps the value of its last occurrence
    if (idx[j] / PER_BLK == blk) tr[PER_BLK + (int)(idx[j] - b0)] = (double)per_powf(a.prio_in[j] + a.eps, a.alpha);
  double m = INFINITY;
  for (int i = 0; i < PER_BLK; ++i)
    if (b0 + i < size) m = fmin(m, tr[PER_BLK + i]);
  a.bsum[blk] = per_tree_root(tr, PER_BLK);
  a.bmin[blk] = m;
  const int64_t e = size - 2;
  if (e >= 0 && e / PER_BLK == blk) a.st->tail_w = per_prefix_reduce(tr, e % PER_BLK, 1, [] { return 0.0; });
}
