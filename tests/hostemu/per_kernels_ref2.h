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
