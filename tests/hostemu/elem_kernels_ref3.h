// elem_kernels_ref3.h -- TEST-ONLY reference form of the kernels of csrc/elem_kernels.h (sequential loops over the same descriptors),
// included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).
// Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.
inline void clip_by_norm_kernel(float* grads, const VarSeg* segs, float clip) {
  if (threadIdx.x != 0) return;
  const VarSeg sg = segs[blockIdx.x];
  float ss = 0.f;
  for (int64_t i = 0; i < sg.n; ++i) ss += grads[sg.off + i] * grads[sg.off + i];
  const float sc = clip / fmaxf(sqrtf(ss), clip);
  for (int64_t i = 0; i < sg.n; ++i) grads[sg.off + i] *= sc;
}
