// ae_kernels_ref1.h -- TEST-ONLY reference form of the kernels of csrc/ae_kernels.h (sequential loops over the same descriptors),
// included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).
// Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.
inline void mse_kernel(MseArgs a) {
  if (threadIdx.x != 0) return;
  const long per = (a.n_total + gridDim.x - 1) / gridDim.x;
  const long i0 = (long)blockIdx.x * per, i1 = std::min(a.n_total, i0 + per);
  float s = 0.f, sg = 0.f;
  for (long i = i0; i < i1; ++i) {
    const float d = a.out[i] - a.x[i];
    const float g = 2.f * d / (float)a.n_total;
    a.g_out[i] = g;
    if (a.g_pad) a.g_pad[mse_pad_index(i)] = g;
    if (a.gp4) a.gp4[mse_gp4_index(i, a.gp4_plane)] = g;
    s += d * d;
    sg += g;
  }
  a.partial[blockIdx.x] = s;
  a.partial_g[blockIdx.x] = sg;
}
