// ae_kernels_ref2.h -- TEST-ONLY reference form of ae_tapsum_mse_kernel (csrc/ae_kernels.h): one sequential loop per workgroup over
// the same descriptor, included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu).  Never part of
// libgrl.so.  No include guard: it is pasted once, inside namespace grl.
inline void ae_tapsum_mse_kernel(TapMseArgs a) {
  if (threadIdx.x != 0) return;
  float sd = 0.f, sg = 0.f;
  for (int t = 0; t < 256; ++t) {
    const long o = (long)blockIdx.x * 256 + t;
    if (o >= a.n_pix) break;
    const long n = o >> 12;
    const int oh = (int)((o >> 6) & 63), ow = (int)(o & 63);
    float s = 0.f;
    for (int kh = 0; kh < 7; ++kh) {
      const int ih = oh + kh - 3;
      if (ih < 0 || ih > 63) continue;
      for (int kw = 0; kw < 7; ++kw) {
        const int iw = ow + kw - 3;
        if (iw < 0 || iw > 63) continue;
        s += a.T[(long)(kh * 7 + kw) * a.ldT + (n << 10) + ((ih >> 1) << 5) + (iw >> 1)];
      }
    }
    const float ov = s + a.bias[0];
    a.m.out[o] = ov;
    const float d = ov - a.m.x[o];
    const float g = 2.f * d / (float)a.m.n_total;
    a.m.g_out[o] = g;
    if (a.m.g_pad) a.m.g_pad[mse_pad_index(o)] = g;
    if (a.m.gp4) a.m.gp4[mse_gp4_index(o, a.m.gp4_plane)] = g;
    sd += d * d;
    sg += g;
  }
  a.m.partial[blockIdx.x] = sd;
  a.m.partial_g[blockIdx.x] = sg;
}
