// act_heads_ref1.h -- TEST-ONLY reference form of act_heads_kernel (csrc/elem_kernels.h): sequential loops over the same
// descriptor, included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).
// Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.
inline void act_heads_kernel(ActHeadsArgs a) {
  if (threadIdx.x != 0) return;
  const int r = blockIdx.x;
  if (r >= a.rows) return;
  float buf[2][ACT_HEADS_MAX_IN];
  for (int i = 0; i < a.K0; ++i) buf[0][i] = a.x[(long)r * a.ldx + i];
  int cur = 0, K = a.K0;
  for (int l = 0; l < a.L; ++l) {
    for (int o = 0; o < a.hid[l]; ++o) {
      float v = a.b[l][o];
      for (int i = 0; i < K; ++i) v += buf[cur][i] * a.w[l][(long)i * a.hid[l] + o];
      buf[cur ^ 1][o] = fmaxf(v, 0.f);
    }
    cur ^= 1;
    K = a.hid[l];
  }
  for (int o = 0; o < a.A; ++o) {
    float u = a.ob[0][o], s = a.ob[1][o];
    for (int i = 0; i < K; ++i) { u += buf[cur][i] * a.ow[0][(long)i * a.A + o]; s += buf[cur][i] * a.ow[1][(long)i * a.A + o]; }
    const int j = r * a.A + o;
    a.mu[j] = u;
    a.ls[j] = s;
    if (!a.deterministic) u += expf(fminf(fmaxf(s, GRL_LOG_STD_MIN), GRL_LOG_STD_MAX)) * a.eps[j];
    a.out[j] = tanhf(u);
  }
}
