// conv_stack_ref1.h -- TEST-ONLY reference form of conv_stack_fwd_kernel (csrc/conv_stack.h): the three VALID convolutions of
// one sample of one network as tap loops with a sequential fmaf accumulation, included by that header ONLY in the g++
// emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).  Never part of libgrl.so.  No include guard: it is
// pasted once, inside namespace grl.
static inline void cs_ref_conv(const float* x, int W, int Ci, const float* w, const float* b, int KH, int S, int OH, int Co, float* y) {
  for (int oh = 0; oh < OH; ++oh)
    for (int ow = 0; ow < OH; ++ow)
      for (int co = 0; co < Co; ++co) {
        float acc = 0.f;
        for (int kh = 0; kh < KH; ++kh)
          for (int kw = 0; kw < KH; ++kw)
            for (int ci = 0; ci < Ci; ++ci)
              acc = fmaf(x[((oh * S + kh) * W + ow * S + kw) * Ci + ci], w[((kh * KH + kw) * Ci + ci) * Co + co], acc);
        y[(oh * OH + ow) * Co + co] = fmaxf(acc + b[co], 0.f);
      }
}
template <int C>
void conv_stack_fwd_kernel(ConvStackArgs a) {
  if (threadIdx.x != 0) return;
  const int net_i = blockIdx.x / a.B, smp = blockIdx.x % a.B;
  const ConvStackNet& net = a.nets[net_i];
  static thread_local float y1[225 * 32], y2[36 * 64], y3[16 * 64];
  cs_ref_conv(net.x + (long)smp * 64 * 64 * C, 64, C, net.w[0], net.b[0], 8, 4, 15, 32, y1);
  cs_ref_conv(y1, 15, 32, net.w[1], net.b[1], 4, 2, 6, 64, y2);
  cs_ref_conv(y2, 6, 64, net.w[2], net.b[2], 3, 1, 4, 64, y3);
  if (net.a1)
    for (int p = 0; p < 225; ++p)
      for (int c = 0; c < 32; ++c) net.a1[((long)smp * 225 + p) * net.ld1 + c] = y1[p * 32 + c];
  if (net.a2)
    for (int i = 0; i < 36 * 64; ++i) net.a2[(long)smp * 36 * 64 + i] = y2[i];
  for (int i = 0; i < 16 * 64; ++i) net.a3[(long)smp * 16 * 64 + i] = y3[i];
}

// TEST-ONLY reference of conv_stack_bwd_kernel: the transposed convolutions as scatter loops over the forward taps
template <int DUMMY>
void conv_stack_bwd_kernel(ConvStackBwdArgs a) {
  if (threadIdx.x != 0) return;
  const int net_i = blockIdx.x / a.B, smp = blockIdx.x % a.B;
  const ConvStackBwdNet& net = a.nets[net_i];
  static thread_local float g2[36 * 64], g1[225 * 32];
  for (int i = 0; i < 36 * 64; ++i) g2[i] = 0.f;
  for (int i = 0; i < 225 * 32; ++i) g1[i] = 0.f;
  const float* g3 = net.g3 + (long)smp * 16 * 64;
  for (int kh = 0; kh < 3; ++kh)
    for (int kw = 0; kw < 3; ++kw)
      for (int oh = 0; oh < 4; ++oh)
        for (int ow = 0; ow < 4; ++ow)
          for (int ci = 0; ci < 64; ++ci) {
            float acc = 0.f;
            for (int co = 0; co < 64; ++co) acc = fmaf(g3[(oh * 4 + ow) * 64 + co], net.w3[((kh * 3 + kw) * 64 + ci) * 64 + co], acc);
            g2[((oh + kh) * 6 + ow + kw) * 64 + ci] += acc;
          }
  for (int i = 0; i < 36 * 64; ++i) {
    if (!(net.a2[(long)smp * 36 * 64 + i] > 0.f)) g2[i] = 0.f;
    net.g2[(long)smp * 36 * 64 + i] = g2[i];
  }
  for (int kh = 0; kh < 4; ++kh)
    for (int kw = 0; kw < 4; ++kw)
      for (int oh = 0; oh < 6; ++oh)
        for (int ow = 0; ow < 6; ++ow)
          for (int ci = 0; ci < 32; ++ci) {
            float acc = 0.f;
            for (int co = 0; co < 64; ++co) acc = fmaf(g2[(oh * 6 + ow) * 64 + co], net.w2[((kh * 4 + kw) * 32 + ci) * 64 + co], acc);
            g1[((2 * oh + kh) * 15 + 2 * ow + kw) * 32 + ci] += acc;
          }
  for (int p = 0; p < 225; ++p)
    for (int ci = 0; ci < 32; ++ci) {
      const long at = ((long)smp * 225 + p) * net.ld1 + ci;
      net.g1[at] = net.a1[at] > 0.f ? g1[p * 32 + ci] : 0.f;
    }
}
