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
