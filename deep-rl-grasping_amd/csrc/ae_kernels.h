// ae_kernels.h -- element-wise pieces of the depth auto-encoder TRAINING step:
// /root/reference/manipulation_main/gripperEnv/encoders.py:110-124 (decoder: Dense -> LeakyReLU -> Reshape,
// then UpSampling2D(2) -> Conv2D 'same' [-> LeakyReLU] three times), :127 loss 'mean_squared_error',
// :130 Adam(lr); config/encoder.yaml.  The convolutions / dense layers and their gradients run on the
// implicit-GEMM kernels (igemm2.h / igemm.h); what is here is HBM-bound glue.
#pragma once
#include "elem_kernels.h"

namespace grl {

// UpSampling2D(size=2), nearest: u[n, 2i+a, 2j+b, c] = h[n, i, j, c].  One thread per 4 channels of an
// output pixel (C % 4 == 0).
__global__ __launch_bounds__(256) void upsample2_kernel(const float* __restrict__ h, float* __restrict__ u, int N,
                                                       int H, int W, int C) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;          // quad index over [N, 2H, 2W, C/4]
  const int C4 = C / 4;
  const long total = (long)N * 2 * H * 2 * W * C4;
  if (q >= total) return;
  const int c4 = (int)(q % C4);
  long r = q / C4;
  const int ow = (int)(r % (2 * W)); r /= 2 * W;
  const int oh = (int)(r % (2 * H));
  const int n = (int)(r / (2 * H));
  const float* src = h + (((long)n * H + oh / 2) * W + ow / 2) * C + 4 * c4;
  float* dst = u + q * 4;
  dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
}

// backward of UpSampling2D(2) fused with the LeakyReLU gradient of the layer that produced h:
// g_h[n,i,j,c] = (h > 0 ? 1 : alpha) * sum_{a,b} g_u[n, 2i+a, 2j+b, c]   (fixed order a-major)
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ gu, const float* __restrict__ h,
                                                           float* __restrict__ gh, int N, int H, int W, int C,
                                                           float alpha) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;          // element index over [N, H, W, C]
  const long total = (long)N * H * W * C;
  if (e >= total) return;
  const int c = (int)(e % C);
  long r = e / C;
  const int j = (int)(r % W); r /= W;
  const int i = (int)(r % H);
  const int n = (int)(r / H);
  const float* p = gu + (((long)n * 2 * H + 2 * i) * 2 * W + 2 * j) * C + c;
  const long rs = (long)2 * W * C;
  float s = p[0];
  s += p[C];
  s += p[rs];
  s += p[rs + C];
  gh[e] = h[e] > 0.f ? s : alpha * s;
}

// MSE: g_out = 2 (out - x) / n_total, partial sums of (out - x)^2 per workgroup (fixed tree order)
struct MseArgs {
  const float* out; const float* x; float* g_out; float* partial; long n_total;
};
#ifdef GRL_HOSTEMU
inline void mse_kernel(MseArgs a) {
  if (threadIdx.x != 0) return;
  const long per = (a.n_total + gridDim.x - 1) / gridDim.x;
  const long i0 = (long)blockIdx.x * per, i1 = std::min(a.n_total, i0 + per);
  float s = 0.f;
  for (long i = i0; i < i1; ++i) {
    const float d = a.out[i] - a.x[i];
    a.g_out[i] = 2.f * d / (float)a.n_total;
    s += d * d;
  }
  a.partial[blockIdx.x] = s;
}
#else
__global__ __launch_bounds__(256) void mse_kernel(MseArgs a) {
  __shared__ float red[256];
  const long per = (a.n_total + gridDim.x - 1) / gridDim.x;
  const long i0 = (long)blockIdx.x * per, i1 = min(a.n_total, i0 + per);
  const float inv = 2.f / (float)a.n_total;
  float s = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    const float d = a.out[i] - a.x[i];
    a.g_out[i] = d * inv;
    s += d * d;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.partial[blockIdx.x] = red[0];
}
#endif

// loss value + Keras-Adam step size lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); advances the beta powers
__global__ void ae_finish_kernel(const float* partial, int n_partial, long n_total, float lr, DevScalars* sc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = 0.f;
  for (int k = 0; k < n_partial; ++k) s += partial[k];
  sc->policy_loss = s / (float)n_total;          // reported as the reconstruction loss
  sc->adam_alpha = lr * sqrtf(1.f - sc->beta2_power) / (1.f - sc->beta1_power);
  sc->beta1_power *= 0.9f;
  sc->beta2_power *= 0.999f;
}

}  // namespace grl
