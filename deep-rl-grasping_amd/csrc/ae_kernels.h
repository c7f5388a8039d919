// ae_kernels.h -- element-wise pieces of the depth auto-encoder TRAINING step:
// /root/reference/manipulation_main/gripperEnv/encoders.py:110-124 (decoder: Dense -> LeakyReLU -> Reshape,
// then UpSampling2D(2) -> Conv2D 'same' [-> LeakyReLU] three times), :127 loss 'mean_squared_error',
// :130 Adam(lr); config/encoder.yaml.  The convolutions / dense layers and their gradients run on the
// implicit-GEMM kernels (igemm2.h / igemm.h); what is here is HBM-bound glue.
#pragma once
#include "elem_kernels.h"

namespace grl {

// UpSampling2D(size=2), nearest: u[n, 2i+a, 2j+b, c] = h[n, i, j, c].  One thread per 4 channels of an
// output pixel (C % 4 == 0).  The result is written into the interior of a buffer with a border of `lo`
// pixels before and `hi` after each image row / column (kept zero), so that the 'same' convolution that
// consumes it is a 'valid' one over the bordered image: no per-tap bounds masks in the GEMM kernels.
__global__ __launch_bounds__(256) void upsample2_kernel(const float* __restrict__ h, float* __restrict__ u, int N,
                                                       int H, int W, int C, int lo, int hi) {
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
  const int Hp = 2 * H + lo + hi, Wp = 2 * W + lo + hi;
  float* dst = u + (((long)n * Hp + oh + lo) * Wp + ow + lo) * C + 4 * c4;
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
  float* out; const float* x; float* g_out; float* partial; long n_total;
  float* g_pad;      // optional copy of g_out with a zero border of 3 pixels: [N, 70, 70] (64x64 images)
  float* partial_g;  // per-workgroup sums of g_out (bias gradient of the 1-channel output conv)
  // optional: g_out de-interleaved into its four 2 x 2 sub-position planes, each [N, 36, 36] with a zero border of 2 --
  // plane (py, px) holds g[n, 2 Y + py, 2 X + px] at [n, Y + 2, X + 2].  The output convolution's weight gradient pairs the
  // gradient at (2 q + s - shift) with pixel q of the layer in FRONT of the up-sampling: along q that is a unit-stride walk
  // through ONE of these planes (plan_ae.inl), where the interleaved image would be walked with stride 2
  float* gp4; long gp4_plane;
};
__device__ __forceinline__ long mse_gp4_index(long i, long plane) {   // i over [N, 64, 64] -> [4][N, 36, 36]
  const long n = i >> 12;
  const int y = (int)((i >> 6) & 63), x = (int)(i & 63);
  return (long)((y & 1) * 2 + (x & 1)) * plane + n * 1296 + ((y >> 1) + 2) * 36 + (x >> 1) + 2;
}
__device__ __forceinline__ long mse_pad_index(long i) {   // i over [N, 64, 64] -> [N, 70, 70] interior
  const long n = i >> 12;
  const int r = (int)(i & 4095);
  return n * 4900 + ((r >> 6) + 3) * 70 + (r & 63) + 3;
}
// Output convolution 7x7 'same', 32 -> 1 channel, restructured for the matrix cores: the GEMM
// T[tap, p] = sum_c W[tap, c] u[p, c] (49 rows instead of N = 1; tap-major so that the reads below are
// coalesced across the pixels of a wavefront and every T element is read exactly once) followed by
//   out[n, oh, ow] = b + sum_{kh, kw} T[kh*7+kw, (n, oh+kh-3, ow+kw-3)]      (taps outside the image skipped)
// T is formed over the 32 x 32 pixels of the layer in FRONT of the up-sampling (u[n, ih, iw] = h[n, ih / 2, iw / 2]): the same 49
// terms in the same order as over the up-sampled image, read at the source pixel.
__global__ __launch_bounds__(256) void ae_tapsum_kernel(const float* __restrict__ T, long ldT, const float* __restrict__ bias,
                                                       float* __restrict__ out, long n_pix) {
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= n_pix) return;
  const long n = o >> 12;
  const int oh = (int)((o >> 6) & 63), ow = (int)(o & 63);
  float s = 0.f;
  for (int kh = 0; kh < 7; ++kh) {
    const int ih = oh + kh - 3;
    if (ih < 0 || ih > 63) continue;
    for (int kw = 0; kw < 7; ++kw) {
      const int iw = ow + kw - 3;
      if (iw < 0 || iw > 63) continue;
      s += T[(long)(kh * 7 + kw) * ldT + (n << 10) + ((ih >> 1) << 5) + (iw >> 1)];
    }
  }
  out[o] = s + bias[0];
}

// The gather-sum above and the MSE behind it as ONE launch (training steps; the forward-only path keeps the plain gather-sum):
// thread = output pixel; it forms out[o] exactly as ae_tapsum_kernel does, then the pixel's loss term 2 (out - x) / n and output
// gradient (g_out and its bordered / de-interleaved copies); the workgroup's sums of (out - x)^2 and
// of g go to partial[blockIdx.x] -- n_pix / 256 of them, added up by ae_finish_kernel.
struct TapMseArgs {
  const float* T; long ldT; const float* bias; long n_pix;
  MseArgs m;          // m.out is written here; m.n_total == n_pix
};
#ifdef GRL_HOSTEMU
#include "ae_kernels_ref2.h"   // tests/hostemu: the emulation build only
#else
__global__ __launch_bounds__(256) void ae_tapsum_mse_kernel(TapMseArgs a) {
  __shared__ float red[256], redg[256];
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  float dd = 0.f, g = 0.f;
  if (o < a.n_pix) {
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
    g = d * (2.f / (float)a.m.n_total);
    a.m.g_out[o] = g;
    if (a.m.g_pad) a.m.g_pad[mse_pad_index(o)] = g;
    if (a.m.gp4) a.m.gp4[mse_gp4_index(o, a.m.gp4_plane)] = g;
    dd = d * d;
  }
  red[threadIdx.x] = dd; redg[threadIdx.x] = g;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) { red[threadIdx.x] += red[threadIdx.x + off]; redg[threadIdx.x] += redg[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { a.m.partial[blockIdx.x] = red[0]; a.m.partial_g[blockIdx.x] = redg[0]; }
}
#endif

// kernel of the 7x7 output convolution re-ordered for its backward-data GEMM: Wp[kh, j, c] = W[kh, 7 - j, c] for
// j = 1..7 and 0 for j = 0 -- the taps of a kernel row in DESCENDING kw, padded to 8, so that the gradient pixels
// a quad of taps needs are 4 ascending neighbours in the bordered gradient image (one 16-byte load)

// Both re-ordered kernels of a training step are formed at its start (ae_prep_pad_kernel below) (the weights do not change inside a step): the flipped
// output kernel above, and the FIRST encoder convolution's 7x7x1 kernel with every kernel row padded to 8 taps (tap 7 = 0):
// K = 49 is no multiple of 4, so its forward ran on the scalar-gather igemm_kernel (28 us per step, rounds 3 - 4); with K = 56 --
// the eighth tap of a row reads one more pixel of the zero-bordered image and multiplies it by 0.0 -- it is an igemm2 problem
// whose quads of taps are 4 neighbouring pixels.  Adding x * 0.0 to a running fmaf sum leaves it bit for bit.
// The two launches that open a training step -- the kernel re-ordering above and the copy of the minibatch into its
// zero-bordered buffer (pad_copy_kernel) -- as ONE: blocks [0, n_prep) re-order, the rest copy (both are independent of
// everything else in the step; a launch costs ~6 us whatever it does)
__global__ __launch_bounds__(256) void ae_prep_pad_kernel(const float* __restrict__ W6, float* __restrict__ W6p, int C6,
                                                         const float* __restrict__ W1, float* __restrict__ W1p, int C1, int n_prep,
                                                         const float* __restrict__ x, float* __restrict__ xp, long total,
                                                         int H, int W, int C, int lo, int hi) {
  if ((int)blockIdx.x < n_prep) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < 7 * 8 * C6) {
      const int c = e % C6, j = (e / C6) % 8, kh = e / (8 * C6);
      const float v = j == 0 ? 0.f : W6[(kh * 7 + (7 - j)) * C6 + c];
#pragma unroll
      for (int k = 0; k < 4; ++k) W6p[e + k * 7 * 8 * C6] = v;
    }
    if (e < 7 * 8 * C1) {
      const int c = e % C1, j = (e / C1) % 8, kh = e / (8 * C1);
      W1p[e] = j == 7 ? 0.f : W1[(kh * 7 + j) * C1 + c];
    }
    return;
  }
  const long e = (long)(blockIdx.x - n_prep) * 256 + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  long r = e / C;
  const int w = (int)(r % W); r /= W;
  const int hh = (int)(r % H);
  const long n = r / H;
  xp[((n * (H + lo + hi) + hh + lo) * (W + lo + hi) + w + lo) * C + c] = x[e];
}

// loss value + Keras-Adam step size lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); advances the beta powers
__global__ void ae_finish_kernel(const float* partial, const float* partial_g, int n_partial, long n_total, float lr,
                                 DevScalars* sc, float* g_out_bias) {
  // (launched with 256 threads.  Thread t adds the partial sums k = t per, t per + 1, ... in that order -- loads of one batch,
  //  not a walk through global memory: one thread walking 256 of them was 12 us of dependent round trips -- and thread 0 adds the
  //  256 results from LDS in order: a fixed association for any number of partials)
  if (blockIdx.x != 0) return;
  float s = 0.f, sg = 0.f;
#ifndef GRL_HOSTEMU
  __shared__ float sp[2][256];
  const int per = (n_partial + 255) / 256, t = threadIdx.x;
  for (int k = t * per; k < (t + 1) * per && k < n_partial; ++k) { s += partial[k]; sg += partial_g[k]; }
  sp[0][t] = s; sp[1][t] = sg;
  __syncthreads();
  if (t != 0) return;
  s = 0.f; sg = 0.f;
  for (int k = 0; k < 256; ++k) { s += sp[0][k]; sg += sp[1][k]; }
#else      // (the emulation runs the threads of a workgroup one after the other: no staging through shared memory)
  if (threadIdx.x != 0) return;
  const int per = (n_partial + 255) / 256;
  for (int t = 0; t < 256; ++t) {
    float a = 0.f, ag = 0.f;
    for (int k = t * per; k < (t + 1) * per && k < n_partial; ++k) { a += partial[k]; ag += partial_g[k]; }
    s += a; sg += ag;
  }
#endif
  if (g_out_bias) g_out_bias[0] = sg;             // d loss / d bias of the output convolution
  sc->policy_loss = s / (float)n_total;          // reported as the reconstruction loss
  sc->adam_alpha = lr * sqrtf(1.f - sc->beta2_power) / (1.f - sc->beta1_power);
  sc->beta1_power *= 0.9f;
  sc->beta2_power *= 0.999f;
}

}  // namespace grl
