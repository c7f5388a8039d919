// conv_stack_bench.hip -- gate for the sample-local convolution stack (csrc/conv_stack.h; VERDICT r4 "Next 2"): the fused
// conv1 -> conv2 -> conv3 forward of 3 networks x B samples as ONE launch, checked against a tap-loop CPU reference and
// timed.  What it replaces (profiles/r04_rocprofv3_summary_sac_depth.txt, graph replay): conv1_fwd 15.5 + conv2_fwd 24.0 +
// conv3_fwd 15.6 = 55.1 us.  Gate: <= 42 us.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/conv_stack_bench.hip -o gpurun_out/conv_stack_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../deep-rl-grasping_amd/csrc/conv_stack.h"

using namespace grl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static std::vector<float> rnd(size_t n, float lo, float hi, unsigned seed) {
  std::vector<float> v(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& x : v) { s = s * 1664525u + 1013904223u; x = lo + (hi - lo) * ((s >> 8) * (1.f / 16777216.f)); }
  return v;
}
template <class T> static T* up(const std::vector<T>& v) {
  T* d; CK(hipMalloc(&d, v.size() * sizeof(T))); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d;
}
static void conv_ref(const float* x, int H, int W, int Ci, const float* w, const float* b, int KH, int S, int Co, std::vector<float>& y) {
  const int OH = (H - KH) / S + 1, OW = (W - KH) / S + 1;
  y.assign((size_t)OH * OW * Co, 0.f);
  for (int oh = 0; oh < OH; ++oh)
    for (int ow = 0; ow < OW; ++ow)
      for (int co = 0; co < Co; ++co) {
        double acc = b[co];
        for (int kh = 0; kh < KH; ++kh)
          for (int kw = 0; kw < KH; ++kw)
            for (int ci = 0; ci < Ci; ++ci)
              acc += (double)x[((oh * S + kh) * W + ow * S + kw) * Ci + ci] * w[((kh * KH + kw) * Ci + ci) * Co + co];
        y[((size_t)oh * OW + ow) * Co + co] = acc > 0 ? (float)acc : 0.f;
      }
}

static int g_dynlds = 0;     // extra (unused) dynamic LDS per workgroup: limits how many workgroups share a CU (measurement)
// the same convolution as a float32 chain of fmaf in increasing k = (kh, kw, ci): what the kernel must reproduce BIT FOR BIT
static void conv_ref_f32(const float* x, int H, int W, int Ci, const float* w, const float* b, int KH, int S, int Co, std::vector<float>& y) {
  const int OH = (H - KH) / S + 1, OW = (W - KH) / S + 1;
  y.assign((size_t)OH * OW * Co, 0.f);
  for (int oh = 0; oh < OH; ++oh)
    for (int ow = 0; ow < OW; ++ow)
      for (int co = 0; co < Co; ++co) {
        float acc = 0.f;
        for (int kh = 0; kh < KH; ++kh)
          for (int kw = 0; kw < KH; ++kw)
            for (int ci = 0; ci < Ci; ++ci)
              acc = fmaf(x[((oh * S + kh) * W + ow * S + kw) * Ci + ci], w[((kh * KH + kw) * Ci + ci) * Co + co], acc);
        const float v = acc + b[co];
        y[((size_t)oh * OW + ow) * Co + co] = v > 0.f ? v : 0.f;
      }
}

template <int C> static int run(int B, int reps) {
  const int NN = 3;
  std::vector<ConvStackNet> nets(NN);
  std::vector<std::vector<float>> hx(2), hw[3], hb[3];
  hx[0] = rnd((size_t)B * 4096 * C, -0.04f, 0.04f, 1);      // (normalised observations / 255: |x| <= 10 / 255)
  hx[1] = rnd((size_t)B * 4096 * C, -0.04f, 0.04f, 2);
  float* dx[2] = {up(hx[0]), up(hx[1])};
  const int K[3] = {64 * C, 512, 576}, N[3] = {32, 64, 64};
  float *a1[3], *a2[3], *a3[3];
  float* a1_pair; CK(hipMalloc(&a1_pair, (size_t)B * 225 * 64 * 4));
  for (int n = 0; n < NN; ++n) {
    for (int l = 0; l < 3; ++l) {
      hw[l].push_back(rnd((size_t)K[l] * N[l], -1.4f / sqrtf((float)K[l]) * 1.7f, 1.4f / sqrtf((float)K[l]) * 1.7f, 10 + 3 * n + l));
      hb[l].push_back(rnd(N[l], -0.05f, 0.05f, 40 + 3 * n + l));
      nets[n].w[l] = up(hw[l][n]); nets[n].b[l] = up(hb[l][n]);
    }
    nets[n].x = dx[n == 2 ? 1 : 0];
    if (n < 2) a1[n] = a1_pair + 32 * n; else CK(hipMalloc(&a1[n], (size_t)B * 225 * 64 * 4));
    CK(hipMalloc(&a2[n], (size_t)B * 36 * 64 * 4)); CK(hipMalloc(&a3[n], (size_t)B * 16 * 64 * 4));
    nets[n].a1 = n < 2 ? a1[n] : nullptr; nets[n].a2 = n < 2 ? a2[n] : nullptr; nets[n].a3 = a3[n]; nets[n].ld1 = 64;
#ifdef CS_NOSTORE      // (measurement: what the activation stores of the two trained networks cost)
    nets[n].a1 = nullptr; nets[n].a2 = nullptr;
#endif
  }
  ConvStackArgs args;
  memset(&args, 0, sizeof(args));
  for (int n = 0; n < NN; ++n) args.nets[n] = nets[n];
  args.B = B; args.n_nets = NN;
#ifdef CS_STAMPS
  CK(hipMalloc(&args.stamps, (size_t)B * NN * 16 * 8));
  CK(hipMemset(args.stamps, 0, (size_t)B * NN * 16 * 8));
#endif
  auto launch = [&] { hipLaunchKernelGGL(conv_stack_fwd_kernel<C>, dim3(B * NN), dim3(256), g_dynlds, 0, args); };
  launch();
  CK(hipDeviceSynchronize());
  // ---- check a few samples of every network against the tap loops
  int bad = 0;
  long bits_bad = 0;
  double worst = 0;
  std::vector<float> g1((size_t)B * 225 * 64), g2((size_t)B * 36 * 64), g3((size_t)B * 16 * 64);
  for (int n = 0; n < NN; ++n) {
    CK(hipMemcpy(g3.data(), a3[n], g3.size() * 4, hipMemcpyDeviceToHost));
    if (n < 2 && nets[n].a1) {
      CK(hipMemcpy(g1.data(), a1_pair, g1.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(g2.data(), a2[n], g2.size() * 4, hipMemcpyDeviceToHost));
    }
    for (int s : {0, 1, B / 2, B - 1}) {
      std::vector<float> y1, y2, y3;
      conv_ref(hx[n == 2 ? 1 : 0].data() + (size_t)s * 4096 * C, 64, 64, C, hw[0][n].data(), hb[0][n].data(), 8, 4, 32, y1);
      conv_ref(y1.data(), 15, 15, 32, hw[1][n].data(), hb[1][n].data(), 4, 2, 64, y2);
      conv_ref(y2.data(), 6, 6, 64, hw[2][n].data(), hb[2][n].data(), 3, 1, 64, y3);
      auto cmp = [&](float got, float ref, const char* what, int idx) {
        const double d = fabs((double)got - ref), tol = 1e-5 + 1e-4 * fabs(ref);
        worst = fmax(worst, d / tol);
        if (d > tol && bad++ < 10) printf("MISMATCH net %d sample %d %s[%d]: got %g ref %g\n", n, s, what, idx, got, ref);
      };
      for (int i = 0; i < 16 * 64; ++i) cmp(g3[(size_t)s * 1024 + i], y3[i], "a3", i);
      {   // bit identity with the sequential fmaf chain (layer by layer on the chain's own activations)
        std::vector<float> z1, z2, z3;
        conv_ref_f32(hx[n == 2 ? 1 : 0].data() + (size_t)s * 4096 * C, 64, 64, C, hw[0][n].data(), hb[0][n].data(), 8, 4, 32, z1);
        conv_ref_f32(z1.data(), 15, 15, 32, hw[1][n].data(), hb[1][n].data(), 4, 2, 64, z2);
        conv_ref_f32(z2.data(), 6, 6, 64, hw[2][n].data(), hb[2][n].data(), 3, 1, 64, z3);
        for (int i = 0; i < 16 * 64; ++i) bits_bad += g3[(size_t)s * 1024 + i] != z3[i];
        if (n < 2 && nets[n].a1) {
          for (int p = 0; p < 225; ++p)
            for (int ch = 0; ch < 32; ++ch) bits_bad += g1[((size_t)s * 225 + p) * 64 + 32 * n + ch] != z1[p * 32 + ch];
          for (int i = 0; i < 36 * 64; ++i) bits_bad += g2[(size_t)s * 2304 + i] != z2[i];
        }
      }
      if (n < 2 && nets[n].a1) {
        for (int p = 0; p < 225; ++p)
          for (int ch = 0; ch < 32; ++ch) cmp(g1[((size_t)s * 225 + p) * 64 + 32 * n + ch], y1[p * 32 + ch], "a1", p * 32 + ch);
        for (int i = 0; i < 36 * 64; ++i) cmp(g2[(size_t)s * 2304 + i], y2[i], "a2", i);
      }
    }
  }
  printf("C=%d B=%d: check %s (worst error / tolerance %.3f); bit-identical to the sequential fmaf chain: %s (%ld elements differ)\n", C, B,
         bad ? "FAILED" : "ok", worst, bits_bad ? "NO" : "yes", bits_bad);
#ifdef CS_STAMPS
  {   // phase stamps of the LAST warm launch: per boundary the mean / min / max over workgroups, relative to the first start
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)B * NN * 16);
    CK(hipMemcpy(st.data(), args.stamps, st.size() * 8, hipMemcpyDeviceToHost));
    // (the shader clocks of the XCDs are not synchronised: only differences inside a workgroup mean anything; the 100 MHz
    //  wall clock in slots 8 / 9 is chip-wide)
    const char* names[8] = {"start", "conv1 done", "barrier 1", "(unused)", "conv2 done", "barrier 2", "a2 copy issued", "end"};
    printf("  phase lengths in shader-clock ticks, mean / max over the workgroups:\n");
    for (int k = 1; k < 8; ++k) {
      if (k == 3) continue;
      const int kp = k == 4 ? 2 : k - 1;
      double dsum = 0, dmx = 0;
      for (int b = 0; b < B * NN; ++b) {
        const double d = (double)(st[b * 16 + k] - st[b * 16 + kp]);
        dsum += d; dmx = std::max(dmx, d);
      }
      printf("    -> %-16s mean %8.0f  max %8.0f\n", names[k], dsum / (B * NN), dmx);
    }
    unsigned long long w0 = ~0ull, w1 = 0;
    for (int b = 0; b < B * NN; ++b) { w0 = std::min(w0, st[b * 16 + 8]); w1 = std::max(w1, st[b * 16 + 9]); }
    printf("  wall clock (10 ns ticks): first start -> last end %llu;  start offsets / durations by dispatch third:\n", w1 - w0);
    for (int third = 0; third < 3; ++third) {
      double s0 = 0, s1 = 0, d = 0, dm = 0; int n = 0;
      for (int b = third * B * NN / 3; b < (third + 1) * B * NN / 3; ++b) {
        s0 += (double)(st[b * 16 + 8] - w0); s1 += (double)(st[b * 16 + 9] - w0);
        d += (double)(st[b * 16 + 9] - st[b * 16 + 8]); dm = std::max(dm, (double)(st[b * 16 + 9] - st[b * 16 + 8])); ++n;
      }
      printf("    blocks %4d..%4d: start %7.1f  end %7.1f  duration mean %7.1f max %7.1f\n", third * B * NN / 3, (third + 1) * B * NN / 3 - 1,
             s0 / n, s1 / n, d / n, dm);
    }
  }
#endif
  // ---- timing
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / reps;
  const double fl = 2.0 * B * NN * (225.0 * 32 * 64 * C + 36.0 * 64 * 512 + 16.0 * 64 * 576);
  printf("C=%d B=%d: %.2f us per launch (back to back, %d launches)  %.1f TFLOP/s algorithmic = %.3f of the f32 MFMA peak\n", C, B, us, reps,
         fl / us / 1e6, fl / us / 1e6 / 157.3);
  return bad;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 300;
  int bad = 0;
  if (argc > 3) {
    g_dynlds = atoi(argv[3]);
    printf("dynamic LDS pad %d bytes per workgroup\n", g_dynlds);
  }
  bad += run<1>(256, reps);
  if (argc > 2) return bad ? 1 : 0;      // (any second argument: the headline shape only)
  bad += run<1>(128, reps);
  bad += run<1>(16, reps);
  bad += run<4>(256, reps);
  bad += run<2>(64, reps);
  return bad ? 1 : 0;
}
