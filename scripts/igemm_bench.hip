// igemm_bench.hip -- stand-alone micro-benchmark of the two implicit-GEMM kernels on dense problems
// with the shapes of the SAC update (development aid; not part of libgrl.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/igemm_bench.hip -o gpurun_out/igemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <vector>
#include "../deep-rl-grasping_amd/csrc/igemm.h"
#include "../deep-rl-grasping_amd/csrc/igemm2.h"

using namespace grl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// calibration: one wave per SIMD issuing dependent-free MFMAs
__global__ __launch_bounds__(256) void mfma_chain_kernel(float* out, int iters) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 a0 = {0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0];
}
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

struct Shape { const char* name; int M, N, K, nprob, split; int variant; };

template <class F>
static float time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  const char* only = argc > 1 ? argv[1] : nullptr;   // substring filter on the shape name
  const int only_cfg = argc > 2 ? atoi(argv[2]) : -2;
  {
    float* o; CK(hipMalloc(&o, 1024 * 256 * 4));
    const int iters = 4096;
    for (int wg : {256, 512}) {
      float ms = time_ms([&] { hipLaunchKernelGGL(mfma_peak_kernel, dim3(wg), dim3(256), 0, 0, o, iters); }, 5);
      double fl = (double)wg * 4 * iters * 4 * 4096.0;
      printf("mfma_peak wg=%d: %.3f ms  %.1f TFLOP/s\n", wg, ms, fl / ms / 1e9);
    }
    for (int wg : {256, 512, 1024}) {
      float ms = time_ms([&] { hipLaunchKernelGGL(mfma_chain_kernel, dim3(wg), dim3(256), 0, 0, o, iters / 4); }, 5);
      double fl = (double)wg * 4 * (iters / 4) * 16 * 4096.0;
      printf("mfma_chain (1 accumulator) wg=%d: %.3f ms  %.1f TFLOP/s\n", wg, ms, fl / ms / 1e9);
    }
    printf("I2_ABLATE=%d\n", I2_ABLATE);
  }
  const Shape shapes[] = {
      {"conv2_fwd-like  ", 9216, 64, 512, 3, 1, 0},
      {"conv3_fwd-like  ", 4096, 64, 576, 3, 1, 0},
      {"fc_fwd          ", 256, 512, 1024, 3, 1, 0},
      {"fc_bwd          ", 256, 1024, 512, 2, 1, 1},
      {"conv1_fwd-like  ", 57600, 32, 64, 3, 1, 0},
      {"conv2_wgrad-like", 512, 64, 9216, 2, 16, 2},
      {"big             ", 4096, 4096, 1024, 1, 1, 0},
      {"conv1_wgrad-like", 64, 32, 57600, 2, 96, 2},
      {"conv3_wgrad-like", 576, 64, 4096, 2, 8, 2},
      {"ksweep          ", 9216, 64, 64, 3, 1, 0},
      {"ksweep          ", 9216, 64, 128, 3, 1, 0},
      {"ksweep          ", 9216, 64, 256, 3, 1, 0},
      {"ksweep          ", 9216, 64, 1024, 3, 1, 0},
      {"ksweep          ", 9216, 64, 2048, 3, 1, 0},
      {"ksweep-1wg/cu   ", 5440, 64, 64, 3, 1, 0},
      {"ksweep-1wg/cu   ", 5440, 64, 512, 3, 1, 0},
      {"ksweep-1wg/cu   ", 5440, 64, 2048, 3, 1, 0},
      {"ksweep-4wg/cu   ", 21824, 64, 64, 3, 1, 0},
      {"ksweep-4wg/cu   ", 21824, 64, 512, 3, 1, 0},
      {"ksweep-4wg/cu   ", 21824, 64, 2048, 3, 1, 0},
  };
  for (const Shape& sh : shapes) {
    if (only && !strstr(sh.name, only)) continue;
    const size_t np = (size_t)sh.M * sh.K, nq = (size_t)sh.K * sh.N, nc = (size_t)sh.M * sh.N * sh.split;
    std::vector<float> hp(np), hq(nq);
    for (size_t i = 0; i < np; ++i) hp[i] = (float)((i * 2654435761u) % 1000) * 2e-3f - 1.f;
    for (size_t i = 0; i < nq; ++i) hq[i] = (float)((i * 40503u) % 1000) * 2e-3f - 1.f;
    std::vector<IgemmProb> probs;
    std::vector<float*> cs;
    for (int n = 0; n < sh.nprob; ++n) {
      float *dp, *dq, *dc;
      CK(hipMalloc(&dp, np * 4)); CK(hipMalloc(&dq, nq * 4)); CK(hipMalloc(&dc, nc * 4));
      CK(hipMemcpy(dp, hp.data(), np * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dq, hq.data(), nq * 4, hipMemcpyHostToDevice));
      IgemmProb p;
      memset(&p, 0, sizeof(p));
      p.p_ones_i = -1; p.M = sh.M; p.N = sh.N; p.K = sh.K; p.p_k0 = p.p_k1 = INT_MAX;
      p.p_base[0] = dp; p.q_base[0] = dq;
      if (sh.variant == 0) { p.p_ld_i[0] = sh.K; p.p_ld_r[0] = 1; p.q_ld_r[0] = sh.N; p.q_ld_j[0] = 1; }
      if (sh.variant == 1) { p.p_ld_i[0] = sh.K; p.p_ld_r[0] = 1; p.q_ld_r[0] = 1; p.q_ld_j[0] = sh.K; }
      if (sh.variant == 2) { p.p_ld_i[0] = 1; p.p_ld_r[0] = sh.M; p.q_ld_r[0] = sh.N; p.q_ld_j[0] = 1; }
      p.c = dc; p.ldc = sh.N; p.vflags = (getenv("NO_CVEC") ? 0 : VF_C_VEC);
      { void* z; CK(hipMalloc(&z, 64 * 8)); CK(hipMemset(z, 0, 64 * 8)); p.dbg_t = z; }
      const int tiles_r = (sh.K + 31) / 32;
      const int per = (tiles_r + sh.split - 1) / sh.split;
      p.k_chunk = per * 32; p.split = (sh.K + p.k_chunk - 1) / p.k_chunk; p.slab_stride = (int64_t)sh.M * sh.N;
      probs.push_back(p);
      cs.push_back(dc);
    }
    IgemmProb* dprobs;
    CK(hipMalloc(&dprobs, probs.size() * sizeof(IgemmProb)));
    CK(hipMemcpy(dprobs, probs.data(), probs.size() * sizeof(IgemmProb), hipMemcpyHostToDevice));
    const double flops = 2.0 * sh.M * sh.N * sh.K * sh.nprob;
    std::vector<float> ref;
    for (int cfg = -1; cfg < 3; ++cfg) {
      if (only_cfg > -2 && cfg != only_cfg) continue;
      if (cfg == 1 && sh.N > 32) continue;
      if (cfg == 2 && sh.variant == 2) continue;
      const int BMt = cfg < 0 ? 64 : i2_bm(cfg), BNt = cfg < 0 ? 64 : i2_bn(cfg);
      std::vector<int4> tiles;
      for (int n = 0; n < sh.nprob; ++n)
        for (int s = 0; s < probs[n].split; ++s)
          for (int ti = 0; ti < (sh.M + BMt - 1) / BMt; ++ti)
            for (int tj = 0; tj < (sh.N + BNt - 1) / BNt; ++tj) tiles.push_back(make_int4(n, s, ti, tj));
      int4* dt;
      CK(hipMalloc(&dt, tiles.size() * sizeof(int4)));
      CK(hipMemcpy(dt, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice));
      {   // the kernels index the descriptors by workgroup: one copy per tile, in work-list order (engine.hip: per_tile_descs)
        std::vector<IgemmProb> per_tile;
        for (const int4& t : tiles) per_tile.push_back(probs[t.x]);
        CK(hipFree(dprobs));
        CK(hipMalloc(&dprobs, per_tile.size() * sizeof(IgemmProb)));
        CK(hipMemcpy(dprobs, per_tile.data(), per_tile.size() * sizeof(IgemmProb), hipMemcpyHostToDevice));
      }
      dim3 grid((unsigned)tiles.size()), block(256);
      auto run = [&] {
        if (cfg < 0) {
          if (sh.variant == 0) hipLaunchKernelGGL((igemm_kernel<PM_AFFINE, QM_AFFINE, true, true, 1>), grid, block, 0, 0, dprobs, dt);
          if (sh.variant == 1) hipLaunchKernelGGL((igemm_kernel<PM_AFFINE, QM_AFFINE, true, false, 1>), grid, block, 0, 0, dprobs, dt);
          if (sh.variant == 2) hipLaunchKernelGGL((igemm_kernel<PM_AFFINE, QM_AFFINE, false, true, 1>), grid, block, 0, 0, dprobs, dt);
        } else {
#define RUN2(CF)                                                                                                            \
  if (cfg == CF) {                                                                                                          \
    if (sh.variant == 0) hipLaunchKernelGGL((igemm2_kernel<I2_P_ALONG_R, I2_Q_ALONG_J, PM_AFFINE, QM_AFFINE, CF, 0>), grid, block, 0, 0, dprobs, dt, (const int*)nullptr); \
    if (sh.variant == 1) hipLaunchKernelGGL((igemm2_kernel<I2_P_ALONG_R, I2_Q_ALONG_R, PM_AFFINE, QM_AFFINE, CF, 0>), grid, block, 0, 0, dprobs, dt, (const int*)nullptr); \
    if (sh.variant == 2) hipLaunchKernelGGL((igemm2_kernel<I2_P_ALONG_I, I2_Q_ALONG_J, PM_AFFINE, QM_AFFINE, CF, 0>), grid, block, 0, 0, dprobs, dt, (const int*)nullptr); \
  }
          RUN2(0) RUN2(1) RUN2(2)
        }
      };
      const float ms = time_ms(run, 20);
      std::vector<float> out((size_t)sh.M * sh.N * probs[0].split);
      CK(hipMemcpy(out.data(), cs[0], out.size() * 4, hipMemcpyDeviceToHost));
      double maxd = 0;
      if (cfg < 0) ref = out;
      else if (!ref.empty()) for (size_t i = 0; i < out.size(); ++i) maxd = std::max(maxd, (double)fabsf(out[i] - ref[i]));
      printf("%s M=%d N=%d K=%d x%d split=%d  %s cfg=%d tiles=%zu: %.2f us  %.1f TFLOP/s  maxdiff_vs_v1=%.2e\n", sh.name, sh.M,
             sh.N, sh.K, sh.nprob, sh.split, cfg < 0 ? "v1" : "v2", cfg, tiles.size(), ms * 1e3, flops / ms / 1e9, maxd);
#ifdef I2_TIMING
      if (cfg >= 0) {
        unsigned long long st[24];
        CK(hipMemcpy(st, probs[0].dbg_t, sizeof(st), hipMemcpyDeviceToHost));
        for (int w = 0; w < 3; ++w)
          printf("    wg %-5s start+%6lld | descr %6lld  prologue %6lld  loop %6lld  epilogue %6lld  (cycles)\n", w == 0 ? "first" : (w == 1 ? "mid" : "last"),
                 (long long)(st[8 * w] - st[0]), (long long)(st[8 * w + 1] - st[8 * w]), (long long)(st[8 * w + 2] - st[8 * w + 1]),
                 (long long)(st[8 * w + 3] - st[8 * w + 2]), (long long)(st[8 * w + 4] - st[8 * w + 3]));
      }
#endif
      CK(hipFree(dt));
    }
  }
  return 0;
}
