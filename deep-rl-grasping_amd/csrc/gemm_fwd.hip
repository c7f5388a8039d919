// gemm_fwd.hip -- forward products (dense layers, VALID and padded convolutions): igemm2_kernel with P along the reduction
// and Q along the output columns in its four workgroup shapes, and the streaming short-K kernel of the first convolution.
// Launchers declared in launch.h; the plans that call them are in plan_*.inl (engine.hip).
#ifdef GRL_HOSTEMU
#include "hostemu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <cstdio>
#include <cstdlib>
#define GRL_ELEM_TYPES_ONLY     // (the element-wise kernels are compiled in engine.hip)
#define GRL_HEADS_TYPES_ONLY
#include "igemm_sk.h"
#include "launch.h"

namespace grl {

void launch_igemm2_fwd(int key, int n_tiles, hipStream_t s, const IgemmProb* probs, const int4* tiles, const char* tag) {
  const dim3 grid(n_tiles), block(256);
#define GRL_I2(PMv, QMv, CF, FL) \
  hipLaunchKernelGGL((igemm2_kernel<I2_P_ALONG_R, I2_Q_ALONG_J, PMv, QMv, CF, FL>), grid, block, 0, s, probs, tiles)
#define GRL_I2_CFGS(base, PMv, QMv, FL)                        \
  case base + 0 + FL: GRL_I2(PMv, QMv, 0, FL); break;          \
  case base + 10 + FL: GRL_I2(PMv, QMv, 1, FL); break;         \
  case base + 20 + FL: GRL_I2(PMv, QMv, 2, FL); break;         \
  case base + 30 + FL: GRL_I2(PMv, QMv, 3, FL); break;
  switch (key) {
    GRL_I2_CFGS(0, PM_AFFINE, QM_AFFINE, 0)             // dense forward
    GRL_I2_CFGS(0, PM_AFFINE, QM_AFFINE, I2F_KTAIL)     //   ... K % 4 != 0
    GRL_I2_CFGS(1000, PM_TABLE, QM_AFFINE, 0)           // VALID conv forward
    GRL_I2_CFGS(2000, PM_TABLE_MASK, QM_AFFINE, 0)      // padded conv forward
    default:
      fprintf(stderr, "grl: no igemm2 forward instantiation for launch '%s' (key %d)\n", tag, key);
      abort();
  }
#undef GRL_I2_CFGS
#undef GRL_I2
}

// one / two image channels (depth with the augmented extractor; nature_cnn on depth + pad channel).  RGB-D (four channels) keeps
// one launch per layer: its first layer cannot stage the image in LDS, and with the 4-byte patch loads the k-order demands the
// stack measured 3 657 against 4 156 updates/s (profiles/r05_bench_sac_rgbd_*.json)
bool conv_stack_ok(int C) { return C == 1 || C == 2 || C == 4; }
void launch_conv_stack_fwd(int C, const ConvStackArgs& a, hipStream_t s) {
  const dim3 grid(a.B * a.n_nets), block(256);
  if (C == 1) hipLaunchKernelGGL((conv_stack_fwd_kernel<1>), grid, block, 0, s, a);
  else if (C == 2) hipLaunchKernelGGL((conv_stack_fwd_kernel<2>), grid, block, 0, s, a);
  else if (C == 4) hipLaunchKernelGGL((conv_stack_fwd_kernel<4>), grid, block, 0, s, a);
  else {
    fprintf(stderr, "grl: no conv-stack instantiation for %d image channels\n", C);
    abort();
  }
}

void launch_conv_stack_bwd(const ConvStackBwdArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((conv_stack_bwd_kernel<0>), dim3(a.B * a.n_nets), dim3(256), 0, s, a);
}

void launch_igemm_sk(int K, int n_tiles, hipStream_t s, const IgemmProb* probs, const int4* work) {
  if (K == 64) hipLaunchKernelGGL((igemm_sk_kernel<64>), dim3(n_tiles), dim3(256), 0, s, probs, work);
  else hipLaunchKernelGGL((igemm_sk_kernel<32>), dim3(n_tiles), dim3(256), 0, s, probs, work);
}

}  // namespace grl
