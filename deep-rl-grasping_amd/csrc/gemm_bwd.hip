// gemm_bwd.hip -- backward-data products (dense layers, convolutions over exact taps or with validity masks): igemm2_kernel
// with both operands along the reduction, and igemm2_pair_kernel (conv3_bwd carrying dense weight-gradient tiles in the
// slots its last dispatch round leaves empty).  Launchers declared in launch.h.
#ifdef GRL_HOSTEMU
#include "hostemu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <cstdio>
#include <cstdlib>
#define GRL_ELEM_TYPES_ONLY     // (the element-wise kernels are compiled in engine.hip)
#define GRL_HEADS_TYPES_ONLY
#include "launch.h"

namespace grl {

void launch_igemm2_bwd(int key, int n_tiles, hipStream_t s, const IgemmProb* probs, const int4* tiles, const char* tag) {
  const dim3 grid(n_tiles), block(256);
#define GRL_I2(PMv, QMv, CF) \
  hipLaunchKernelGGL((igemm2_kernel<I2_P_ALONG_R, I2_Q_ALONG_R, PMv, QMv, CF, 0>), grid, block, 0, s, probs, tiles)
#define GRL_I2_CFGS(base, PMv, QMv)                    \
  case base + 0: GRL_I2(PMv, QMv, 0); break;           \
  case base + 10: GRL_I2(PMv, QMv, 1); break;          \
  case base + 20: GRL_I2(PMv, QMv, 2); break;          \
  case base + 30: GRL_I2(PMv, QMv, 3); break;
  switch (key) {
    GRL_I2_CFGS(10000, PM_AFFINE, QM_AFFINE)        // dense backward-data
    GRL_I2_CFGS(10100, PM_AFFINE, QM_TABLE)         // dense backward-data over several kernels
    GRL_I2_CFGS(11100, PM_TABLE, QM_TABLE)          // conv backward-data, exact taps
    GRL_I2_CFGS(12100, PM_TABLE_MASK, QM_TABLE)     // conv backward-data, masked taps (padded convolutions)
    default:
      fprintf(stderr, "grl: no igemm2 backward-data instantiation for launch '%s' (key %d)\n", tag, key);
      abort();
  }
#undef GRL_I2_CFGS
#undef GRL_I2
}

void launch_igemm2_pair(int ka, int n_a, int n_b, hipStream_t s, const IgemmProb* pa, const int4* ta, const IgemmProb* pb, const int4* tb,
                        const char* tag) {
  const dim3 grid(n_a + n_b), block(256);
  switch (ka) {
    case 11130:
      hipLaunchKernelGGL((igemm2_pair_kernel<I2_P_ALONG_R, I2_Q_ALONG_R, PM_TABLE, QM_TABLE, 3, 0, I2_P_ALONG_I, I2_Q_ALONG_J, PM_TABLE, QM_AFFINE, 0, I2F_ONES>),
                         grid, block, 0, s, pa, ta, n_a, pb, tb);
      break;
    default:
      fprintf(stderr, "grl: no igemm2 pair instantiation for launch '%s' (key %d)\n", tag, ka);
      abort();
  }
}

}  // namespace grl
