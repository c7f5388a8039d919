// gemm_wgrad.hip -- weight-gradient products (P along the output rows: x^T or gathered patches^T times dY, bias gradient as
// the ones row) on igemm2_kernel, and igemm_kernel, the scalar-gather form every problem the vectorised kernel cannot take
// falls back to (unaligned operands, multi-part inputs).  Launchers declared in launch.h.
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

void launch_igemm2_wgrad(int key, int n_tiles, hipStream_t s, const IgemmProb* probs, const int4* tiles, const char* tag) {
  const dim3 grid(n_tiles), block(256);
#define GRL_I2(PMv, CF, FL) \
  hipLaunchKernelGGL((igemm2_kernel<I2_P_ALONG_I, I2_Q_ALONG_J, PMv, QM_AFFINE, CF, FL>), grid, block, 0, s, probs, tiles)
  switch (key) {
    case 20000: GRL_I2(PM_AFFINE, 0, 0); break;            // dense weight gradient
    case 20001: GRL_I2(PM_AFFINE, 0, I2F_ONES); break;     //   ... with bias row
    case 20010: GRL_I2(PM_AFFINE, 1, 0); break;
    case 20011: GRL_I2(PM_AFFINE, 1, I2F_ONES); break;
    case 21000: GRL_I2(PM_TABLE, 0, 0); break;             // conv weight gradient
    case 21001: GRL_I2(PM_TABLE, 0, I2F_ONES); break;
    case 21010: GRL_I2(PM_TABLE, 1, 0); break;
    case 21011: GRL_I2(PM_TABLE, 1, I2F_ONES); break;
    default:
      fprintf(stderr, "grl: no igemm2 weight-gradient instantiation for launch '%s' (key %d)\n", tag, key);
      abort();
  }
#undef GRL_I2
}

void launch_igemm(int key, int n_tiles, hipStream_t s, const IgemmProb* probs, const int4* tiles, const char* tag) {
  const dim3 grid(n_tiles), block(256);
#define GRL_IGEMM(PMv, QMv, PR, QJ, NPv) hipLaunchKernelGGL((igemm_kernel<PMv, QMv, PR, QJ, NPv>), grid, block, 0, s, probs, tiles)
  switch (key) {
    case 1000: GRL_IGEMM(PM_AFFINE, QM_AFFINE, true, true, 1); break;        // dense forward
    case 3000: GRL_IGEMM(PM_AFFINE, QM_AFFINE, true, true, 3); break;        //   ... concatenated input
    case 1100: GRL_IGEMM(PM_TABLE, QM_AFFINE, true, true, 1); break;         // VALID conv forward
    case 1200: GRL_IGEMM(PM_TABLE_MASK, QM_AFFINE, true, true, 1); break;    // padded conv forward
    case 1001: GRL_IGEMM(PM_AFFINE, QM_AFFINE, true, false, 1); break;       // dense backward-data
    case 3001: GRL_IGEMM(PM_AFFINE, QM_AFFINE, true, false, 3); break;       //   ... summed over heads
    case 1211: GRL_IGEMM(PM_TABLE_MASK, QM_TABLE, true, false, 1); break;    // conv backward-data
    case 1111: GRL_IGEMM(PM_TABLE, QM_TABLE, true, false, 1); break;         //   ... over exact taps
    case 1011: GRL_IGEMM(PM_AFFINE, QM_TABLE, true, false, 1); break;        // dense backward-data over several kernels
    case 1002: GRL_IGEMM(PM_AFFINE, QM_AFFINE, false, true, 1); break;       // dense weight gradient
    case 1102: GRL_IGEMM(PM_TABLE, QM_AFFINE, false, true, 1); break;        // conv weight gradient
    case 1202: GRL_IGEMM(PM_TABLE_MASK, QM_AFFINE, false, true, 1); break;   //   ... of a padded conv
    default:
      fprintf(stderr, "grl: no igemm instantiation for launch '%s' (key %d)\n", tag, key);
      abort();
  }
#undef GRL_IGEMM
}

}  // namespace grl
