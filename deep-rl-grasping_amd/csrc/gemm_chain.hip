// gemm_chain.hip -- dependent GEMM stages in ONE launch (igemm2.h: igemm2_chain_kernel; the host side that derives the tile
// dependencies is engine.hip: chain_ops): conv3_fwd -> fc_fwd -> heads_l0 and heads_dfeat -> fc_bwd of the SAC update.
// Producer stages store write-through (I2F_WT), consumer stages load their P operand at system scope (I2F_PSYS).
// Launcher declared in launch.h.
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

void launch_igemm2_chain(int kind, const ChainArgs& ca, int n_blocks, hipStream_t s) {
  const dim3 grid(n_blocks), block(256);
  if (kind == CHAIN_FWD3)
    hipLaunchKernelGGL((igemm2_chain_kernel<I2_P_ALONG_R, I2_Q_ALONG_J, PM_TABLE, QM_AFFINE, 3, I2F_WT,
                                             I2_P_ALONG_R, I2_Q_ALONG_J, PM_AFFINE, QM_AFFINE, 3, I2F_WT | I2F_PSYS,
                                             I2_P_ALONG_R, I2_Q_ALONG_J, PM_AFFINE, QM_AFFINE, 3, I2F_PSYS>),
                       grid, block, 0, s, ca);
  else if (kind == CHAIN_FWD3_KTAIL)
    hipLaunchKernelGGL((igemm2_chain_kernel<I2_P_ALONG_R, I2_Q_ALONG_J, PM_TABLE, QM_AFFINE, 3, I2F_WT,
                                             I2_P_ALONG_R, I2_Q_ALONG_J, PM_AFFINE, QM_AFFINE, 3, I2F_WT | I2F_PSYS,
                                             I2_P_ALONG_R, I2_Q_ALONG_J, PM_AFFINE, QM_AFFINE, 3, I2F_KTAIL | I2F_PSYS>),
                       grid, block, 0, s, ca);
  else
    hipLaunchKernelGGL((igemm2_chain_kernel<I2_P_ALONG_R, I2_Q_ALONG_R, PM_AFFINE, QM_TABLE, 3, I2F_WT,
                                             I2_P_ALONG_R, I2_Q_ALONG_R, PM_AFFINE, QM_AFFINE, 3, I2F_PSYS,
                                             I2_P_ALONG_R, I2_Q_ALONG_R, PM_AFFINE, QM_AFFINE, 3, I2F_PSYS>),
                       grid, block, 0, s, ca);
}

}  // namespace grl
