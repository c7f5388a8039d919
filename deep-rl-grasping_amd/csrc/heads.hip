// heads.hip -- the row-local chains: every MLP head of the SAC update forward + backward on the 16x16x4 matrix cores
// (heads_mfma.h), the two-launch VALU chains other widths fall back to (heads_kernels.h), and the DQN / BDQ towers
// (q_kernels.h).  Launchers declared in launch.h.
#ifdef GRL_HOSTEMU
#include "hostemu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define GRL_ELEM_TYPES_ONLY     // (the element-wise kernels are compiled in engine.hip)
#include "launch.h"

namespace grl {

void launch_heads_fused(int shape, int nblk, hipStream_t s, const HeadsFusedArgs* args, const GatherArgs* ride, int ride_gx) {
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  int n_ride = 0;
  if (ride) { g = *ride; n_ride = gather_rider_blocks(g, ride_gx); }
  const dim3 grid(nblk, 4 + (n_ride + nblk - 1) / nblk);
  if (shape == HEADS_FAST_128) hipLaunchKernelGGL((heads_fused_kernel<128, true>), grid, dim3(256), 0, s, args, g, ride_gx, n_ride);
  else if (shape == HEADS_FAST_64) hipLaunchKernelGGL((heads_fused_kernel<64, true>), grid, dim3(256), 0, s, args, g, ride_gx, n_ride);
  else hipLaunchKernelGGL((heads_fused_kernel<64, false>), grid, dim3(256), 0, s, args, g, ride_gx, n_ride);
}
bool q_mfma_built() {
#ifdef GRL_HOSTEMU
  return false;
#else
  return true;
#endif
}
// the backward chains that form the loss and the weight gradients of their rows (q_chain.h): on the device they are built from
// the matrix-core stages, so they go with them (mfma); the emulation has a sequential reference form for every shape that
// fits those stages (fits), whichever forward kernels the plan took
bool q_chain_available(bool mfma, bool fits) {
#ifdef GRL_HOSTEMU
  (void)mfma;
  return fits;
#else
  (void)fits;
  return mfma;
#endif
}
bool act_mfma_built() { return q_mfma_built(); }
int device_lds_bytes() {
#ifdef GRL_HOSTEMU
  return 1 << 30;
#else
  int dev = 0, lds = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return 0;
  return lds;
#endif
}
size_t heads_fused_lds_bytes(int shape) {
#ifdef GRL_HOSTEMU
  (void)shape;
  return 0;
#else
  return shape == HEADS_FAST_128 ? sizeof(HmLds<128>) : sizeof(HmLds<64>);
#endif
}
void launch_heads_fwd(const HeadsFwdArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(heads_fwd_kernel, dim3((a.B + HT_RB - 1) / HT_RB, 6), dim3(256), 0, s, a);
}
void launch_heads_bwd(const HeadsBwdArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(heads_bwd_kernel, dim3((a.B + HT_RB - 1) / HT_RB, 4), dim3(256), 0, s, a);
}
void launch_act_heads_mfma(const ActHeadsArgs& a, hipStream_t s) {
#ifndef GRL_HOSTEMU
  hipLaunchKernelGGL(act_heads_mfma_kernel, dim3((a.rows + HT_RB - 1) / HT_RB), dim3(256), 0, s, a);
#else
  (void)a; (void)s;
#endif
}
void launch_q_fwd(const QFusedArgs& a, hipStream_t s) {
  const dim3 grid((a.B + HT_RB - 1) / HT_RB, 3, a.D + 1);
#ifndef GRL_HOSTEMU
  if (a.mfma) { hipLaunchKernelGGL(q_fwd_mfma_kernel, grid, dim3(256), 0, s, a); return; }
#endif
  hipLaunchKernelGGL(q_fwd_fused_kernel, grid, dim3(256), 0, s, a);
}
void launch_q_bwd(const QFusedArgs& a, hipStream_t s) {
  const dim3 towers((a.B + HT_RB - 1) / HT_RB, a.D + 1), trunk((a.B + HT_RB - 1) / HT_RB);
#ifndef GRL_HOSTEMU
  if (a.mfma) {
    hipLaunchKernelGGL(q_bwd_towers_mfma_kernel, towers, dim3(256), 0, s, a);
    if (a.bwd_tr) hipLaunchKernelGGL(q_bwd_trunk_mfma_kernel, trunk, dim3(256), 0, s, a);
    return;
  }
#endif
  hipLaunchKernelGGL(q_bwd_towers_kernel, towers, dim3(256), 0, s, a);
  if (a.bwd_tr) hipLaunchKernelGGL(q_bwd_trunk_kernel, trunk, dim3(256), 0, s, a);
}
void launch_q_bwd_chain(const QChainArgs& a, hipStream_t s) {
  const dim3 towers((a.f.B + HT_RB - 1) / HT_RB, a.f.D + 1), trunk((a.f.B + HT_RB - 1) / HT_RB, qc_trunk_rows(a));
  hipLaunchKernelGGL(q_bwd_towers_chain_kernel, towers, dim3(256), 0, s, a);
  if (a.f.bwd_tr) hipLaunchKernelGGL(q_bwd_trunk_chain_kernel, trunk, dim3(256), 0, s, a);
}

}  // namespace grl
