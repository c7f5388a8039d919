// launch.h -- the seam between the translation units of libgrl.so.  engine.hip (host logic: parameter layout, addressing
// tables, problem builders, launch plans, the C ABI, plus the element-wise / replay / exchange kernels) sees the GEMM and
// head kernels only through these launchers; the kernel templates are instantiated in
//   gemm_fwd.hip    igemm2_kernel with P along the reduction, Q along the output columns (forward products), igemm_sk_kernel
//   gemm_bwd.hip    igemm2_kernel with both operands along the reduction (backward-data), igemm2_pair_kernel
//   gemm_wgrad.hip  igemm2_kernel with P along the output rows (weight gradients), the scalar-gather igemm_kernel
//   heads.hip       heads_fused_kernel (MFMA), the VALU head chains, the DQN / BDQ tower chains (q_mfma.h on the matrix cores, q_kernels.h VALU)
// so that the four compile side by side (the single translation unit of round 3 took 62 s).
#pragma once
#include "igemm.h"
#include "igemm2.h"
#include "heads_kernels.h"
#include "heads_mfma.h"
#include "q_kernels.h"
#include "q_mfma.h"
#include "q_chain.h"
#include "act_mfma.h"
#include "conv_stack.h"

namespace grl {

// igemm2 instantiation key: variant * 10000 + pm * 1000 + qm * 100 + cfg * 10 + flags (engine.hip, v2_key)
void launch_igemm2_fwd(int key, int n_tiles, hipStream_t s, const IgemmProb* probs, const int4* tiles, const char* tag);     // variant 0
void launch_igemm2_bwd(int key, int n_tiles, hipStream_t s, const IgemmProb* probs, const int4* tiles, const char* tag);     // variant 1
void launch_igemm2_wgrad(int key, int n_tiles, hipStream_t s, const IgemmProb* probs, const int4* tiles, const char* tag);   // variant 2
// a backward-data stage (key ka, n_a tiles) carrying n_b weight-gradient tiles (key 21001) behind its own
void launch_igemm2_pair(int ka, int n_a, int n_b, hipStream_t s, const IgemmProb* pa, const int4* ta, const IgemmProb* pb, const int4* tb,
                        const char* tag);
// scalar-gather fallback: key = np * 1000 + pm * 100 + qm * 10 + variant
void launch_igemm(int key, int n_tiles, hipStream_t s, const IgemmProb* probs, const int4* tiles, const char* tag);
void launch_igemm_sk(int K, int n_tiles, hipStream_t s, const IgemmProb* probs, const int4* work);
// the three convolutions of the extractor as one sample-local launch (conv_stack.h); C = image channels, 1 / 2 / 4
bool conv_stack_ok(int C);
void launch_conv_stack_fwd(int C, const ConvStackArgs& a, hipStream_t s);
void launch_conv_stack_bwd(const ConvStackBwdArgs& a, hipStream_t s);

// What THIS build of the kernels offers -- answered where the kernels are instantiated, so that the launch plans (plan_*.inl)
// carry no build switches of their own: the g++ emulation build (tests only) has sequential reference forms for the VALU chains
// and the scalar element-wise kernels, but none for the matrix-core chains or the 16-byte element-wise variants.
bool q_mfma_built();                 // q_mfma.h: tower / trunk chains on 16x16x4 MFMA stages
bool q_chain_available(bool mfma, bool fits);   // q_chain.h: backward chains that form the loss and the weight gradients of their rows
bool act_mfma_built();               // act_mfma.h: policy head of grl_act on MFMA stages
int device_lds_bytes();              // shared memory a workgroup may declare on the current device (emulation: no limit)
enum { HEADS_GENERAL_64 = 0, HEADS_FAST_64 = 1, HEADS_FAST_128 = 2 };
// ride: the next update's image gather as extra workgroups (NULL: none); ride_gx tiles per row
void launch_heads_fused(int shape, int nblk, hipStream_t s, const HeadsFusedArgs* args, const GatherArgs* ride = nullptr, int ride_gx = 0);
size_t heads_fused_lds_bytes(int shape);
void launch_heads_fwd(const HeadsFwdArgs& a, hipStream_t s);
void launch_heads_bwd(const HeadsBwdArgs& a, hipStream_t s);
void launch_act_heads_mfma(const ActHeadsArgs& a, hipStream_t s);     // one workgroup per 16 observations (act_mfma.h)
void launch_q_fwd(const QFusedArgs& a, hipStream_t s);
void launch_q_bwd(const QFusedArgs& a, hipStream_t s);     // towers, then the trunk (when there is one)
void launch_q_bwd_chain(const QChainArgs& a, hipStream_t s);     // the same with loss + weight gradients inside (q_chain.h)

}  // namespace grl
