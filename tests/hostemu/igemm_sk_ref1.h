// igemm_sk_ref1.h -- TEST-ONLY reference form of the kernels of csrc/igemm_sk.h (sequential loops over the same descriptors),
// included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).
// Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.
// TEST-ONLY reference: work item {problem, first tile, tiles}; tile = 128 rows, all columns
template <int K>
void igemm_sk_kernel(const IgemmProb* probs, const int4* work) {
  if (threadIdx.x != 0) return;
  const int4 wk = work[blockIdx.x];
  const IgemmProb& pb = probs[blockIdx.x];   // per-workgroup descriptor copy
  if (pb.K != K || pb.N > 32 || !pb.p_tab_i || pb.p_vmask_i || pb.q_tab_r || pb.c_tab_i || pb.relu_mask || pb.accumulate ||
      pb.split != 1)
    abort();
  for (int i = wk.y * 128; i < std::min(pb.M, (wk.y + wk.z) * 128); ++i)
    for (int j = 0; j < pb.N; ++j) {
      float acc = 0.f;
      for (int r = 0; r < K; ++r)
        acc = fmaf(pb.p_base[0][(long)pb.p_tab_i[i] + pb.p_tab_r[r]], pb.q_base[0][(long)r * pb.q_ld_r[0] + j], acc);
      float v = acc * (pb.out_scale != 0.f ? pb.out_scale : 1.f) + (pb.bias ? pb.bias[j] : 0.f);
      if (pb.act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (pb.act == ACT_LEAKY) v = v > 0.f ? v : pb.act_alpha * v;
      pb.c[(long)i * pb.ldc + j] = v;
    }
}
