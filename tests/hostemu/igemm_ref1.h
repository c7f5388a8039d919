// igemm_ref1.h -- TEST-ONLY reference form of the kernels of csrc/igemm.h (sequential loops over the same descriptors),
// included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).
// Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.
// TEST-ONLY reference of the descriptor semantics (see hostemu.h); one call computes one tile.
template <int PM, int QM, bool P_CONTIG_R, bool Q_CONTIG_J, int NP>
void igemm_kernel(const IgemmProb* probs, const int4* tiles) {
  if (threadIdx.x != 0) return;
  const int4 tl = tiles[blockIdx.x];
  const IgemmProb& pb = probs[blockIdx.x];   // per-workgroup descriptor copy
  // the launch-time mode must agree with what the descriptor carries
  if ((PM != PM_AFFINE) != (pb.p_tab_i != nullptr && pb.p_tab_r != nullptr)) abort();
  if ((PM == PM_TABLE_MASK) != (pb.p_vmask_i != nullptr)) abort();
  if ((QM == QM_TABLE) != (pb.q_tab_r != nullptr)) abort();
  if (NP == 1 && pb.p_k0 < pb.K) abort();   // single-part instantiation given a multi-part problem
  const int r_begin = tl.y * pb.k_chunk, r_end = std::min(pb.K, r_begin + pb.k_chunk);
  float* cbase = pb.c + (long)tl.y * pb.slab_stride;
  for (int i = tl.z * 64; i < std::min(pb.M, tl.z * 64 + 64); ++i)
    for (int j = tl.w * 64; j < std::min(pb.N, tl.w * 64 + 64); ++j) {
      float acc = 0.f;
      for (int r = r_begin; r < r_end; ++r) {
        const int part = (r >= pb.p_k0) + (r >= pb.p_k1);
        const int rstart = part == 0 ? 0 : (part == 1 ? pb.p_k0 : pb.p_k1);
        float pv = 0.f;
        bool ok = true;
        if (pb.p_vmask_i && i != pb.p_ones_i)
          ok = pb.p_mask_swap ? ((pb.p_vmask_i[r] >> pb.p_tap_r[i]) & 1ull) : ((pb.p_vmask_i[i] >> pb.p_tap_r[r]) & 1ull);
        if (i == pb.p_ones_i) pv = 1.f;
        else if (ok) {
          const long rowterm = pb.p_tab_i ? pb.p_tab_i[i] : (long)i * pb.p_ld_i[part];
          const long colterm = pb.p_tab_r ? pb.p_tab_r[r] : (long)(r - rstart) * pb.p_ld_r[part];
          pv = pb.p_base[part][rowterm + colterm];
        }
        const long qrow = pb.q_tab_r ? pb.q_tab_r[r] : (long)(r - rstart) * pb.q_ld_r[part];
        const float qv = pb.q_base[part][qrow + (long)j * pb.q_ld_j[part]];
        acc = fmaf(pv, qv, acc);
      }
      long off;
      if (pb.c_tab_i) {
        if (pb.c_tab_i[i] < 0) continue;
        off = (long)pb.c_tab_i[i] + j;
      } else off = (long)i * pb.ldc + j;
      float v = acc * (pb.out_scale != 0.f ? pb.out_scale : 1.f) + (pb.bias ? pb.bias[j] : 0.f);
      if (pb.accumulate) v += cbase[off];
      if (pb.act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (pb.act == ACT_LEAKY) v = v > 0.f ? v : pb.act_alpha * v;
      if (pb.relu_mask) v = pb.relu_mask[pb.m_tab_i ? (long)pb.m_tab_i[i] + j : off] > 0.f ? v : pb.act_alpha * v;   // ReLU (alpha 0) / LeakyReLU gradient
      cbase[off] = v;
    }
}
