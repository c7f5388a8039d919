// igemm2_ref1.h -- TEST-ONLY reference form of the kernels of csrc/igemm2.h (sequential loops over the same descriptors),
// included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).
// Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.
// TEST-ONLY reference of the v2 tile semantics (see hostemu.h): tile = BM x BN of CFG; the ones row
// (p_ones_i == M-1) is produced by row tile 0 and excluded from the row tiling.
template <int PL, int QL, int PM, int QM, int CFG, int FLAGS>
void igemm2_tile(const IgemmProb* probs, const int4 tl) {   // probs: THIS tile's descriptor copy
  if (threadIdx.x != 0) return;
  if (((FLAGS & 1) != 0) != (probs[0].p_ones_i >= 0)) abort();
  const int BM = i2_bm(CFG), BN = i2_bn(CFG);
  const IgemmProb& pb = probs[0];
  if ((PM != PM_AFFINE) != (pb.p_tab_i != nullptr && pb.p_tab_r != nullptr)) abort();
  if ((PM == PM_TABLE_MASK) != (pb.p_vmask_i != nullptr)) abort();
  if ((QM == QM_TABLE) != (pb.q_tab_r != nullptr)) abort();
  if (pb.p_k0 < pb.K) abort();
  if (pb.p_ones_i >= 0 && pb.p_ones_i != pb.M - 1) abort();
  const int Meff = pb.p_ones_i >= 0 ? pb.M - 1 : pb.M;
  const int r_begin = tl.y * pb.k_chunk, r_end = std::min(pb.K, r_begin + pb.k_chunk);
  float* cbase = pb.c + (long)tl.y * pb.slab_stride;
  auto row = [&](int i) {
    for (int j = tl.w * BN; j < std::min(pb.N, tl.w * BN + BN); ++j) {
      float acc = 0.f;
      for (int r = r_begin; r < r_end; ++r) {
        float pv = 0.f;
        bool ok = true;
        if (pb.p_vmask_i && i != pb.p_ones_i)
          ok = pb.p_mask_swap ? ((pb.p_vmask_i[r] >> pb.p_tap_r[i]) & 1ull) : ((pb.p_vmask_i[i] >> pb.p_tap_r[r]) & 1ull);
        if (i == pb.p_ones_i) pv = 1.f;
        else if (ok) {
          long rowterm = pb.p_tab_i ? pb.p_tab_i[i] : (long)i * pb.p_ld_i[0];
          long colterm = pb.p_tab_r ? pb.p_tab_r[r] : (long)r * pb.p_ld_r[0];
          pv = pb.p_base[0][rowterm + colterm];
        }
        long qrow = pb.q_tab_r ? pb.q_tab_r[r] : (long)r * pb.q_ld_r[0];
        acc = fmaf(pv, pb.q_base[0][qrow + (long)j * pb.q_ld_j[0]], acc);
      }
      long off;
      if (pb.c_tab_i) {
        const int ct = pb.c_tab_i[i];
        if (ct < 0) continue;
        off = (long)ct + j;
      } else off = (long)i * pb.ldc + j;
      float v = acc * (pb.out_scale != 0.f ? pb.out_scale : 1.f) + (pb.bias ? pb.bias[j] : 0.f);
      if (pb.accumulate) v += cbase[off];
      if (pb.act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (pb.act == ACT_LEAKY) v = v > 0.f ? v : pb.act_alpha * v;
      if (pb.relu_mask) v = pb.relu_mask[pb.m_tab_i ? (long)pb.m_tab_i[i] + j : off] > 0.f ? v : pb.act_alpha * v;
      cbase[off] = v;
    }
  };
  for (int i = tl.z * BM; i < std::min(Meff, tl.z * BM + BM); ++i) row(i);
  if (pb.p_ones_i >= 0 && tl.z == 0) row(pb.p_ones_i);
}
template <int PL, int QL, int PM, int QM, int CFG, int FLAGS>
void igemm2_kernel(const IgemmProb* probs, const int4* tiles) {
  igemm2_tile<PL, QL, PM, QM, CFG, FLAGS>(probs + blockIdx.x, tiles[blockIdx.x]);
}
// two kinds of tiles in one launch: blocks [0, n_a) run kind A (the launch's own stage), the rest kind B (fillers)
template <int PLa, int QLa, int PMa, int QMa, int CFGa, int FLa, int PLb, int QLb, int PMb, int QMb, int CFGb, int FLb>
void igemm2_pair_kernel(const IgemmProb* pa, const int4* ta, int n_a, const IgemmProb* pb, const int4* tb) {
  const int b = (int)blockIdx.x;
  const bool is_a = b < n_a;
  const int k = is_a ? b : b - n_a;
  if (is_a) igemm2_tile<PLa, QLa, PMa, QMa, CFGa, FLa>(pa + k, ta[k]);
  else igemm2_tile<PLb, QLb, PMb, QMb, CFGb, FLb>(pb + k, tb[k]);
}
