// elem_kernels_ref2.h -- TEST-ONLY reference form of the kernels of csrc/elem_kernels.h (sequential loops over the same descriptors),
// included by that header ONLY in the g++ emulation build (-DGRL_HOSTEMU -I tests/hostemu, tests/conftest.py).
// Never part of libgrl.so.  No include guard: it is pasted once, inside namespace grl.
inline void q_loss_kernel(QLossArgs a) {
  if (threadIdx.x != 0) return;
  float s3[3] = {0, 0, 0};
  for (int b = 0; b < a.B; ++b) {
    float r3[3] = {0, 0, 0};
    q_loss_row(a, b, r3);
    for (int k = 0; k < 3; ++k) { a.row_part[3 * b + k] = r3[k]; s3[k] += r3[k]; }
  }
  if (!a.defer_finish) q_loss_finish(a, s3[0], s3[1], s3[2]);   // deferred: the launch that applies the update sums row_part
}
