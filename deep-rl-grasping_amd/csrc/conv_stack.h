// conv_stack.h -- the convolution stack of the feature extractor as ONE sample-local launch.
//
// /root/reference/manipulation_main/training/custom_obs_policy.py:34-37 (and stable-baselines' nature_cnn): three VALID
// convolutions with bias + ReLU, 64x64xC -> 8x8 s4 -> 15x15x32 -> 4x4 s2 -> 6x6x64 -> 3x3 s1 -> 4x4x64, then the NHWC
// flatten that feeds cnn_fc1.  Per sample the intermediate activations are 28.8 KB + 9.2 KB + 4 KB and every layer has
// N <= 64 output channels, so ONE workgroup can own a whole sample of one network and keep the chain in its LDS: the
// hand-overs between the layers are __syncthreads(), not launches, and conv2 / conv3 never read their input from memory
// (round 4 wrote 22 MB of layer-1 activations per update and read them straight back).  Networks whose activations the
// backward pass needs (model/pi, model/values_fn) still store them -- as 16-byte rows from LDS while the next layer
// computes; the target network stores only its last layer.
//
// Work decomposition (f32 matrix cores, v_mfma_f32_16x16x4_f32, exact fp32):
//   * grid = networks x batch samples, 256 threads.  42 KB of LDS per workgroup: three workgroups share a CU and hide each
//     other's load / barrier latencies (3 x B = 768 workgroups at B = 256 = exactly three per CU);
//   * the GEMM rows are the output pixels of THIS sample (225 / 36 / 16: 15 / 3 / 1 row tiles of 16), the columns the output
//     channels (two / four / four column tiles of 16);
//   * A operand (patches): conv1 straight from the minibatch tensor -- a lane's four reduction steps are four neighbouring
//     pixels of one patch row (C = 1) or the four channels of one pixel (C = 4): one 16-byte load; conv2 / conv3 from the
//     activation tile in LDS (pixel stride padded to 36 / 68 floats: 16-byte reads, at most 2-way bank conflicts);
//   * B operand (kernels, TF's HWIO layout untouched): global memory -> registers, prefetched two reduction chunks ahead;
//     every workgroup of a network reads the same 283 KB, which stay in L2 (and, for the three workgroups of a CU, in L1);
//   * reduction order: chunks of 16 in increasing k; inside a chunk the MFMA's own order.  conv3 (one row tile per wave)
//     keeps four independent accumulators -- one per step of a chunk -- and adds them at the end: a single accumulator
//     would wait 40 cycles per 32-cycle MFMA.
// Results agree with the per-layer implicit-GEMM launches to fp32 rounding (different summation order), and with the
// oracle within the forward tolerance of tests/parity_util.py.
#pragma once
#include <stdint.h>

namespace grl {

struct ConvStackNet {
  const float* x;        // [B][64 * 64 * C]  normalised, scaled images of this network's minibatch side (obs or next_obs)
  const float* w[3];     // HWIO kernels: [8,8,C,32], [4,4,32,64], [3,3,64,64]
  const float* b[3];     // biases
  float* a1;             // [B * 225][ld1]  layer-1 activations (nullptr: not stored)
  float* a2;             // [B * 36][64]    layer-2 activations (nullptr: not stored)
  float* a3;             // [B * 16][64]    layer-3 activations = the flattened features' input
  int ld1;               // pixel stride of a1 (64: two networks side by side, plan_sac.inl)
  int pad;
};

struct ConvStackArgs {
  const ConvStackNet* nets;   // device array [n_nets]
  int B;
  int n_nets;
#ifdef CS_STAMPS
  unsigned long long* stamps;
#endif
};

enum { CS_P1 = 36, CS_P2 = 68, CS_LDS_FLOATS = 225 * CS_P1 + 36 * CS_P2 };

#ifdef GRL_HOSTEMU
#include "conv_stack_ref1.h"   // tests/hostemu: the emulation build only
#else

typedef float cs_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t cs_rsrc(const float* p) {
  // descriptor inputs made provably wave-uniform (otherwise every buffer load sits in a waterfall loop)
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float cs_ld(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
// the four reduction steps of chunk j (k = 16 j + 4 q + i) of column n of a [K][ldw] kernel matrix
__device__ __forceinline__ void cs_load_b(float (&bw)[4], __amdgpu_buffer_rsrc_t rs, int voff, int j, int ldw) {
#pragma unroll
  for (int i = 0; i < 4; ++i) bw[i] = cs_ld(rs, voff + i * ldw * 4, j * 16 * ldw * 4);
}

// block -> (network, sample).  With a whole number of workgroups per CU the three (..) workgroups that share a CU -- block b
// runs on CU b % 256 (observed placement, used for speed only) -- belong to the same network and read the same kernels.
__device__ __forceinline__ void cs_unit(const ConvStackArgs& a, int& net, int& smp) {
  const int n_units = a.B * a.n_nets;
  int j = blockIdx.x;
  if ((n_units & 255) == 0) {
    const int slots = n_units >> 8;
    j = (blockIdx.x & 255) * slots + (blockIdx.x >> 8);
  }
  net = j / a.B;
  smp = j - net * a.B;
}

// measurement build (scripts/conv_stack_bench.hip -DCS_STAMPS): wave 0 of every workgroup records the shader clock at its
// phase boundaries
#ifdef CS_STAMPS
#define CS_STAMP(k) do { if (a.stamps && t == 0) a.stamps[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define CS_WALL(k) do { if (a.stamps && t == 0) a.stamps[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)     // 100 MHz, chip-wide
#else
#define CS_STAMP(k) do { } while (0)
#define CS_WALL(k) do { } while (0)
#endif
#ifndef CS_WG_PER_CU
#define CS_WG_PER_CU 3
#endif
#ifndef CS_PREFETCH
#define CS_PREFETCH 2
#endif

template <int C>
__global__ __launch_bounds__(256, CS_WG_PER_CU) void conv_stack_fwd_kernel(ConvStackArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[CS_LDS_FLOATS];
  float* const act1 = lds;
  float* const act2 = lds + 225 * CS_P1;
  int net_i, smp;
  cs_unit(a, net_i, smp);
  const ConvStackNet& net = a.nets[net_i];
  const int t = threadIdx.x, w = t >> 6, l = t & 63, c = l & 15, q = l >> 4;
  CS_STAMP(0);
  CS_WALL(8);
  const __amdgpu_buffer_rsrc_t rsw1 = cs_rsrc(net.w[0]), rsw2 = cs_rsrc(net.w[1]), rsw3 = cs_rsrc(net.w[2]);
  const int voff23 = ((4 * q) * 64 + 16 * w + c) * 4;      // conv2 / conv3: wave w owns output channels 16 w .. 16 w + 15
  float bq[CS_PREFETCH + 1][4];                            // conv2 / conv3 kernel chunks in flight: CS_PREFETCH ahead of the multiply

  // =========================================================================== conv1: 8x8 stride 4, C -> 32
  {
    constexpr int NJ = 4 * C;                       // reduction chunks of 16 (K = 64 C)
    constexpr bool RES = C == 1;                    // the whole kernel matrix stays in registers (32 per lane)
    const float* x = net.x + (int64_t)smp * (64 * 64 * C);
    // lane's share of chunk j: reduction steps k = 16 j + 4 q .. + 3 of the HWIO order k = (kh * 8 + kw) * C + ch -- four
    // neighbouring pixels of a patch row (C = 1), two pixels (C = 2) or the four channels of one pixel (C = 4): 16 bytes
    auto a_off = [&](int row, int j) -> int {
      const int r = row < 225 ? row : 224;
      const int oh = r / 15, ow = r - oh * 15;
      const int k0 = 16 * j + 4 * q, p = k0 / C, ch = k0 - p * C;
      return ((oh * 4 + (p >> 3)) * 64 + ow * 4 + (p & 7)) * C + ch;
    };
    const int voff1 = ((4 * q) * 32 + c) * 4;
    cs_f4 an[RES ? NJ : 1];
    float bw[RES ? NJ : 1][2][4];
    if (RES) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) an[RES ? j : 0] = *(const cs_f4*)(x + a_off(16 * w + c, j));
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) cs_load_b(bw[RES ? j : 0][nt], rsw1, voff1 + 64 * nt, j, 32);
    }
    const float bias0 = net.b[0][c], bias1 = net.b[0][16 + c];
    float* const a1g = net.a1 ? net.a1 + (int64_t)smp * 225 * net.ld1 : nullptr;
    for (int mt = w; mt < 15; mt += 4) {
      cs_f4 acc[2] = {cs_f4{0.f, 0.f, 0.f, 0.f}, cs_f4{0.f, 0.f, 0.f, 0.f}};
      if (RES) {
        cs_f4 ac[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) ac[j] = an[RES ? j : 0];
        if (mt + 4 < 15) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) an[RES ? j : 0] = *(const cs_f4*)(x + a_off(16 * (mt + 4) + c, j));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[j][i], bw[RES ? j : 0][nt][i], acc[nt], 0, 0, 0);
      } else {
        // K = 64 C: the kernel chunks come from L1 every row tile (128 registers would hold them for C = 4)
        cs_f4 av = *(const cs_f4*)(x + a_off(16 * mt + c, 0));
        float b0[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) cs_load_b(b0[nt], rsw1, voff1 + 64 * nt, 0, 32);
#pragma unroll 4
        for (int j = 0; j < NJ; ++j) {
          float b1[2][4];
          const int jn = j + 1 < NJ ? j + 1 : j;
          const cs_f4 avn = *(const cs_f4*)(x + a_off(16 * mt + c, jn));
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) cs_load_b(b1[nt], rsw1, voff1 + 64 * nt, jn, 32);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b0[nt][i], acc[nt], 0, 0, 0);
          av = avn;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) b0[nt][i] = b1[nt][i];
        }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * mt + 4 * q + v;
        if (row < 225) {
          act1[row * CS_P1 + c] = fmaxf(acc[0][v] + bias0, 0.f);
          act1[row * CS_P1 + 16 + c] = fmaxf(acc[1][v] + bias1, 0.f);
        }
      }
      if (a1g) {
        // this wave's 16 x 32 tile goes out as whole 128-byte rows (two 16-byte stores per lane) right away: a burst of all
        // 225 rows after the barrier queued 14 k cycles behind the other workgroups' stores (scripts/conv_stack_bench.hip
        // stamps).  LDS operations of one wave execute in order: the fence only keeps the compiler from moving the reads up.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = 16 * mt + 8 * h + (l >> 3), ch = l & 7;
          if (row < 225) *(cs_f4*)(a1g + (int64_t)row * net.ld1 + 4 * ch) = *(const cs_f4*)(act1 + row * CS_P1 + 4 * ch);
        }
      }
    }
  }
  // the first kernel chunks of conv2 do not depend on this workgroup's activations: requested before the barrier
#pragma unroll
  for (int j = 0; j < CS_PREFETCH; ++j) cs_load_b(bq[j], rsw2, voff23, j, 64);
  CS_STAMP(1);
  __syncthreads();
  CS_STAMP(2);

  // =========================================================================== conv2: 4x4 stride 2, 32 -> 64 (wave w: columns 16 w ..)
  {
    int rb[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      const int r = 16 * mt + c < 36 ? 16 * mt + c : 35;
      const int oh = r / 6, ow = r - oh * 6;
      rb[mt] = (oh * 2 * 15 + ow * 2) * CS_P1 + 4 * q;
    }
    cs_f4 acc[3] = {cs_f4{0.f, 0.f, 0.f, 0.f}, cs_f4{0.f, 0.f, 0.f, 0.f}, cs_f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < 32; ++j) {      // chunk j: tap j / 2 = (kh, kw), channels 16 (j % 2) ..
      if (j + CS_PREFETCH < 32) cs_load_b(bq[(j + CS_PREFETCH) % (CS_PREFETCH + 1)], rsw2, voff23, j + CS_PREFETCH, 64);
      const int tap = j >> 1, kh = tap >> 2, kw = tap & 3;
      const int koff = (kh * 15 + kw) * CS_P1 + 16 * (j & 1);
      cs_f4 av[3];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) av[mt] = *(const cs_f4*)(act1 + rb[mt] + koff);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][i], bq[j % (CS_PREFETCH + 1)][i], acc[mt], 0, 0, 0);
    }
    const float bias = net.b[1][16 * w + c];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * mt + 4 * q + v;
        if (row < 36) act2[row * CS_P2 + 16 * w + c] = fmaxf(acc[mt][v] + bias, 0.f);
      }
  }
#pragma unroll
  for (int j = 0; j < CS_PREFETCH; ++j) cs_load_b(bq[j], rsw3, voff23, j, 64);
  CS_STAMP(4);
  __syncthreads();
  CS_STAMP(5);
  if (net.a2) {      // 36 rows x 256 bytes, 16 bytes per lane: in flight while conv3 computes
    float* dst = net.a2 + (int64_t)smp * 36 * 64;
    for (int e = t; e < 36 * 16; e += 256) {
      const int row = e >> 4, ch = e & 15;
      *(cs_f4*)(dst + row * 64 + 4 * ch) = *(const cs_f4*)(act2 + row * CS_P2 + 4 * ch);
    }
  }
  CS_STAMP(6);

  // =========================================================================== conv3: 3x3 stride 1, 64 -> 64 (one row tile)
  {
    const int rb = ((c >> 2) * 6 + (c & 3)) * CS_P2 + 4 * q;
    cs_f4 acc[4] = {cs_f4{0.f, 0.f, 0.f, 0.f}, cs_f4{0.f, 0.f, 0.f, 0.f}, cs_f4{0.f, 0.f, 0.f, 0.f}, cs_f4{0.f, 0.f, 0.f, 0.f}};
    auto a_at = [&](int j) -> cs_f4 {      // chunk j: tap j / 4, channels 16 (j % 4) ..
      const int tap = j >> 2, kh = tap / 3, kw = tap - 3 * kh;
      return *(const cs_f4*)(act2 + rb + (kh * 6 + kw) * CS_P2 + 16 * (j & 3));
    };
    cs_f4 av0 = a_at(0), av1 = a_at(1);
#pragma unroll
    for (int j = 0; j < 36; ++j) {
      if (j + CS_PREFETCH < 36) cs_load_b(bq[(j + CS_PREFETCH) % (CS_PREFETCH + 1)], rsw3, voff23, j + CS_PREFETCH, 64);
      const cs_f4 av = av0;
      av0 = av1;
      if (j + 2 < 36) av1 = a_at(j + 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bq[j % (CS_PREFETCH + 1)][i], acc[i], 0, 0, 0);
    }
    const float bias = net.b[2][16 * w + c];
    float* dst = net.a3 + (int64_t)smp * 16 * 64 + 16 * w + c;
#pragma unroll
    for (int v = 0; v < 4; ++v)
      dst[(4 * q + v) * 64] = fmaxf(((acc[0][v] + acc[1][v]) + (acc[2][v] + acc[3][v])) + bias, 0.f);
  }
  CS_STAMP(7);
  CS_WALL(9);
}

#endif  // GRL_HOSTEMU

}  // namespace grl
