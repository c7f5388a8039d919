// conv_stack.h -- the convolution stack of the feature extractor as ONE sample-local launch.
//
// /root/reference/manipulation_main/training/custom_obs_policy.py:34-37 (and stable-baselines' nature_cnn): three VALID
// convolutions with bias + ReLU, 64x64xC -> 8x8 s4 -> 15x15x32 -> 4x4 s2 -> 6x6x64 -> 3x3 s1 -> 4x4x64, then the NHWC
// flatten that feeds cnn_fc1.  Per sample the intermediate activations are 28.8 KB + 9.2 KB + 4 KB and every layer has
// N <= 64 output channels, so ONE workgroup can own a whole sample of one network and keep the chain in its LDS: the
// hand-overs between the layers are __syncthreads(), not launches, and conv2 / conv3 never read their input from memory
// (round 4 wrote 22 MB of layer-1 activations per update and read them straight back).  Networks whose activations the
// backward pass needs (model/pi, model/values_fn) still store them -- whole 128-byte rows from LDS, each wave its own tile as
// soon as it is done; the target network stores only its last layer.
//
// Work decomposition (f32 matrix cores, v_mfma_f32_16x16x4_f32, exact fp32):
//   * grid = networks x batch samples, 256 threads.  48.8 KB of LDS per workgroup (one image channel; two: 65 KB): three
//     workgroups share a CU (3 x B = 768 workgroups at B = 256 = exactly three per CU, those of a CU from ONE network);
//   * the GEMM rows are the output pixels of THIS sample (225 / 36 / 16: 15 / 3 / 1 row tiles of 16), the columns the output
//     channels (two / four / four column tiles of 16);
//   * A operand (patches): conv1 from the sample's image, staged in LDS once with 16-byte loads (four 4-byte LDS reads per
//     chunk of 16: the k-order below rules out one 16-byte read); conv2 / conv3 from the activation tile in LDS (pixel stride
//     padded to 36 / 68 floats: 16-byte reads, at most 2-way bank conflicts);
//   * B operand (kernels, TF's HWIO layout untouched): global memory -> registers, prefetched two reduction chunks ahead;
//     every workgroup of a network reads the same 283 KB from L2 (perfectly cached kernels would buy 3 of the 40 us: measured
//     with -DCS_FAKE_B);
//   * reduction order: increasing k, one fmaf per step -- bit-identical to the per-layer implicit-GEMM launches and to a scalar
//     fmaf loop (see "Reduction order" below; scripts/conv_stack_bench.hip checks the bits against such a loop).
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

enum { CS_MAX_NETS = 3 };
struct ConvStackArgs {
  ConvStackNet nets[CS_MAX_NETS];   // by value: the descriptors arrive with the kernel arguments (no dependent load of their own)
  int B;
  int n_nets;
#ifdef CS_STAMPS
  unsigned long long* stamps;
#endif
};

// backward-data of conv3 and conv2 for one network (conv_stack_bwd_kernel)
struct ConvStackBwdNet {
  const float* g3;       // [B * 16][64]   gradient w.r.t. the layer-3 activations (ReLU mask already applied by the dense backward)
  const float* a2;       // [B * 36][64]   layer-2 activations (ReLU mask of g2)
  const float* a1;       // [B * 225][ld1] layer-1 activations (ReLU mask of g1)
  const float* w2;       // [4,4,32,64] HWIO
  const float* w3;       // [3,3,64,64] HWIO
  float* g2;             // [B * 36][64]   out: gradient w.r.t. the layer-2 pre-activations
  float* g1;             // [B * 225][ld1] out: gradient w.r.t. the layer-1 pre-activations
  int ld1;
  int pad;
};

struct ConvStackBwdArgs {
  ConvStackBwdNet nets[2];   // by value, like ConvStackArgs
  int B;
  int n_nets;
};

// LDS of a workgroup: the layer-1 tile [225][CS_P1] and a second region that holds the INPUT IMAGE while conv1 runs (one
// channel: 4096 floats, staged with 16-byte loads) and the layer-2 tile [36][CS_P2] afterwards
enum { CS_P1 = 36, CS_P2 = 68, CS_R2 = 64 * 64, CS_LDS_FLOATS = 225 * CS_P1 + CS_R2 };
static_assert(CS_R2 >= 36 * CS_P2, "the layer-2 tile lives where the image was");
// (two image channels: 8192 floats staged, 65 KB per workgroup, two per CU; four channels are not staged -- 64 KB of image per
//  sample: see CS_BAND below)
// (four image channels: 64 KB per sample -- staged in eight BANDS of 16 image rows, channel-planar: CS_BAND floats; 48.9 KB per
//  workgroup like the one-channel form, three per CU)
enum { CS_PLANE = 16 * 64 + 4, CS_BAND = 4 * CS_PLANE };
static_assert(CS_BAND >= 36 * CS_P2, "the layer-2 tile lives where the band was");
constexpr int cs_lds_floats(int C) { return 225 * CS_P1 + (C <= 2 ? CS_R2 * C : (C == 4 ? CS_BAND : 36 * CS_P2)); }

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
// Reduction order.  The sums of every output are formed in INCREASING k, one fused multiply-add per step, exactly like the
// per-layer implicit-GEMM launches and the scalar reference loops (the f32 MFMA is a chain of fmaf over its four k values,
// lane group q = 0 .. 3 in that order): a ReLU whose pre-activation is zero up to rounding then takes the same side in every
// implementation (scripts/conv_stack_flips.py shows what happens otherwise: conv1 outputs of 1e-10 among activations
// of 1e-2 flip their mask between two summation orders).  Hence, inside a chunk of 16, MFMA step i must see k = 4 i + q from
// lane group q -- while 16-byte LDS reads hand lane group q the positions 4 q .. 4 q + 3.  The activation tiles in LDS
// therefore keep the channels of every 16-chunk TRANSPOSED: channel 4 i + q sits at position 4 q + i (cs_pos).
__device__ __forceinline__ constexpr int cs_pos(int ch) { return (ch & ~15) | ((ch & 3) << 2) | ((ch >> 2) & 3); }
// the four reduction steps k = 16 j + 4 i + q (i = 0 .. 3) of column n of a [K][ldw] kernel matrix; voff = (q * ldw + n) * 4
__device__ __forceinline__ void cs_load_b(float (&bw)[4], __amdgpu_buffer_rsrc_t rs, int voff, int j, int ldw) {
#pragma unroll
#ifdef CS_FAKE_B      // measurement only (WRONG results): every chunk reads chunk 0 -- what would perfect kernel caching buy?
  for (int i = 0; i < 4; ++i) bw[i] = cs_ld(rs, voff + i * 4 * ldw * 4, 0 * j);
#else
  for (int i = 0; i < 4; ++i) bw[i] = cs_ld(rs, voff + i * 4 * ldw * 4, j * 16 * ldw * 4);
#endif
}
// 16 bytes of channels 4 m .. 4 m + 3 of an activation row in LDS (un-transposed: what memory and the backward pass see)
__device__ __forceinline__ cs_f4 cs_row4(const float* row, int m) {
  const float* p = row + ((4 * m) & ~15) + (m & 3);
  return cs_f4{p[0], p[4], p[8], p[12]};
}

// block -> (network, sample).  With a whole number of workgroups per CU the three (..) workgroups that share a CU -- block b
// runs on CU b % 256 (observed placement, used for speed only) -- belong to the same network and read the same kernels.
__device__ __forceinline__ void cs_unit(int B, int n_nets, int& net, int& smp) {
  const int n_units = B * n_nets;
  int j = blockIdx.x;
  if ((n_units & 255) == 0) {
    const int slots = n_units >> 8;
    j = (blockIdx.x & 255) * slots + (blockIdx.x >> 8);
  }
  net = j / B;
  smp = j - net * B;
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
// Issue priority falls as a workgroup advances (3 in conv1 .. 0 in conv3).  The SIMD arbitrates by priority, then AGE: left
// alone, the first-dispatched of the three workgroups of a CU wins every MFMA slot, finishes at 28 us and leaves the youngest
// to run the last 5 us alone at a lone workgroup's efficiency (wall-clock stamps of scripts/conv_stack_bench.hip).  With the
// laggard always in front they finish together.
#ifndef CS_PRIO
#define CS_PRIO 1
#endif
#if CS_PRIO
#define CS_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define CS_SETPRIO(p) do { } while (0)
#endif

template <int C>
__global__ __launch_bounds__(256, (C == 1 ? CS_WG_PER_CU : (C == 4 ? 3 : 2))) void conv_stack_fwd_kernel(ConvStackArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[cs_lds_floats(C)];
  float* const act1 = lds;
  float* const act2 = lds + 225 * CS_P1;
  float* const ximg = lds + 225 * CS_P1;      // (C <= 2) the sample's image, dead once conv1 is done
  int net_i, smp;
  cs_unit(a.B, a.n_nets, net_i, smp);
  // the descriptor of this workgroup's network, selected field by field from the kernel arguments (scalar selects: indexing the
  // by-value array with a run-time index would send it through scratch memory)
  ConvStackNet net = a.nets[0];
  if (net_i == 1) net = a.nets[1];
  if (net_i == 2) net = a.nets[2];
  const int t = threadIdx.x, w = t >> 6, l = t & 63, c = l & 15, q = l >> 4;
  CS_STAMP(0);
  CS_WALL(8);
  CS_SETPRIO(3);
  const __amdgpu_buffer_rsrc_t rsw1 = cs_rsrc(net.w[0]), rsw2 = cs_rsrc(net.w[1]), rsw3 = cs_rsrc(net.w[2]);
  const int voff23 = (q * 64 + 16 * w + c) * 4;            // conv2 / conv3: wave w owns output channels 16 w .. 16 w + 15
  float bq[CS_PREFETCH + 1][4];                            // conv2 / conv3 kernel chunks in flight: CS_PREFETCH ahead of the multiply

  // =========================================================================== conv1: 8x8 stride 4, C -> 32
  {
    constexpr int NJ = 4 * C;                       // reduction chunks of 16 (K = 64 C)
    constexpr bool RES = C == 1;                    // the whole kernel matrix stays in registers (32 per lane)
    const float* x = net.x + (int64_t)smp * (64 * 64 * C);
    // lane (row, q), chunk j, step i reads the patch element k = 16 j + 4 i + q of the HWIO order k = (kh * 8 + kw) * C + ch:
    // a base that depends on the lane plus an offset known at compile time
    auto a_base = [&](int row) -> int {
      const int r = row < 225 ? row : 224;
      const int oh = r / 15, ow = r - oh * 15, px = oh * 4 * 64 + ow * 4;
      return C == 1 ? px + q : (C == 2 ? (px + (q >> 1)) * 2 + (q & 1) : px * 4 + q);
    };
    auto a_koff = [](int j, int i) -> int {
      return C == 1 ? (2 * j + (i >> 1)) * 64 + 4 * (i & 1) : (C == 2 ? (j * 64 + 2 * i) * 2 : (((4 * j + i) >> 3) * 64 + ((4 * j + i) & 7)) * 4);
    };
    // C <= 2: the image is staged in LDS once (four 16-byte loads per thread and channel) and the patches are read from there --
    // four 4-byte global loads per chunk and row tile cost the first layer 6 k cycles (scripts/conv_stack_bench.hip stamps)
    constexpr bool STAGE = C <= 2;
    if (STAGE) {
#pragma unroll
      for (int r = 0; r < 4 * C; ++r) *(cs_f4*)(ximg + 4 * (t + 256 * r)) = *(const cs_f4*)(x + 4 * (t + 256 * r));
    }
    auto a_load = [&](cs_f4& v, int base, int j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = STAGE ? ximg[base + a_koff(j, i)] : x[base + a_koff(j, i)];
    };
    const int voff1 = (q * 32 + c) * 4;
    cs_f4 an[RES ? NJ : 1];
    float bw[RES ? NJ : 1][2][4];
    if (RES) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) cs_load_b(bw[RES ? j : 0][nt], rsw1, voff1 + 64 * nt, j, 32);
    }
    if (STAGE) __syncthreads();
    if (RES) {
      const int b0 = a_base(16 * w + c);
#pragma unroll
      for (int j = 0; j < NJ; ++j) a_load(an[RES ? j : 0], b0, j);
    }
    const float bias0 = net.b[0][c], bias1 = net.b[0][16 + c];
    float* const a1g = net.a1 ? net.a1 + (int64_t)smp * 225 * net.ld1 : nullptr;
    // bias + ReLU of one 16 x 32 tile into the layer-1 tile in LDS (channels transposed, cs_pos) and, for the networks whose
    // weight gradients need them, out to memory
    auto emit_tile = [&](int mt, const cs_f4 (&acc)[2]) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * mt + 4 * q + v;
        if (row < 225) {
          act1[row * CS_P1 + cs_pos(c)] = fmaxf(acc[0][v] + bias0, 0.f);
          act1[row * CS_P1 + 16 + cs_pos(c)] = fmaxf(acc[1][v] + bias1, 0.f);
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
          if (row < 225) *(cs_f4*)(a1g + (int64_t)row * net.ld1 + 4 * ch) = cs_row4(act1 + row * CS_P1, ch);
        }
      }
    };
    if constexpr (C == 4) {
      // ---- RGB-D: the image (64 KB) is staged in eight BANDS of 16 rows, channel-PLANAR: lane group q reads channel q, and the
      // four taps kw .. kw + 3 of a reduction chunk are 16 contiguous bytes of its plane -- one LDS read per chunk where the
      // interleaved image needed four.  Round r: row tiles 2 r and 2 r + 1 (outputs 32 r .. 32 r + 31 lie in output rows 2 r ..
      // 2 r + 2, i.e. image rows 8 r .. 8 r + 15); waves 0 / 1 take the two 16-channel halves of tile 2 r, waves 2 / 3 those of
      // tile 2 r + 1, so a wave keeps HALF of the 256 x 32 kernel in registers (64 per lane) and three workgroups fit a CU
      // (with whole tiles per wave -- 128 kernel registers, 24-row bands, two workgroups per CU -- 768 workgroups filled 512
      // slots one and a half times: 79 us, no better than the three per-layer launches).  The next band travels from memory
      // while a tile is multiplied.
      float* const xb = ximg;
      const int rt = w >> 1, nt = w & 1;
      float bw4[16][4];
#pragma unroll
      for (int j = 0; j < 16; ++j) cs_load_b(bw4[j], rsw1, voff1 + 64 * nt, j, 32);
      const float bias_n = net.b[0][16 * nt + c];
      cs_f4 pre[4];
      auto band_load = [&](int r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = t + 256 * e, row = 8 * r + (idx >> 6);
          pre[e] = row < 64 ? *(const cs_f4*)(x + 4 * (row * 64 + (idx & 63))) : cs_f4{0.f, 0.f, 0.f, 0.f};
        }
      };
      auto band_store = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = t + 256 * e;
#pragma unroll
          for (int ch = 0; ch < 4; ++ch) xb[ch * CS_PLANE + idx] = pre[e][ch];
        }
      };
      band_load(0);
#pragma unroll 1
      for (int r = 0; r < 8; ++r) {
        if (r > 0) __syncthreads();          // every wave is done reading band r - 1
        band_store();
        __syncthreads();
        if (r < 7) band_load(r + 1);
        const int mt = 2 * r + rt;
        if (mt < 15) {
          const int p = 16 * mt + c < 225 ? 16 * mt + c : 224;
          const int oh = p / 15, ow = p - oh * 15;
          const float* src = xb + q * CS_PLANE + (4 * oh - 8 * r) * 64 + 4 * ow;
          cs_f4 acc = {0.f, 0.f, 0.f, 0.f};       // ONE chain: the sum runs in increasing k (the other waves of the SIMD fill the gaps)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const cs_f4 av = *(const cs_f4*)(src + (j >> 1) * 64 + 4 * (j & 1));      // taps (kh = j / 2, kw = 4 (j % 2) .. + 3), channel q
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bw4[j][i], acc, 0, 0, 0);
          }
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int row = 16 * mt + 4 * q + v;
            if (row < 225) act1[row * CS_P1 + 16 * nt + cs_pos(c)] = fmaxf(acc[v] + bias_n, 0.f);
          }
          if (a1g) {       // this wave's 16 rows x 16 channels go out right away: one 16-byte store per lane (see emit_tile)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int row = 16 * mt + (l >> 2), m = 4 * nt + (l & 3);
            if (row < 225) *(cs_f4*)(a1g + (int64_t)row * net.ld1 + 4 * m) = cs_row4(act1 + row * CS_P1, m);
          }
        }
      }
        } else
    for (int mt = w; mt < 15; mt += 4) {
      cs_f4 acc[2] = {cs_f4{0.f, 0.f, 0.f, 0.f}, cs_f4{0.f, 0.f, 0.f, 0.f}};
      if (RES) {
        cs_f4 ac[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) ac[j] = an[RES ? j : 0];
        if (mt + 4 < 15) {
          const int bnx = a_base(16 * (mt + 4) + c);
#pragma unroll
          for (int j = 0; j < NJ; ++j) a_load(an[RES ? j : 0], bnx, j);
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
        const int ab = a_base(16 * mt + c);
        cs_f4 av;
        a_load(av, ab, 0);
        float b0[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) cs_load_b(b0[nt], rsw1, voff1 + 64 * nt, 0, 32);
#pragma unroll 4
        for (int j = 0; j < NJ; ++j) {
          float b1[2][4];
          const int jn = j + 1 < NJ ? j + 1 : j;
          cs_f4 avn;
          a_load(avn, ab, jn);
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
      emit_tile(mt, acc);
    }
  }
  // the first kernel chunks of conv2 do not depend on this workgroup's activations: requested before the barrier
#pragma unroll
  for (int j = 0; j < CS_PREFETCH; ++j) cs_load_b(bq[j], rsw2, voff23, j, 64);
  CS_STAMP(1);
  __syncthreads();
  CS_STAMP(2);
  CS_SETPRIO(2);

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
      if (j == 16) CS_SETPRIO(1);
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
        if (row < 36) act2[row * CS_P2 + 16 * w + cs_pos(c)] = fmaxf(acc[mt][v] + bias, 0.f);
      }
  }
#pragma unroll
  for (int j = 0; j < CS_PREFETCH; ++j) cs_load_b(bq[j], rsw3, voff23, j, 64);
  CS_STAMP(4);
  __syncthreads();
  CS_STAMP(5);
  CS_SETPRIO(0);
  if (net.a2) {      // 36 rows x 256 bytes, 16 bytes per lane: in flight while conv3 computes
    float* dst = net.a2 + (int64_t)smp * 36 * 64;
    for (int e = t; e < 36 * 16; e += 256) {
      const int row = e >> 4, ch = e & 15;
      *(cs_f4*)(dst + row * 64 + 4 * ch) = cs_row4(act2 + row * CS_P2, ch);
    }
  }
  CS_STAMP(6);

  // =========================================================================== conv3: 3x3 stride 1, 64 -> 64 (one row tile)
  {
    const int rb = ((c >> 2) * 6 + (c & 3)) * CS_P2 + 4 * q;
    // (ONE accumulator: the k-order above; its 40-cycle dependent latency against the 32-cycle issue is covered by the other
    //  workgroups' waves of this SIMD)
    cs_f4 acc = {0.f, 0.f, 0.f, 0.f};
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
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bq[j % (CS_PREFETCH + 1)][i], acc, 0, 0, 0);
    }
    const float bias = net.b[2][16 * w + c];
    float* dst = net.a3 + (int64_t)smp * 16 * 64 + 16 * w + c;
#pragma unroll
    for (int v = 0; v < 4; ++v) dst[(4 * q + v) * 64] = fmaxf(acc[v] + bias, 0.f);
  }
  CS_STAMP(7);
  CS_WALL(9);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward-data of conv3 and conv2, sample-local (VERDICT r4 "Next 2": conv3_bwd -> conv2_bwd).  One workgroup owns a sample
// of a trained network; both gradients are formed in SCATTER form -- per kernel tap one small product
//     contribution[output pixels, c_in] = dY[output pixels, c_out] . W[tap][c_in, c_out]^T
// added into the input pixel each output pixel's tap touches -- so the MACs are exactly those of the forward pass (the gather
// form pays 2.25x / 1.56x for the taps that fall outside the image, and the exact-tap decomposition of round 3 needs 25 / 36
// problems per network).  Both operands are contiguous along the reduction (c_out is HWIO's fastest index): 16-byte loads.
//   conv3: wave w owns input channels 16 w .. + 15 of all 36 pixels: nine taps x 16 MFMAs, accumulated in LDS by that wave alone;
//   conv2: wave w owns one stride-parity class of the 15 x 15 input pixels (the four classes never touch the same pixel): its
//          four taps x (3 row tiles x 2 channel tiles x 16) MFMAs.
// Every LDS cell has ONE writer and the additions happen in program order: sums are deterministic.  The ReLU masks are applied
// where the results leave for memory (g2 also stays in LDS as conv2's operand).
// (two trained networks x B samples = 2 workgroups per CU at B = 256: the register budget of two waves per SIMD)
// Latency measures, as in the forward kernel: descriptors in the kernel arguments; the ReLU masks (a2 rows, the a1 rows of the
// wave's own parity class) are requested long before they are needed; every wave finishes ITS share of g1 alone -- no barrier
// and no burst of all 225 rows at the end; the two workgroups of a CU belong to the same network; issue priority falls with
// progress.
template <int DUMMY>
__global__ __launch_bounds__(256, 2) void conv_stack_bwd_kernel(ConvStackBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[225 * CS_P1 + 36 * CS_P2];
  float* const g1acc = lds;
  float* const g2acc = lds + 225 * CS_P1;
  int net_i, smp;
  cs_unit(a.B, a.n_nets, net_i, smp);
  ConvStackBwdNet net = a.nets[0];
  if (net_i == 1) net = a.nets[1];
  const int t = threadIdx.x, w = t >> 6, l = t & 63, c = l & 15, q = l >> 4;
  const cs_f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  CS_SETPRIO(2);

  // ---- operands of the first stage and the layer-2 mask, requested before anything else; accumulators cleared meanwhile
  const float* g3 = net.g3 + ((int64_t)smp * 16 + c) * 64 + 4 * q;
  cs_f4 a3[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a3[j] = *(const cs_f4*)(g3 + 16 * j);
  const float* w3 = net.w3 + (16 * w + c) * 64 + 4 * q;         // + tap * 4096 + 16 j
  cs_f4 bn[2][4];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int j = 0; j < 4; ++j) bn[d][j] = *(const cs_f4*)(w3 + d * 4096 + 16 * j);
  const float* a2 = net.a2 + (int64_t)smp * 36 * 64;
  cs_f4 m2[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int e = t + 256 * r;
    m2[r] = e < 36 * 16 ? *(const cs_f4*)(a2 + 4 * e) : zero4;
  }
  for (int e = t; e < (225 * CS_P1 + 36 * CS_P2) / 4; e += 256) *(cs_f4*)(lds + 4 * e) = zero4;
  __syncthreads();

  // =========================================================================== conv3, 3x3 stride 1: g3 [16, 64] -> g2 [36, 64]
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    cs_f4 bw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bw[j] = bn[tap & 1][j];
    if (tap + 2 < 9) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bn[tap & 1][j] = *(const cs_f4*)(w3 + (tap + 2) * 4096 + 16 * j);
    }
    cs_f4 acc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[j][i], bw[j][i], acc[j], 0, 0, 0);
    const int kh = tap / 3, kw = tap - 3 * kh;
    float* dst = g2acc + ((q + kh) * 6 + kw) * CS_P2 + 16 * w + c;      // output pixel (q, v) -> input pixel (q + kh, v + kw)
    // (the four cells of a tap are distinct: all reads, then all writes -- one LDS round trip per tap instead of four)
    float old[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) old[v] = dst[v * CS_P2];
#pragma unroll
    for (int v = 0; v < 4; ++v) dst[v * CS_P2] = old[v] + ((acc[0][v] + acc[1][v]) + (acc[2][v] + acc[3][v]));
  }
  // ---- across the barrier travel: the first kernel tap of conv2 (this wave's parity class) and the layer-1 mask rows of the
  // class (pixel (2 a + ph, 2 b + pw), a < na, b < nb: 64 / 56 / 56 / 49 pixels of 32 channels = up to 8 x 16 bytes per lane)
  const int ph = w >> 1, pw = w & 1, na = 8 - ph, nb = 8 - pw, n4 = na * nb * 8;
  const float* w2 = net.w2 + c * 64 + 4 * q;                       // + tap * 2048 + nt * 1024 + 16 j
  cs_f4 b2n[2][4];
  {
    const int tap0 = ph * 4 + pw;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) b2n[nt][j] = *(const cs_f4*)(w2 + tap0 * 2048 + nt * 1024 + 16 * j);
  }
  const float* a1 = net.a1 + (int64_t)smp * 225 * net.ld1;
  float* g1 = net.g1 + (int64_t)smp * 225 * net.ld1;
  cs_f4 m1[8];
  int pix1[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int e = l + 64 * r, px = e >> 3, pa = px / nb, pb = px - pa * nb;
    pix1[r] = e < n4 ? (2 * pa + ph) * 15 + 2 * pb + pw : -1;
    m1[r] = pix1[r] >= 0 ? *(const cs_f4*)(a1 + (int64_t)pix1[r] * net.ld1 + 4 * (l & 7)) : zero4;
  }
  CS_SETPRIO(1);
  __syncthreads();
  // ---- g2 = sums x (a2 > 0): back into LDS (conv2's operand) and out to memory (conv2's weight gradient reads it)
  {
    float* g2 = net.g2 + (int64_t)smp * 36 * 64;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int e = t + 256 * r;
      if (e < 36 * 16) {
        const int row = e >> 4, ch = e & 15;
        cs_f4 v = *(const cs_f4*)(g2acc + row * CS_P2 + 4 * ch);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = m2[r][i] > 0.f ? v[i] : 0.f;
        *(cs_f4*)(g2acc + row * CS_P2 + 4 * ch) = v;
        *(cs_f4*)(g2 + 4 * e) = v;
      }
    }
  }
  __syncthreads();

  // =========================================================================== conv2, 4x4 stride 2: g2 [36, 64] -> g1 [225, 32]
  {
    cs_f4 ag[3][4];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      const int r = 16 * mt + c < 36 ? 16 * mt + c : 35;
#pragma unroll
      for (int j = 0; j < 4; ++j) ag[mt][j] = *(const cs_f4*)(g2acc + r * CS_P2 + 16 * j + 4 * q);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int kh = ph + 2 * (d >> 1), kw = pw + 2 * (d & 1);
      cs_f4 bw[2][4];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) bw[nt][j] = b2n[nt][j];
      if (d + 1 < 4) {
        const int tapn = (ph + 2 * ((d + 1) >> 1)) * 4 + pw + 2 * ((d + 1) & 1);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) b2n[nt][j] = *(const cs_f4*)(w2 + tapn * 2048 + nt * 1024 + 16 * j);
      }
      if (d == 2) CS_SETPRIO(0);
      cs_f4 acc[3][2] = {{zero4, zero4}, {zero4, zero4}, {zero4, zero4}};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[mt][j][i], bw[nt][j][i], acc[mt][nt], 0, 0, 0);
      // output pixel (oh, ow) of row 16 mt + 4 q + v -> input pixel (2 oh + kh, 2 ow + kw).  The cells of one tap are distinct:
      // all reads, then all writes (one LDS round trip per tap instead of twelve dependent ones)
      int cell[3][4];
      float old[3][4][2];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * mt + 4 * q + v;
          const int oh = row / 6, ow = row - 6 * oh;
          cell[mt][v] = row < 36 ? ((2 * oh + kh) * 15 + 2 * ow + kw) * CS_P1 + c : -1;
          if (cell[mt][v] >= 0) { old[mt][v][0] = g1acc[cell[mt][v]]; old[mt][v][1] = g1acc[cell[mt][v] + 16]; }
        }
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (cell[mt][v] >= 0) {
            g1acc[cell[mt][v]] = old[mt][v][0] + acc[mt][0][v];
            g1acc[cell[mt][v] + 16] = old[mt][v][1] + acc[mt][1][v];
          }
    }
  }
  // ---- g1 of this wave's parity class = sums x (a1 > 0): nobody else wrote these pixels, nobody else stores them (LDS
  // operations of one wave execute in order: the fence only keeps the compiler from moving the reads up)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (pix1[r] >= 0) {
      cs_f4 v = *(const cs_f4*)(g1acc + pix1[r] * CS_P1 + 4 * (l & 7));
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = m1[r][i] > 0.f ? v[i] : 0.f;
      *(cs_f4*)(g1 + (int64_t)pix1[r] * net.ld1 + 4 * (l & 7)) = v;
    }
}

#endif  // GRL_HOSTEMU

}  // namespace grl
