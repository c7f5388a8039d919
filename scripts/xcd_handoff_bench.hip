// xcd_handoff_bench.hip -- GATE measurement for in-launch dataflow on MI355X (round-4 review, item 2): what does ONE
// producer -> consumer hand-over of a 32 KB tile cost inside a launch when both workgroups sit on the SAME XCD (shared L2:
// plain stores, drained, a flag; the consumer bypasses only its L1), against the same hand-over across XCDs (write-through
// stores / system-scope loads, or release / acquire fences) and against a dependent kernel boundary inside a hipGraph?
//
//   hipcc --offload-arch=gfx950 -O3 scripts/xcd_handoff_bench.hip -o build/xcd_handoff_bench && build/xcd_handoff_bench
//
// Work per stage and workgroup (identical in every variant): read a 32 KB tile another workgroup wrote in the previous stage
// (8 x 16-byte loads per thread, all in flight), add 1, write an own 32 KB tile.  256 workgroups (one per CU), 64 stages.
//   graph      64 dependent kernels of 256 workgroups in one hipGraph                       -> us per stage = boundary + work
//   xcd        one launch; workgroup l of XCD x reads the tile of workgroup l+1 of the SAME XCD (XCC_ID + per-XCD ticket);
//              producer: plain stores, s_waitcnt vmcnt(0), barrier, relaxed flag store; consumer: one lane polls (sc1 load),
//              barrier, payload with sc1 loads (L1 bypass, served by the XCD's L2)
//   cross_wt   the partner sits on the NEXT XCD; write-through stores (sc0 sc1) + drained flag, sc0 sc1 loads
//   cross_fence the partner sits on the next XCD; plain stores + agent release fence + flag, agent acquire fence + plain loads
//              (the form of round 3's rejected chain kernel)
// Every variant checks every word it reads (stale data is counted, not assumed away) and every wait is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

enum { NWG = 256, TILE = 8192, STAGES = 64, SPIN = 1 << 22 };
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* p) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
template <int AUX> __device__ __forceinline__ f4 ld(const float* base, int quad) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc(base), quad << 4, 0, AUX));
}
template <int AUX> __device__ __forceinline__ void st(float* base, int quad, f4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rsrc(base), quad << 4, 0, AUX);
}

struct Ctl {
  unsigned ticket[8];        // per-XCD arrival tickets
  unsigned arrived;          // workgroups registered
  unsigned errors;           // stale words / time-outs
  unsigned xcd_count[8];
  unsigned pad[14];
};

// one stage of work: tile_in (32 KB) -> +1 -> tile_out; returns the number of words that were not `expect`
template <int LD_AUX, int ST_AUX>
__device__ __forceinline__ int stage_work(const float* in, float* out, float expect) {
  f4 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = ld<LD_AUX>(in, k * 256 + threadIdx.x);
  int bad = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    bad += (v[k].x != expect) + (v[k].y != expect) + (v[k].z != expect) + (v[k].w != expect);
    st<ST_AUX>(out, k * 256 + threadIdx.x, v[k] + 1.f);
  }
  return bad;
}

// ---- variant "graph": one kernel per stage
__global__ __launch_bounds__(256) void stage_kernel(const float* in_all, float* out_all, float expect, Ctl* ctl) {
  const int w = blockIdx.x, src = (w + 1) % NWG;
  const int bad = stage_work<0, 0>(in_all + (size_t)src * TILE, out_all + (size_t)w * TILE, expect);
  if (bad) atomicAdd(&ctl->errors, 1u);
}

// ---- the in-launch variants.  MODE 0: same XCD, plain + sc1;  1: next XCD, write-through;  2: next XCD, fences
template <int MODE>
__global__ __launch_bounds__(256) void chain_kernel(float* tiles, unsigned* flags, Ctl* ctl, unsigned long long* t_out) {
  __shared__ int sh[4];
  const int t = threadIdx.x;
  // register: XCD id (hardware register, speed only) and a slot inside it
  if (t == 0) {
    const unsigned xcd = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;     // HW_REG_XCC_ID[3:0]
    const unsigned local = atomicAdd(&ctl->ticket[xcd], 1u);
    sh[0] = (int)xcd; sh[1] = (int)local;
    __threadfence();
    atomicAdd(&ctl->arrived, 1u);
    int spins = 0;
    while (__hip_atomic_load(&ctl->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)NWG && ++spins < SPIN) __builtin_amdgcn_s_sleep(4);
    sh[2] = spins >= SPIN;
    __threadfence();
  }
  __syncthreads();
  const int xcd = sh[0], local = sh[1];
  if (sh[2]) { if (t == 0) atomicAdd(&ctl->errors, 1000000u); return; }
  // partner: same XCD next slot (MODE 0) / same slot on the next XCD that has it (MODE 1, 2)
  int pxcd = xcd, plocal = local;
  if (MODE == 0) {
    const int cnt = (int)__hip_atomic_load(&ctl->ticket[xcd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    plocal = (local + 1) % cnt;
  } else {
    for (int k = 1; k <= 8; ++k) {
      pxcd = (xcd + k) & 7;
      if ((int)__hip_atomic_load(&ctl->ticket[pxcd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > local) break;
    }
  }
  const int me = xcd * 64 + local, partner = pxcd * 64 + plocal;      // slots (up to 64 workgroups per XCD)
  float* mine_base = tiles + (size_t)me * TILE;
  const float* part_base = tiles + (size_t)partner * TILE;
  const size_t stage_stride = (size_t)8 * 64 * TILE;
  unsigned* my_flag = flags + me * 16;
  unsigned* p_flag = flags + partner * 16;
  // stage 0: everybody writes its own tile (value 1)
  const unsigned long long t0 = wall_clock64();
  {
    f4 one = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 1) st<17>(mine_base, k * 256 + t, one); else st<0>(mine_base, k * 256 + t, one);
    }
  }
  int bad = 0;
  for (int s = 1; s <= STAGES; ++s) {
    // ---- publish stage s-1
    if (MODE == 2) {
      __syncthreads();
      if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(my_flag, (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave: its stores have left (L2 for plain, memory for write-through)
      __syncthreads();
      if (t == 0) __hip_atomic_store(my_flag, (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (s == STAGES) break;
    // ---- wait for the partner's stage s-1
    if (t == 0) {
      int spins = 0;
      while (__hip_atomic_load(p_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)s && ++spins < SPIN) __builtin_amdgcn_s_sleep(2);
      if (spins >= SPIN) atomicAdd(&ctl->errors, 1000000u);
      if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    const float* in = part_base + (size_t)(s - 1) * stage_stride;
    float* out = mine_base + (size_t)s * stage_stride;
    if (MODE == 0) bad += stage_work<16, 0>(in, out, (float)s);          // sc1 loads (L1 bypass), plain stores
    else if (MODE == 1) bad += stage_work<17, 17>(in, out, (float)s);    // sc0 sc1 both ways
    else bad += stage_work<0, 0>(in, out, (float)s);
  }
  const unsigned long long t1 = wall_clock64();
  if (bad) atomicAdd(&ctl->errors, (unsigned)bad);
  if (t == 0) { t_out[2 * blockIdx.x] = t0; t_out[2 * blockIdx.x + 1] = t1; }
}

int main() {
  float* tiles;
  unsigned* flags;
  Ctl* ctl;
  unsigned long long* t_out;
  const size_t tile_floats = (size_t)(STAGES + 1) * 8 * 64 * TILE;
  CHK(hipMalloc(&tiles, tile_floats * 4));
  CHK(hipMalloc(&flags, 8 * 64 * 16 * 4));
  CHK(hipMalloc(&ctl, sizeof(Ctl)));
  CHK(hipMalloc(&t_out, 2 * NWG * 8));
  hipStream_t s;
  CHK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  Ctl hc;

  // ---------------- graph of dependent kernels
  {
    float* buf[2];
    CHK(hipMalloc(&buf[0], (size_t)NWG * TILE * 4)); CHK(hipMalloc(&buf[1], (size_t)NWG * TILE * 4));
    std::vector<float> ones((size_t)NWG * TILE, 1.f);
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < STAGES; ++k)
      hipLaunchKernelGGL(stage_kernel, dim3(NWG), dim3(256), 0, s, buf[k & 1], buf[(k + 1) & 1], (float)(k + 1), ctl);
    CHK(hipStreamEndCapture(s, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 6; ++rep) {
      CHK(hipMemcpy(buf[0], ones.data(), ones.size() * 4, hipMemcpyHostToDevice));
      CHK(hipMemset(ctl, 0, sizeof(Ctl)));
      CHK(hipEventRecord(e0, s));
      CHK(hipGraphLaunch(ge, s));
      CHK(hipEventRecord(e1, s));
      CHK(hipStreamSynchronize(s));
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      CHK(hipMemcpy(&hc, ctl, sizeof(hc), hipMemcpyDeviceToHost));
      if (rep >= 3) printf("graph       %2d dependent kernels: %7.2f us per stage (kernel + boundary)   errors %u\n", STAGES, 1e3 * ms / STAGES, hc.errors);
    }
    // the same kernels WITHOUT the dependency cost visible: one kernel alone, repeated on independent buffers is still a boundary;
    // so also time a single stage kernel in isolation (events around one launch of the warm kernel)
    CHK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(stage_kernel, dim3(NWG), dim3(256), 0, s, buf[0], buf[1], -1.f, ctl);
    CHK(hipEventRecord(e1, s));
    CHK(hipStreamSynchronize(s));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("one stage kernel alone (events around one launch): %7.2f us\n", 1e3 * ms);
  }

  // ---------------- in-launch chains
  const char* names[3] = {"xcd         same XCD, plain stores + drained flag, sc1 loads", "cross_wt    next XCD, write-through stores / system-scope loads",
                          "cross_fence next XCD, release / acquire fences (agent)"};
  for (int mode = 0; mode < 3; ++mode)
    for (int rep = 0; rep < 5; ++rep) {
      CHK(hipMemset(ctl, 0, sizeof(Ctl)));
      CHK(hipMemset(flags, 0, 8 * 64 * 16 * 4));
      CHK(hipMemsetAsync(tiles, 0, tile_floats * 4, s));
      CHK(hipStreamSynchronize(s));
      CHK(hipEventRecord(e0, s));
      if (mode == 0) hipLaunchKernelGGL(chain_kernel<0>, dim3(NWG), dim3(256), 0, s, tiles, flags, ctl, t_out);
      else if (mode == 1) hipLaunchKernelGGL(chain_kernel<1>, dim3(NWG), dim3(256), 0, s, tiles, flags, ctl, t_out);
      else hipLaunchKernelGGL(chain_kernel<2>, dim3(NWG), dim3(256), 0, s, tiles, flags, ctl, t_out);
      CHK(hipEventRecord(e1, s));
      CHK(hipStreamSynchronize(s));
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      CHK(hipMemcpy(&hc, ctl, sizeof(hc), hipMemcpyDeviceToHost));
      std::vector<unsigned long long> tt(2 * NWG);
      CHK(hipMemcpy(tt.data(), t_out, tt.size() * 8, hipMemcpyDeviceToHost));
      double in_kernel = 0;
      for (int w = 0; w < NWG; ++w) in_kernel = std::max(in_kernel, (double)(tt[2 * w + 1] - tt[2 * w]) / 100.0);
      if (rep >= 2)
        printf("%-66s %7.2f us per stage (in-kernel %7.2f)   XCD census %u %u %u %u %u %u %u %u   errors %u\n", names[mode], 1e3 * ms / STAGES,
               in_kernel / STAGES, hc.ticket[0], hc.ticket[1], hc.ticket[2], hc.ticket[3], hc.ticket[4], hc.ticket[5], hc.ticket[6], hc.ticket[7], hc.errors);
    }
  return 0;
}
