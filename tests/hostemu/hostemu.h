// hostemu.h -- TEST-ONLY shim that lets engine.hip be compiled with plain g++ (-DGRL_HOSTEMU) so the
// host-side planning logic (parameter layout, addressing tables, problem descriptors, launch order)
// can be exercised in the GPU-less build container.  Kernels are run as sequential loops over
// (block, thread); the MFMA kernel and the block-reduction kernel are replaced by straightforward
// reference loops with the same descriptor semantics.
//
// This is NOT a product path: libgrl.so is never built with GRL_HOSTEMU, the Python host never loads
// the emulation library, and every product entry point fails loudly without a HIP device.  The only
// consumer is tests/test_hostemu_plan.py, which builds tests/_build/libgrl_hostemu.so.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

// ONE instance for the whole library (C++17 inline variables): the kernel templates are instantiated in several translation
// units and the linker keeps one copy of each -- which must read the indices the launcher of ANY unit set
inline thread_local dim3 threadIdx, blockIdx, gridDim, blockDim;
static inline void __syncthreads() {}
static inline uint64_t __umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
using std::max;
using std::min;

typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamCaptureModeThreadLocal = 0 };
static inline const char* hipGetErrorString(hipError_t) { return "hostemu"; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
enum { hipHostMallocCoherent = 0x40000000 };
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = malloc(n); return *p ? 0 : 1; }
static inline hipError_t hipHostFree(void* p) { free(p); return 0; }
static inline hipError_t hipGetLastError() { return 0; }
struct hipPointerAttribute_t { int type; };
enum { hipMemoryTypeHost = 1, hipMemoryTypeManaged = 3 };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t*, const void*) { return 1; }   // every pointer is pageable host memory here
static inline hipError_t hipEventCreate(hipEvent_t*) { return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0; return 0; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t*, int) { return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t*, int) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, int) { return 0; }
static inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return 1; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return 1; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, int) { return 1; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return 0; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return 0; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 1; }

// exchange-buffer plumbing of the data-parallel path (csrc/dp_kernels.h): plain host memory, "IPC handles" carry the
// pointer (one process only)
struct hipIpcMemHandle_t { char reserved[64]; };
enum { hipDeviceMallocFinegrained = 1, hipIpcMemLazyEnablePeerAccess = 1 };
static inline hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { *p = calloc(1, n); return *p ? 0 : 1; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n); return *p ? 0 : 1; }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) { memset(h, 0, sizeof(*h)); memcpy(h->reserved, &p, sizeof(p)); return 0; }
static inline hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) { memcpy(p, h.reserved, sizeof(*p)); return 0; }
static inline hipError_t hipIpcCloseMemHandle(void*) { return 0; }
static inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p += v; return o; }

template <class K, class... Args>
static inline void hostemu_launch(K kernel, dim3 grid, dim3 block, Args... args) {
  gridDim = grid;
  blockDim = block;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx)
        for (unsigned tx = 0; tx < block.x; ++tx) {
          blockIdx = dim3(bx, by, bz);
          threadIdx = dim3(tx, 0, 0);
          kernel(args...);
        }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hostemu_launch(kernel, grid, block, __VA_ARGS__)
