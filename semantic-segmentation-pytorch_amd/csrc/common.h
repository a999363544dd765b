// Shared device/host helpers for libsemseg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "semseg_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// While the host has a side-by-side scope open (csrc/batch.h) the launches of the converted kernel families are recorded, not
// issued.  Every OTHER launch of the library must not overtake them: it first issues whatever has been recorded (correct for any
// kernel; only the converted ones are batched) and leaves on the scope's stream -- the stream the recorded launches will be issued on,
// whatever stream the host had current when it made the call.  One load of a bool when no scope is open.
namespace semseg_batch {
bool recording();
hipStream_t direct_stream(hipStream_t requested, const char* what);      // no scope: `requested`; else: flush what is recorded, then the scope's stream
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                  \
    do {                                                                                                                   \
        hipStream_t st__ = (hipStream_t)(streamId);                                                                        \
        if (semseg_batch::recording()) st__ = semseg_batch::direct_stream(st__, "BODY = " #kernelName "]");                                      \
        kernelName<<<(numBlocks), (numThreads), (memPerBlock), st__>>>(__VA_ARGS__);                                       \
    } while (0)

#define SEMSEG_LAUNCH_CHECK()                       \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): one cache per launch site, one entry per device,
// so a process that drives several devices (the in-process peer tests) sets it on each of them before its first > 64 KiB launch.
struct SmemAttrCache {
    size_t set[32] = {};
};
static inline int ensure_smem_attr(SmemAttrCache& c, const void* kernel, size_t smem) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = -1;
    if (dev >= 0 && c.set[dev] >= smem) return 0;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0) c.set[dev] = smem;
    return 0;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t ceil_div_sz(size_t a, size_t b) { return (a + b - 1) / b; }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// load up to 4 consecutive floats; VEC: nvalid is 0 or 4 and p is 16-byte aligned
template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ p, int nvalid) {
    if (VEC) {
        return nvalid > 0 ? *reinterpret_cast<const float4*>(p) : f4zero();
    } else {
        float4 v = f4zero();
        if (nvalid > 0) v.x = p[0];
        if (nvalid > 1) v.y = p[1];
        if (nvalid > 2) v.z = p[2];
        if (nvalid > 3) v.w = p[3];
        return v;
    }
}

// 64-lane wave reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#include "block_order.h"     // xcd_remap, gemm_block, wgrad_block (shared with the host-side enumeration test)
