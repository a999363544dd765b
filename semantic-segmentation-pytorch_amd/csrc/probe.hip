// Box probes for bench.py's `box` block: what THIS MI355X sustains, measured in the same process as the step, so that a
// slow box shows up as a slow box (round-3 review: the same tree ran 137 ... 153 img/s on different boxes and nothing in the
// line said which kind of box produced it).  Not on the product path; nothing in mit_semseg/ops.py calls these.
//   semseg_probe_timestamp   one constant-rate (100 MHz) timestamp into a device word: a capturable marker for the backward
//                            timeline of a replayed step (bench.py scaling_model: when each gradient bucket is complete)
//   semseg_probe_mfma_f16    dense v_mfma_f32_32x32x16_f16 loop, no memory traffic: TFLOP/s and shader clock under MFMA load
//   semseg_probe_copy        float4 streaming copy: achievable HBM bandwidth
//   semseg_probe_empty       empty kernel: per-node latency of a hipGraph chain
#include "common.h"

typedef _Float16 p_f16x8 __attribute__((ext_vector_type(8)));

__global__ void probe_timestamp_kernel(unsigned long long* slot) {
    if (threadIdx.x == 0) *slot = __builtin_amdgcn_s_memrealtime();
}

extern "C" int semseg_probe_timestamp(void* slot, void* stream) {
    if (!slot || (((uintptr_t)slot) & 7)) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(probe_timestamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)slot);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// 4 waves per block, 8 independent accumulator chains per wave (128 accumulator registers): the MFMA pipe of every SIMD is fed
// back to back by 4 waves per SIMD at 4 blocks per CU.  cycles[block] = s_memtime ticks (shader cycles) of wave 0's loop.
__global__ __launch_bounds__(256) void probe_mfma_f16_kernel(float* __restrict__ sink, int iters,
                                                             unsigned long long* __restrict__ cycles) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    p_f16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (_Float16)(0.001f * (float)((threadIdx.x + 3 * e) & 15));
        b[e] = (_Float16)(0.002f * (float)((threadIdx.x + 5 * e) & 7));
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) sink[0] = s;                       // keeps the loop alive; never true
    if (threadIdx.x == 0 && cycles) cycles[blockIdx.x] = t1 - t0;
}

// flop of one launch = blocks * 4 waves * iters * 8 MFMAs * (2 * 32 * 32 * 16)
extern "C" int semseg_probe_mfma_f16(void* sink, int blocks, int iters, void* cycles, void* stream) {
    if (!sink || blocks <= 0 || iters <= 0) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(probe_mfma_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)sink, iters,
                       (unsigned long long*)cycles);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void probe_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // four independent 16-byte loads in flight per lane
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const float4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
        dst[i] = v0; dst[i + stride] = v1; dst[i + 2 * stride] = v2; dst[i + 3 * stride] = v3;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

extern "C" int semseg_probe_copy(const void* src, void* dst, size_t bytes, void* stream) {
    if (!src || !dst || !aligned16(src) || !aligned16(dst) || (bytes & 15) || bytes == 0) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(probe_copy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst,
                       bytes / 16);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

__global__ void probe_empty_kernel() {}

extern "C" int semseg_probe_empty(void* stream) {
    hipLaunchKernelGGL(probe_empty_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}
