// Depthwise 3x3 convolution (fp32, NHWC): the nn.Conv2d(C, C, 3, groups=C) of MobileNetV2's inverted-residual blocks
// (models/mobilenet.py:48,60; stride / dilation as rewritten by models.py:297-311).  9 MACs per output element: HBM-bound
// streaming kernels, one thread per 4 channels of one pixel; weights tap-major [9][C] (36 KB at C = 960: cache resident).
// The weight gradient is a two-stage fixed-order reduction (per-chunk partials, then the chunks in order): deterministic.
// Per-element code: depthwise_math.h (shared with the host emulation test).
#include "common.h"
#include "depthwise_math.h"

__global__ __launch_bounds__(256) void dw_fwd_kernel(const DwGeom g, const float* __restrict__ x, const float* __restrict__ wt,
                                                     float* __restrict__ y, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        dw_fwd_element(g, x, wt, y, i);
}
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const DwGeom g, const float* __restrict__ dy, const float* __restrict__ wt,
                                                       float* __restrict__ dx, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        dw_dgrad_element(g, dy, wt, dx, i);
}
__global__ __launch_bounds__(256) void dw_wgrad_partial_kernel(const DwGeom g, const float* __restrict__ x,
                                                               const float* __restrict__ dy, float* __restrict__ partial,
                                                               int rows_per_chunk, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        dw_wgrad_partial_element(g, x, dy, partial, rows_per_chunk, i);
}
__global__ __launch_bounds__(256) void dw_wgrad_finish_kernel(int C, int chunks, const float* __restrict__ partial,
                                                              float* __restrict__ dwt, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        dw_wgrad_finish_element(C, chunks, partial, dwt, i);
}

static inline unsigned dw_blocks(long total) {
    long b = (total + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (unsigned)b;
}

static int dw_geom(DwGeom& g, int N, int H, int W, int C, int stride, int pad, int dil, int x_ld, int y_ld) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) || stride <= 0 || dil <= 0 || pad < 0 || x_ld < C || y_ld < C) return SEMSEG_EINVAL;
    g.N = N; g.H = H; g.W = W; g.C = C; g.stride = stride; g.pad = pad; g.dil = dil; g.x_ld = x_ld; g.y_ld = y_ld;
    g.OH = (H + 2 * pad - dil * 2 - 1) / stride + 1;
    g.OW = (W + 2 * pad - dil * 2 - 1) / stride + 1;
    return (g.OH > 0 && g.OW > 0) ? 0 : SEMSEG_EINVAL;
}

// pixels per partial-sum chunk of the weight gradient: enough chunks to fill the chip with (chunks * C/4) threads
static int dw_rows_per_chunk(const DwGeom& g) {
    const long P = (long)g.N * g.OH * g.OW;
    long chunks = (65536 + g.C / 4 - 1) / (g.C / 4);
    if (chunks > P) chunks = P;
    if (chunks < 1) chunks = 1;
    return (int)((P + chunks - 1) / chunks);
}

extern "C" size_t semseg_depthwise3x3_workspace_bytes(int N, int H, int W, int C, int stride, int pad, int dil) {
    DwGeom g;
    if (dw_geom(g, N, H, W, C, stride, pad, dil, C, C)) return 0;
    const long P = (long)N * g.OH * g.OW;
    const int rpc = dw_rows_per_chunk(g);
    return (size_t)((P + rpc - 1) / rpc) * 9 * C * sizeof(float);
}

extern "C" int semseg_depthwise3x3_fwd(const float* x, int x_ld, const float* w_taps, float* y, int y_ld, int N, int H, int W,
                                       int C, int stride, int pad, int dil, void* stream) {
    DwGeom g;
    if (!x || !w_taps || !y || dw_geom(g, N, H, W, C, stride, pad, dil, x_ld, y_ld)) return SEMSEG_EINVAL;
    const long total = (long)N * g.OH * g.OW * (C / 4);
    hipLaunchKernelGGL(dw_fwd_kernel, dim3(dw_blocks(total)), dim3(256), 0, (hipStream_t)stream, g, x, w_taps, y, total);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

extern "C" int semseg_depthwise3x3_dgrad(const float* dy, int dy_ld, const float* w_taps, float* dx, int dx_ld, int N, int H,
                                         int W, int C, int stride, int pad, int dil, void* stream) {
    DwGeom g;
    if (!dy || !w_taps || !dx || dw_geom(g, N, H, W, C, stride, pad, dil, dx_ld, dy_ld)) return SEMSEG_EINVAL;
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(dw_dgrad_kernel, dim3(dw_blocks(total)), dim3(256), 0, (hipStream_t)stream, g, dy, w_taps, dx, total);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

extern "C" int semseg_depthwise3x3_wgrad(const float* x, int x_ld, const float* dy, int dy_ld, float* dw_taps, int N, int H, int W,
                                         int C, int stride, int pad, int dil, void* workspace, size_t workspace_bytes,
                                         void* stream) {
    DwGeom g;
    if (!x || !dy || !dw_taps || dw_geom(g, N, H, W, C, stride, pad, dil, x_ld, dy_ld)) return SEMSEG_EINVAL;
    const long P = (long)N * g.OH * g.OW;
    const int rpc = dw_rows_per_chunk(g);
    const int chunks = (int)((P + rpc - 1) / rpc);
    if (!workspace || workspace_bytes < (size_t)chunks * 9 * C * sizeof(float)) return SEMSEG_EWORKSPACE;
    float* partial = (float*)workspace;
    const long t1 = (long)chunks * (C / 4);
    hipLaunchKernelGGL(dw_wgrad_partial_kernel, dim3(dw_blocks(t1)), dim3(256), 0, (hipStream_t)stream, g, x, dy, partial, rpc, t1);
    SEMSEG_LAUNCH_CHECK();
    const long t2 = (long)9 * C;
    hipLaunchKernelGGL(dw_wgrad_finish_kernel, dim3(dw_blocks(t2)), dim3(256), 0, (hipStream_t)stream, C, chunks,
                       (const float*)partial, dw_taps, t2);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}
