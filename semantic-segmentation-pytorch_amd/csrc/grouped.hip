// Grouped 3x3 convolution (fp32, NHWC) for the 32-group bottleneck convs of ResNeXt (models/resnext.py:30-31): direct
// kernels, one thread per 4 output (forward) / input (data gradient) channels of one pixel, cg = C / g multiply-adds per
// tap and channel; weight gradient as per-chunk partial sums reduced in a fixed order.  These layers are 3 % of ResNeXt's
// FLOPs: the point is to stop paying g-fold redundant MFMA work for them (layers.GroupedConv2d), not peak throughput.
// Per-element code: grouped_math.h (shared with the host emulation test).  Opt-in (SEMSEG_GROUPED_DIRECT=1).
#include "common.h"
#include "grouped_math.h"

__global__ __launch_bounds__(256) void gr_fwd_kernel(const GrGeom g, const float* __restrict__ x, const float* __restrict__ wt,
                                                     float* __restrict__ y, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        gr_fwd_element(g, x, wt, y, i);
}
__global__ __launch_bounds__(256) void gr_dgrad_kernel(const GrGeom g, const float* __restrict__ dy, const float* __restrict__ wt,
                                                       float* __restrict__ dx, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        gr_dgrad_element(g, dy, wt, dx, i);
}
__global__ __launch_bounds__(256) void gr_wgrad_partial_kernel(const GrGeom g, const float* __restrict__ x,
                                                               const float* __restrict__ dy, float* __restrict__ partial,
                                                               int rows_per_chunk, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        gr_wgrad_partial_element(g, x, dy, partial, rows_per_chunk, i);
}
__global__ __launch_bounds__(256) void gr_wgrad_finish_kernel(long slab, int chunks, const float* __restrict__ partial,
                                                              float* __restrict__ dwt) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < slab; i += (long)gridDim.x * blockDim.x)
        gr_wgrad_finish_element(slab, chunks, partial, dwt, i);
}

static inline unsigned gr_blocks(long total) {
    long b = (total + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (unsigned)b;
}

static int gr_geom(GrGeom& g, int N, int H, int W, int C, int K, int groups, int stride, int pad, int dil, int x_ld, int y_ld) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || groups <= 0 || (C % groups) || (K % groups) || ((C / groups) % 4) ||
        ((K / groups) % 4) || stride <= 0 || dil <= 0 || pad < 0 || x_ld < C || y_ld < K)
        return SEMSEG_EINVAL;
    g.N = N; g.H = H; g.W = W; g.C = C; g.K = K; g.groups = groups; g.stride = stride; g.pad = pad; g.dil = dil;
    g.x_ld = x_ld; g.y_ld = y_ld;
    g.OH = (H + 2 * pad - dil * 2 - 1) / stride + 1;
    g.OW = (W + 2 * pad - dil * 2 - 1) / stride + 1;
    return (g.OH > 0 && g.OW > 0) ? 0 : SEMSEG_EINVAL;
}

// output pixels per partial-sum chunk: ~64 K threads over (chunk, k, tap, channel quad), at least 32 pixels per chunk
static int gr_rows_per_chunk(const GrGeom& g) {
    const long P = (long)g.N * g.OH * g.OW;
    const long per_chunk = (long)g.K * 9 * (g.C / g.groups / 4);
    long chunks = (65536 + per_chunk - 1) / per_chunk;
    if (chunks > (P + 31) / 32) chunks = (P + 31) / 32;
    if (chunks < 1) chunks = 1;
    return (int)((P + chunks - 1) / chunks);
}

extern "C" size_t semseg_grouped3x3_workspace_bytes(int N, int H, int W, int C, int K, int groups, int stride, int pad, int dil) {
    GrGeom g;
    if (gr_geom(g, N, H, W, C, K, groups, stride, pad, dil, C, K)) return 0;
    const long P = (long)N * g.OH * g.OW;
    const int rpc = gr_rows_per_chunk(g);
    return (size_t)((P + rpc - 1) / rpc) * K * 9 * (C / groups) * sizeof(float);
}

extern "C" int semseg_grouped3x3_fwd(const float* x, int x_ld, const float* w_taps, float* y, int y_ld, int N, int H, int W, int C,
                                     int K, int groups, int stride, int pad, int dil, void* stream) {
    GrGeom g;
    if (!x || !w_taps || !y || gr_geom(g, N, H, W, C, K, groups, stride, pad, dil, x_ld, y_ld)) return SEMSEG_EINVAL;
    const long total = (long)N * g.OH * g.OW * (K / 4);
    hipLaunchKernelGGL(gr_fwd_kernel, dim3(gr_blocks(total)), dim3(256), 0, (hipStream_t)stream, g, x, w_taps, y, total);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

extern "C" int semseg_grouped3x3_dgrad(const float* dy, int dy_ld, const float* w_taps, float* dx, int dx_ld, int N, int H, int W,
                                       int C, int K, int groups, int stride, int pad, int dil, void* stream) {
    GrGeom g;
    if (!dy || !w_taps || !dx || gr_geom(g, N, H, W, C, K, groups, stride, pad, dil, dx_ld, dy_ld)) return SEMSEG_EINVAL;
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(gr_dgrad_kernel, dim3(gr_blocks(total)), dim3(256), 0, (hipStream_t)stream, g, dy, w_taps, dx, total);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

extern "C" int semseg_grouped3x3_wgrad(const float* x, int x_ld, const float* dy, int dy_ld, float* dw_taps, int N, int H, int W,
                                       int C, int K, int groups, int stride, int pad, int dil, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    GrGeom g;
    if (!x || !dy || !dw_taps || gr_geom(g, N, H, W, C, K, groups, stride, pad, dil, x_ld, dy_ld)) return SEMSEG_EINVAL;
    const long P = (long)N * g.OH * g.OW;
    const int rpc = gr_rows_per_chunk(g);
    const int chunks = (int)((P + rpc - 1) / rpc);
    const long slab = (long)K * 9 * (C / groups);
    if (!workspace || workspace_bytes < (size_t)chunks * slab * sizeof(float)) return SEMSEG_EWORKSPACE;
    float* partial = (float*)workspace;
    const long t1 = (long)chunks * slab / 4;
    hipLaunchKernelGGL(gr_wgrad_partial_kernel, dim3(gr_blocks(t1)), dim3(256), 0, (hipStream_t)stream, g, x, dy, partial, rpc, t1);
    SEMSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(gr_wgrad_finish_kernel, dim3(gr_blocks(slab)), dim3(256), 0, (hipStream_t)stream, slab, chunks,
                       (const float*)partial, dw_taps);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}
