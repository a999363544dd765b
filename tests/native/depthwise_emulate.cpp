// TEST INFRASTRUCTURE: host build of the per-element code of csrc/depthwise.hip (depthwise_math.h) behind the same C ABI,
// with the same index ranges and the same chunking of the weight-gradient reduction, so that the depthwise kernels and the
// autograd wrapper above them can be checked against torch's grouped convolution without a GPU
// (tests/test_depthwise_cpu.py).  Never shipped, never loaded by the product.
#include <stddef.h>
#include "depthwise_math.h"

static int geom(DwGeom& g, int N, int H, int W, int C, int stride, int pad, int dil, int x_ld, int y_ld) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) || stride <= 0 || dil <= 0 || pad < 0 || x_ld < C || y_ld < C) return -1;
    g.N = N; g.H = H; g.W = W; g.C = C; g.stride = stride; g.pad = pad; g.dil = dil; g.x_ld = x_ld; g.y_ld = y_ld;
    g.OH = (H + 2 * pad - dil * 2 - 1) / stride + 1;
    g.OW = (W + 2 * pad - dil * 2 - 1) / stride + 1;
    return (g.OH > 0 && g.OW > 0) ? 0 : -1;
}
static int rows_per_chunk(const DwGeom& g) {
    const long P = (long)g.N * g.OH * g.OW;
    long chunks = (65536 + g.C / 4 - 1) / (g.C / 4);
    if (chunks > P) chunks = P;
    if (chunks < 1) chunks = 1;
    return (int)((P + chunks - 1) / chunks);
}

extern "C" size_t semseg_depthwise3x3_workspace_bytes(int N, int H, int W, int C, int stride, int pad, int dil) {
    DwGeom g;
    if (geom(g, N, H, W, C, stride, pad, dil, C, C)) return 0;
    const long P = (long)N * g.OH * g.OW;
    const int rpc = rows_per_chunk(g);
    return (size_t)((P + rpc - 1) / rpc) * 9 * C * sizeof(float);
}
extern "C" int semseg_depthwise3x3_fwd(const float* x, int x_ld, const float* w_taps, float* y, int y_ld, int N, int H, int W,
                                       int C, int stride, int pad, int dil, void*) {
    DwGeom g;
    if (geom(g, N, H, W, C, stride, pad, dil, x_ld, y_ld)) return -1;
    for (long i = 0; i < (long)N * g.OH * g.OW * (C / 4); ++i) dw_fwd_element(g, x, w_taps, y, i);
    return 0;
}
extern "C" int semseg_depthwise3x3_dgrad(const float* dy, int dy_ld, const float* w_taps, float* dx, int dx_ld, int N, int H,
                                         int W, int C, int stride, int pad, int dil, void*) {
    DwGeom g;
    if (geom(g, N, H, W, C, stride, pad, dil, dx_ld, dy_ld)) return -1;
    for (long i = 0; i < (long)N * H * W * (C / 4); ++i) dw_dgrad_element(g, dy, w_taps, dx, i);
    return 0;
}
extern "C" int semseg_depthwise3x3_wgrad(const float* x, int x_ld, const float* dy, int dy_ld, float* dw_taps, int N, int H, int W,
                                         int C, int stride, int pad, int dil, void* workspace, size_t workspace_bytes, void*) {
    DwGeom g;
    if (geom(g, N, H, W, C, stride, pad, dil, x_ld, dy_ld)) return -1;
    const long P = (long)N * g.OH * g.OW;
    const int rpc = rows_per_chunk(g);
    const int chunks = (int)((P + rpc - 1) / rpc);
    if (!workspace || workspace_bytes < (size_t)chunks * 9 * C * sizeof(float)) return -2;
    float* partial = (float*)workspace;
    for (long i = 0; i < (long)chunks * (C / 4); ++i) dw_wgrad_partial_element(g, x, dy, partial, rpc, i);
    for (long i = 0; i < (long)9 * C; ++i) dw_wgrad_finish_element(C, chunks, partial, dw_taps, i);
    return 0;
}
