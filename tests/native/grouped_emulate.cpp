// TEST INFRASTRUCTURE: host build of the per-element code of csrc/grouped.hip (grouped_math.h) behind the same C ABI, same
// index ranges and chunking (see depthwise_emulate.cpp).  Never shipped, never loaded by the product.
#include <stddef.h>
#include "grouped_math.h"

static int geom(GrGeom& g, int N, int H, int W, int C, int K, int groups, int stride, int pad, int dil, int x_ld, int y_ld) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || groups <= 0 || (C % groups) || (K % groups) || ((C / groups) % 4) ||
        ((K / groups) % 4) || stride <= 0 || dil <= 0 || pad < 0 || x_ld < C || y_ld < K)
        return -1;
    g.N = N; g.H = H; g.W = W; g.C = C; g.K = K; g.groups = groups; g.stride = stride; g.pad = pad; g.dil = dil;
    g.x_ld = x_ld; g.y_ld = y_ld;
    g.OH = (H + 2 * pad - dil * 2 - 1) / stride + 1;
    g.OW = (W + 2 * pad - dil * 2 - 1) / stride + 1;
    return (g.OH > 0 && g.OW > 0) ? 0 : -1;
}
static int rows_per_chunk(const GrGeom& g) {
    const long P = (long)g.N * g.OH * g.OW;
    const long per_chunk = (long)g.K * 9 * (g.C / g.groups / 4);
    long chunks = (65536 + per_chunk - 1) / per_chunk;
    if (chunks > (P + 31) / 32) chunks = (P + 31) / 32;
    if (chunks < 1) chunks = 1;
    return (int)((P + chunks - 1) / chunks);
}

extern "C" size_t semseg_grouped3x3_workspace_bytes(int N, int H, int W, int C, int K, int groups, int stride, int pad, int dil) {
    GrGeom g;
    if (geom(g, N, H, W, C, K, groups, stride, pad, dil, C, K)) return 0;
    const long P = (long)N * g.OH * g.OW;
    const int rpc = rows_per_chunk(g);
    return (size_t)((P + rpc - 1) / rpc) * K * 9 * (C / groups) * sizeof(float);
}
extern "C" int semseg_grouped3x3_fwd(const float* x, int x_ld, const float* w_taps, float* y, int y_ld, int N, int H, int W, int C,
                                     int K, int groups, int stride, int pad, int dil, void*) {
    GrGeom g;
    if (geom(g, N, H, W, C, K, groups, stride, pad, dil, x_ld, y_ld)) return -1;
    for (long i = 0; i < (long)N * g.OH * g.OW * (K / 4); ++i) gr_fwd_element(g, x, w_taps, y, i);
    return 0;
}
extern "C" int semseg_grouped3x3_dgrad(const float* dy, int dy_ld, const float* w_taps, float* dx, int dx_ld, int N, int H, int W,
                                       int C, int K, int groups, int stride, int pad, int dil, void*) {
    GrGeom g;
    if (geom(g, N, H, W, C, K, groups, stride, pad, dil, dx_ld, dy_ld)) return -1;
    for (long i = 0; i < (long)N * H * W * (C / 4); ++i) gr_dgrad_element(g, dy, w_taps, dx, i);
    return 0;
}
extern "C" int semseg_grouped3x3_wgrad(const float* x, int x_ld, const float* dy, int dy_ld, float* dw_taps, int N, int H, int W,
                                       int C, int K, int groups, int stride, int pad, int dil, void* workspace,
                                       size_t workspace_bytes, void*) {
    GrGeom g;
    if (geom(g, N, H, W, C, K, groups, stride, pad, dil, x_ld, dy_ld)) return -1;
    const long P = (long)N * g.OH * g.OW;
    const int rpc = rows_per_chunk(g);
    const int chunks = (int)((P + rpc - 1) / rpc);
    const long slab = (long)K * 9 * (C / groups);
    if (!workspace || workspace_bytes < (size_t)chunks * slab * sizeof(float)) return -2;
    float* partial = (float*)workspace;
    for (long i = 0; i < (long)chunks * slab / 4; ++i) gr_wgrad_partial_element(g, x, dy, partial, rpc, i);
    for (long i = 0; i < slab; ++i) gr_wgrad_finish_element(slab, chunks, partial, dw_taps, i);
    return 0;
}
