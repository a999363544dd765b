// Per-element arithmetic of the depthwise 3x3 convolution kernels (depthwise.hip) as plain inline functions, so that
// tests/native/depthwise_emulate.cpp runs the SAME code on the host (see input_pipeline_math.h for the pattern).
// Replaces nn.Conv2d(C, C, 3, stride, pad, dilation, groups=C, bias=False) of the reference's MobileNetV2
// (models/mobilenet.py:48,60 after models.py:297-311): forward, data gradient (gather form), weight gradient (per-chunk
// partial sums + ordered finish: deterministic).  fp32, NHWC, 4 channels per element; weights tap-major [9][C].
#pragma once
#include <stdint.h>

#ifndef SEMSEG_HD
#ifdef __HIPCC__
#define SEMSEG_HD __host__ __device__ __forceinline__
#else
#define SEMSEG_HD static inline
#endif
#endif

struct DwGeom {
    int N, H, W, C, OH, OW, stride, pad, dil;
    int x_ld, y_ld;          // pixel pitch (floats) of the input-side and output-side tensors
};

// y[p][c..c+3]: idx = p * (C/4) + quad, p = (n * OH + oh) * OW + ow
SEMSEG_HD void dw_fwd_element(const DwGeom& g, const float* x, const float* wt, float* y, long idx) {
    const int qpr = g.C >> 2;
    const long p = idx / qpr;
    const int c = (int)(idx - p * qpr) << 2;
    const int ow = (int)(p % g.OW);
    const long t = p / g.OW;
    const int oh = (int)(t % g.OH);
    const int n = (int)(t / g.OH);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int r = 0; r < 3; ++r) {
        const int ih = oh * g.stride - g.pad + r * g.dil;
        if (ih < 0 || ih >= g.H) continue;
        for (int s = 0; s < 3; ++s) {
            const int iw = ow * g.stride - g.pad + s * g.dil;
            if (iw < 0 || iw >= g.W) continue;
            const float* xp = x + ((long)(n * g.H + ih) * g.W + iw) * g.x_ld + c;
            const float* wp = wt + (long)(r * 3 + s) * g.C + c;
            a0 += xp[0] * wp[0]; a1 += xp[1] * wp[1]; a2 += xp[2] * wp[2]; a3 += xp[3] * wp[3];
        }
    }
    float* o = y + p * g.y_ld + c;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
}

// dx[p][c..c+3]: idx = p * (C/4) + quad, p = (n * H + ih) * W + iw; gathers the output pixels whose tap (r, s) reads (ih, iw)
SEMSEG_HD void dw_dgrad_element(const DwGeom& g, const float* dy, const float* wt, float* dx, long idx) {
    const int qpr = g.C >> 2;
    const long p = idx / qpr;
    const int c = (int)(idx - p * qpr) << 2;
    const int iw = (int)(p % g.W);
    const long t = p / g.W;
    const int ih = (int)(t % g.H);
    const int n = (int)(t / g.H);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int r = 0; r < 3; ++r) {
        const int th = ih + g.pad - r * g.dil;
        if (th < 0 || th % g.stride) continue;
        const int oh = th / g.stride;
        if (oh >= g.OH) continue;
        for (int s = 0; s < 3; ++s) {
            const int tw = iw + g.pad - s * g.dil;
            if (tw < 0 || tw % g.stride) continue;
            const int ow = tw / g.stride;
            if (ow >= g.OW) continue;
            const float* gp = dy + ((long)(n * g.OH + oh) * g.OW + ow) * g.y_ld + c;
            const float* wp = wt + (long)(r * 3 + s) * g.C + c;
            a0 += gp[0] * wp[0]; a1 += gp[1] * wp[1]; a2 += gp[2] * wp[2]; a3 += gp[3] * wp[3];
        }
    }
    float* o = dx + p * g.x_ld + c;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
}

// partial[chunk][tap][c..c+3] = sum over the chunk's output pixels of dy * x(tap): idx = chunk * (C/4) + quad
SEMSEG_HD void dw_wgrad_partial_element(const DwGeom& g, const float* x, const float* dy, float* partial, int rows_per_chunk,
                                        long idx) {
    const int qpr = g.C >> 2;
    const long chunk = idx / qpr;
    const int c = (int)(idx - chunk * qpr) << 2;
    const long P = (long)g.N * g.OH * g.OW;
    const long p0 = chunk * rows_per_chunk;
    const long p1 = p0 + rows_per_chunk < P ? p0 + rows_per_chunk : P;
    float acc[9][4];
    for (int t = 0; t < 9; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
    for (long p = p0; p < p1; ++p) {
        const int ow = (int)(p % g.OW);
        const long tt = p / g.OW;
        const int oh = (int)(tt % g.OH);
        const int n = (int)(tt / g.OH);
        const float* gp = dy + p * g.y_ld + c;
        const float g0 = gp[0], g1 = gp[1], g2 = gp[2], g3 = gp[3];
        for (int r = 0; r < 3; ++r) {
            const int ih = oh * g.stride - g.pad + r * g.dil;
            if (ih < 0 || ih >= g.H) continue;
            for (int s = 0; s < 3; ++s) {
                const int iw = ow * g.stride - g.pad + s * g.dil;
                if (iw < 0 || iw >= g.W) continue;
                const float* xp = x + ((long)(n * g.H + ih) * g.W + iw) * g.x_ld + c;
                float* a = acc[r * 3 + s];
                a[0] += g0 * xp[0]; a[1] += g1 * xp[1]; a[2] += g2 * xp[2]; a[3] += g3 * xp[3];
            }
        }
    }
    for (int t = 0; t < 9; ++t) {
        float* o = partial + ((long)chunk * 9 + t) * g.C + c;
        o[0] = acc[t][0]; o[1] = acc[t][1]; o[2] = acc[t][2]; o[3] = acc[t][3];
    }
}

// dwt[tap][c] = sum over chunks (in chunk order) of partial[chunk][tap][c]: idx = tap * C + c
SEMSEG_HD void dw_wgrad_finish_element(int C, int chunks, const float* partial, float* dwt, long idx) {
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += partial[(long)k * 9 * C + idx];
    dwt[idx] = s;
}
