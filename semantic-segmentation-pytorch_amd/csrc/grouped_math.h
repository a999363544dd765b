// Per-element arithmetic of the grouped 3x3 convolution kernels (grouped.hip), shared with the host emulation test
// (tests/native/grouped_emulate.cpp; pattern of depthwise_math.h).  Replaces nn.Conv2d(C, K, 3, stride, padding = 1,
// groups = g, bias = False) of the reference's ResNeXt bottleneck (models/resnext.py:30-31: g = 32, 4 .. 32 channels per
// group) and its autograd.  fp32, NHWC; weights in the reference's own grouped shape re-ordered tap-innermost-channel:
// wt[k][tap][ci], ci < cg = C / g.  One element = 4 consecutive output (fwd) / input (dgrad) channels of one pixel -- they
// lie in one group because cg % 4 == 0 and kg % 4 == 0.
#pragma once
#include <stdint.h>

#ifndef SEMSEG_HD
#ifdef __HIPCC__
#define SEMSEG_HD __host__ __device__ __forceinline__
#else
#define SEMSEG_HD static inline
#endif
#endif

struct GrGeom {
    int N, H, W, C, K, groups, OH, OW, stride, pad, dil;
    int x_ld, y_ld;
};

// y[p][k..k+3]: idx = p * (K/4) + quad
SEMSEG_HD void gr_fwd_element(const GrGeom& g, const float* x, const float* wt, float* y, long idx) {
    const int qpr = g.K >> 2, cg = g.C / g.groups, kg = g.K / g.groups;
    const long p = idx / qpr;
    const int k = (int)(idx - p * qpr) << 2;
    const int grp = k / kg;
    const int ow = (int)(p % g.OW);
    const long t = p / g.OW;
    const int oh = (int)(t % g.OH);
    const int n = (int)(t / g.OH);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < 3; ++r) {
        const int ih = oh * g.stride - g.pad + r * g.dil;
        if (ih < 0 || ih >= g.H) continue;
        for (int s = 0; s < 3; ++s) {
            const int iw = ow * g.stride - g.pad + s * g.dil;
            if (iw < 0 || iw >= g.W) continue;
            const float* xp = x + ((long)(n * g.H + ih) * g.W + iw) * g.x_ld + grp * cg;
            for (int j = 0; j < 4; ++j) {
                const float* wp = wt + ((long)(k + j) * 9 + r * 3 + s) * cg;
                float acc = a[j];
                for (int ci = 0; ci < cg; ++ci) acc += xp[ci] * wp[ci];
                a[j] = acc;
            }
        }
    }
    float* o = y + p * g.y_ld + k;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
}

// dx[p][c..c+3]: idx = p * (C/4) + quad; gathers over the kg output channels of the group and the taps that read (ih, iw)
SEMSEG_HD void gr_dgrad_element(const GrGeom& g, const float* dy, const float* wt, float* dx, long idx) {
    const int qpr = g.C >> 2, cg = g.C / g.groups, kg = g.K / g.groups;
    const long p = idx / qpr;
    const int c = (int)(idx - p * qpr) << 2;
    const int grp = c / cg, cl = c - grp * cg;
    const int iw = (int)(p % g.W);
    const long t = p / g.W;
    const int ih = (int)(t % g.H);
    const int n = (int)(t / g.H);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
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
            const float* gp = dy + ((long)(n * g.OH + oh) * g.OW + ow) * g.y_ld + grp * kg;
            for (int ko = 0; ko < kg; ++ko) {
                const float gv = gp[ko];
                const float* wp = wt + ((long)(grp * kg + ko) * 9 + r * 3 + s) * cg + cl;
                a[0] += gv * wp[0]; a[1] += gv * wp[1]; a[2] += gv * wp[2]; a[3] += gv * wp[3];
            }
        }
    }
    float* o = dx + p * g.x_ld + c;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
}

// partial[chunk][k][tap][ci..ci+3] over the chunk's output pixels: idx = ((chunk * K + k) * 9 + tap) * (cg/4) + quad
SEMSEG_HD void gr_wgrad_partial_element(const GrGeom& g, const float* x, const float* dy, float* partial, int rows_per_chunk,
                                        long idx) {
    const int cg = g.C / g.groups, kg = g.K / g.groups, qpg = cg >> 2;
    const int q = (int)(idx % qpg);
    long t0 = idx / qpg;
    const int tap = (int)(t0 % 9); t0 /= 9;
    const int k = (int)(t0 % g.K);
    const long chunk = t0 / g.K;
    const int r = tap / 3, s = tap - r * 3;
    const int grp = k / kg;
    const long P = (long)g.N * g.OH * g.OW;
    const long p0 = chunk * rows_per_chunk;
    const long p1 = p0 + rows_per_chunk < P ? p0 + rows_per_chunk : P;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (long p = p0; p < p1; ++p) {
        const int ow = (int)(p % g.OW);
        const long tt = p / g.OW;
        const int oh = (int)(tt % g.OH);
        const int n = (int)(tt / g.OH);
        const int ih = oh * g.stride - g.pad + r * g.dil, iw = ow * g.stride - g.pad + s * g.dil;
        if (ih < 0 || ih >= g.H || iw < 0 || iw >= g.W) continue;
        const float gv = dy[p * g.y_ld + k];
        const float* xp = x + ((long)(n * g.H + ih) * g.W + iw) * g.x_ld + grp * cg + q * 4;
        a[0] += gv * xp[0]; a[1] += gv * xp[1]; a[2] += gv * xp[2]; a[3] += gv * xp[3];
    }
    float* o = partial + idx * 4;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
}

// dwt[k][tap][ci] = sum over chunks in order: idx over K * 9 * cg
SEMSEG_HD void gr_wgrad_finish_element(long slab, int chunks, const float* partial, float* dwt, long idx) {
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += partial[(long)k * slab + idx];
    dwt[idx] = s;
}
