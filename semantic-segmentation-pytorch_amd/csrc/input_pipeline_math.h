// Per-element arithmetic of the input-pipeline kernels (input_pipeline.hip), kept in a header of plain inline functions so
// that tests/native/input_emulate.cpp can run the SAME code on the host (no GPU) against the Pillow goldens.
// Contract: mit_semseg/dataset.py:110-199 of the reference (TrainDataset.__getitem__): PIL BILINEAR resize (Pillow
// Resample.c: 22-bit fixed-point taps, horizontal then vertical pass, clip to 8 bits after each), PIL NEAREST resize of the
// label map (twice: to the image size, then by the segmentation rate from a zero canvas), ToTensor + Normalize, label - 1.
#pragma once
#include <stdint.h>

#ifndef SEMSEG_HD
#ifdef __HIPCC__
#define SEMSEG_HD __host__ __device__ __forceinline__
#else
#define SEMSEG_HD static inline
#endif
#endif

constexpr int PIL_PRECISION_BITS = 32 - 8 - 2;

SEMSEG_HD uint8_t pil_clip8(int32_t acc) {
    const int32_t v = acc >> PIL_PRECISION_BITS;            // arithmetic shift, as Pillow's lookup table index
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: element idx = (y * ow + xx); src [H][W][3] (column W-1-x when flip), tmp [H][ow][3]
struct ResampleH {
    const uint8_t* src;
    uint8_t* tmp;
    const int32_t* bounds;      // [ow][2] = (first source column, tap count)
    const int32_t* kk;          // [ow][ksize]
    int H, W, ow, ksize, flip;
};
SEMSEG_HD void resample_h_element(const ResampleH& a, long idx) {
    const int xx = (int)(idx % a.ow);
    const int y = (int)(idx / a.ow);
    const int xmin = a.bounds[2 * xx], n = a.bounds[2 * xx + 1];
    const int32_t* k = a.kk + (long)xx * a.ksize;
    int32_t s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    const uint8_t* row = a.src + (long)y * a.W * 3;
    for (int x = 0; x < n; ++x) {
        const int col = a.flip ? a.W - 1 - (xmin + x) : xmin + x;
        const uint8_t* p = row + (long)col * 3;
        s0 += (int32_t)p[0] * k[x];
        s1 += (int32_t)p[1] * k[x];
        s2 += (int32_t)p[2] * k[x];
    }
    uint8_t* o = a.tmp + idx * 3;
    o[0] = pil_clip8(s0); o[1] = pil_clip8(s1); o[2] = pil_clip8(s2);
}

// vertical pass + ToTensor + Normalize: element idx = (yy * ow + x); tmp [H][ow][3] -> dst NHWC float, pixel pitch 3,
// row pitch BW * 3 (the sample's slice of the zero-initialised batch tensor)
struct ResampleVNorm {
    const uint8_t* tmp;
    float* dst;
    const int32_t* bounds;      // [oh][2]
    const int32_t* kk;          // [oh][ksize]
    int H, ow, oh, ksize, BW;
    float mean[3], std[3];
};
SEMSEG_HD void resample_v_norm_element(const ResampleVNorm& a, long idx) {
    const int x = (int)(idx % a.ow);
    const int yy = (int)(idx / a.ow);
    const int ymin = a.bounds[2 * yy], n = a.bounds[2 * yy + 1];
    const int32_t* k = a.kk + (long)yy * a.ksize;
    int32_t s[3] = {1 << (PIL_PRECISION_BITS - 1), 1 << (PIL_PRECISION_BITS - 1), 1 << (PIL_PRECISION_BITS - 1)};
    for (int y = 0; y < n; ++y) {
        const uint8_t* p = a.tmp + ((long)(ymin + y) * a.ow + x) * 3;
        s[0] += (int32_t)p[0] * k[y];
        s[1] += (int32_t)p[1] * k[y];
        s[2] += (int32_t)p[2] * k[y];
    }
    float* o = a.dst + ((long)yy * a.BW + x) * 3;
    for (int c = 0; c < 3; ++c) {
        const float t = (float)pil_clip8(s[c]) / 255.0f;         // np.float32(img) / 255.   (dataset.py:55)
        o[c] = (t - a.mean[c]) / a.std[c];                       // Normalize: sub_(mean).div_(std)
    }
}

// label path: both NEAREST resizes and the zero canvas are folded into two index tables (host side); element idx =
// (y * lw + x); src uint8 [H][W]; dst int64 with row pitch LW
struct LabelGather {
    const uint8_t* src;
    int64_t* dst;
    const int32_t* ytab;        // [lh] source row or -1 (canvas / outside: value 0 before the shift)
    const int32_t* xtab;        // [lw] source column (flip already applied) or -1
    int W, lh, lw, LW;
};
SEMSEG_HD void label_gather_element(const LabelGather& a, long idx) {
    const int x = (int)(idx % a.lw);
    const int y = (int)(idx / a.lw);
    const int ys = a.ytab[y], xs = a.xtab[x];
    const int v = (ys >= 0 && xs >= 0) ? (int)a.src[(long)ys * a.W + xs] : 0;
    a.dst[(long)y * a.LW + x] = (int64_t)v - 1;                  // segm_transform: .long() - 1   (dataset.py:62)
}
