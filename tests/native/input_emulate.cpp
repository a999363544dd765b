// TEST INFRASTRUCTURE: runs the per-element code of csrc/input_pipeline.hip (input_pipeline_math.h, the very functions the
// kernels call) on the HOST, with the C ABI of the three kernel entry points, so that the input pipeline -- tables, index
// decomposition, fixed-point taps, normalisation, label gather, and the Python assembly above it -- can be checked against
// the Pillow / reference goldens in a container without a GPU (tests/test_input_pipeline_cpu.py).  Never shipped, never
// loaded by the product.
#include "input_pipeline_math.h"

extern "C" int semseg_input_resample_h_u8(const uint8_t* src, int H, int W, int flip, const int32_t* bounds, const int32_t* kk,
                                          int ksize, uint8_t* tmp, int ow, void*) {
    ResampleH a;
    a.src = src; a.tmp = tmp; a.bounds = bounds; a.kk = kk;
    a.H = H; a.W = W; a.ow = ow; a.ksize = ksize; a.flip = flip ? 1 : 0;
    for (long i = 0; i < (long)H * ow; ++i) resample_h_element(a, i);
    return 0;
}

extern "C" int semseg_input_resample_v_normalize(const uint8_t* tmp, int H, int ow, const int32_t* bounds, const int32_t* kk,
                                                 int ksize, int oh, const float* mean_std_host, float* dst, int BW, void*) {
    ResampleVNorm a;
    a.tmp = tmp; a.dst = dst; a.bounds = bounds; a.kk = kk;
    a.H = H; a.ow = ow; a.oh = oh; a.ksize = ksize; a.BW = BW;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean_std_host[c]; a.std[c] = mean_std_host[3 + c]; }
    for (long i = 0; i < (long)oh * ow; ++i) resample_v_norm_element(a, i);
    return 0;
}

extern "C" int semseg_input_label_gather(const uint8_t* src, int W, const int32_t* ytab, const int32_t* xtab, int lh, int lw,
                                         int64_t* dst, int LW, void*) {
    LabelGather a;
    a.src = src; a.dst = dst; a.ytab = ytab; a.xtab = xtab;
    a.W = W; a.lh = lh; a.lw = lw; a.LW = LW;
    for (long i = 0; i < (long)lh * lw; ++i) label_gather_element(a, i);
    return 0;
}
