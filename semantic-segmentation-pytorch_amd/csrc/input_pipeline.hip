// On-device assembly of a training batch from decoded uint8 images: the contract of
// mit_semseg/dataset.py:110-199 (TrainDataset.__getitem__) -- Pillow BILINEAR resize of the image, NEAREST resize of the
// label map (x2, through a zero canvas), flip, ToTensor + Normalize, label - 1, placement into the zero-padded batch -- with
// the JPEG/PNG decode left on the host.  Integer work, bit-exact against Pillow (tests/golden/input_golden.npz).
//   host side (plain C++, callable without a GPU): tap tables of Pillow's resampler and the NEAREST index tables
//   device side: three streaming kernels per sample (horizontal taps, vertical taps + normalise, label gather)
// The per-element arithmetic lives in input_pipeline_math.h so that the host emulation test runs the same code.
#include "common.h"
#include "input_pipeline_math.h"
#include <math.h>

// ---- Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc, bilinear filter, full source box ----------------------
extern "C" int semseg_input_resample_ksize(int in_size, int out_size) {
    if (in_size <= 0 || out_size <= 0) return 0;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    return (int)ceil(support) * 2 + 1;
}

extern "C" int semseg_input_resample_coeffs(int in_size, int out_size, int32_t* bounds_host, int32_t* kk_host) {
    if (in_size <= 0 || out_size <= 0 || !bounds_host || !kk_host) return SEMSEG_EINVAL;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    const double ss = 1.0 / filterscale;
    double* w = new double[ksize];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < ksize; ++x) w[x] = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            const double v = a < 1.0 ? 1.0 - a : 0.0;
            w[x] = v;
            ww += v;
        }
        if (ww != 0.0)
            for (int x = 0; x < xmax; ++x) w[x] /= ww;
        int32_t* k = kk_host + (size_t)xx * ksize;
        for (int x = 0; x < ksize; ++x)
            k[x] = w[x] < 0 ? (int32_t)(-0.5 + w[x] * (1 << PIL_PRECISION_BITS)) : (int32_t)(0.5 + w[x] * (1 << PIL_PRECISION_BITS));
        bounds_host[2 * xx] = xmin;
        bounds_host[2 * xx + 1] = xmax;
    }
    delete[] w;
    return 0;
}

// ---- Pillow Geometry.c ImagingScaleAffine (NEAREST): the source coordinate advances by repeated double additions -----------
extern "C" int semseg_input_nearest_table(int in_size, int out_size, int32_t* tab_host) {
    if (in_size <= 0 || out_size <= 0 || !tab_host) return SEMSEG_EINVAL;
    const double a = (double)in_size / (double)out_size;
    double xo = a * 0.5;
    for (int x = 0; x < out_size; ++x) {
        const int xin = xo < 0.0 ? -1 : (int)xo;
        tab_host[x] = (xin >= 0 && xin < in_size) ? xin : -1;
        xo += a;
    }
    return 0;
}

// ---- kernels --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void input_resample_h_kernel(const ResampleH a) {
    const long total = (long)a.H * a.ow;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        resample_h_element(a, i);
}
__global__ __launch_bounds__(256) void input_resample_v_norm_kernel(const ResampleVNorm a) {
    const long total = (long)a.oh * a.ow;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        resample_v_norm_element(a, i);
}
__global__ __launch_bounds__(256) void input_label_gather_kernel(const LabelGather a) {
    const long total = (long)a.lh * a.lw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        label_gather_element(a, i);
}

static inline unsigned input_blocks(long total) {
    long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

extern "C" int semseg_input_resample_h_u8(const uint8_t* src, int H, int W, int flip, const int32_t* bounds, const int32_t* kk,
                                          int ksize, uint8_t* tmp, int ow, void* stream) {
    if (!src || !bounds || !kk || !tmp || H <= 0 || W <= 0 || ow <= 0 || ksize <= 0) return SEMSEG_EINVAL;
    ResampleH a;
    a.src = src; a.tmp = tmp; a.bounds = bounds; a.kk = kk;
    a.H = H; a.W = W; a.ow = ow; a.ksize = ksize; a.flip = flip ? 1 : 0;
    hipLaunchKernelGGL(input_resample_h_kernel, dim3(input_blocks((long)H * ow)), dim3(256), 0, (hipStream_t)stream, a);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

extern "C" int semseg_input_resample_v_normalize(const uint8_t* tmp, int H, int ow, const int32_t* bounds, const int32_t* kk,
                                                 int ksize, int oh, const float* mean_std_host, float* dst, int BW, void* stream) {
    if (!tmp || !bounds || !kk || !dst || !mean_std_host || H <= 0 || ow <= 0 || oh <= 0 || ksize <= 0 || BW < ow)
        return SEMSEG_EINVAL;
    ResampleVNorm a;
    a.tmp = tmp; a.dst = dst; a.bounds = bounds; a.kk = kk;
    a.H = H; a.ow = ow; a.oh = oh; a.ksize = ksize; a.BW = BW;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean_std_host[c]; a.std[c] = mean_std_host[3 + c]; }
    hipLaunchKernelGGL(input_resample_v_norm_kernel, dim3(input_blocks((long)oh * ow)), dim3(256), 0, (hipStream_t)stream, a);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

extern "C" int semseg_input_label_gather(const uint8_t* src, int W, const int32_t* ytab, const int32_t* xtab, int lh, int lw,
                                         int64_t* dst, int LW, void* stream) {
    if (!src || !ytab || !xtab || !dst || W <= 0 || lh <= 0 || lw <= 0 || LW < lw) return SEMSEG_EINVAL;
    LabelGather a;
    a.src = src; a.dst = dst; a.ytab = ytab; a.xtab = xtab;
    a.W = W; a.lh = lh; a.lw = lw; a.LW = LW;
    hipLaunchKernelGGL(input_label_gather_kernel, dim3(input_blocks((long)lh * lw)), dim3(256), 0, (hipStream_t)stream, a);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}
