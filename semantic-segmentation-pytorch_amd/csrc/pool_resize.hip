// HBM-bound NHWC side ops of the segmentation hot path (gfx950): max-pool, adaptive average pool,
// bilinear resize (align_corners=False), strided copies / adds for concat and FPN/HRNet fusion,
// layout changes at the API edge.  All are float4-per-lane coalesced along the channel axis; every
// backward is written in GATHER form (each output element owned by exactly one thread, fixed
// summation order) so results are deterministic without atomics.
#include "common.h"
#include "batch.h"
#include <stdlib.h>

static inline int stream_blocks(size_t items) {
    size_t b = ceil_div_sz(items, 256);
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}
#define GRID_STRIDE(i, total) \
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (size_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------ elementwise -----------------
template <bool RELU>
struct add_act_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ a, int a_ld, const float* __restrict__ b, int b_ld,
                               float* __restrict__ out, int out_ld, int P, int C) {
    const int qpr = C / 4;
    const size_t total = (size_t)P * qpr;
    GRID_STRIDE(i, total) {
        const int p = (int)(i / qpr);
        const int c = (int)(i - (size_t)p * qpr) * 4;
        const float4 x = *reinterpret_cast<const float4*>(a + (size_t)p * a_ld + c);
        const float4 y = *reinterpret_cast<const float4*>(b + (size_t)p * b_ld + c);
        float4 o = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<float4*>(out + (size_t)p * out_ld + c) = o;
    }
    }
};

extern "C" int semseg_add_act(const float* a, int a_ld, const float* b, int b_ld, int relu, float* out, int out_ld,
                              int P, int C, void* stream) {
    if (!a || !b || !out || P <= 0 || C <= 0 || (C % 4) || (a_ld % 4) || (b_ld % 4) || (out_ld % 4)) return SEMSEG_EINVAL;
    const int blocks = stream_blocks((size_t)P * (C / 4));
    if (relu) SEMSEG_LAUNCH_BODY((add_act_kernel_body<true>), dim3(blocks), 0, (hipStream_t)stream, a, a_ld, b, b_ld, out, out_ld, P, C);
    else      SEMSEG_LAUNCH_BODY((add_act_kernel_body<false>), dim3(blocks), 0, (hipStream_t)stream, a, a_ld, b, b_ld, out, out_ld, P, C);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

struct relu_bwd_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ dy, int dy_ld, const float* __restrict__ y, int y_ld,
                                float* __restrict__ dx, int dx_ld, int P, int C) {
    const int qpr = C / 4;
    const size_t total = (size_t)P * qpr;
    GRID_STRIDE(i, total) {
        const int p = (int)(i / qpr);
        const int c = (int)(i - (size_t)p * qpr) * 4;
        const float4 g = *reinterpret_cast<const float4*>(dy + (size_t)p * dy_ld + c);
        const float4 v = *reinterpret_cast<const float4*>(y + (size_t)p * y_ld + c);
        const float4 o = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
        *reinterpret_cast<float4*>(dx + (size_t)p * dx_ld + c) = o;
    }
    }
};

extern "C" int semseg_relu_bwd(const float* dy, int dy_ld, const float* y, int y_ld, float* dx, int dx_ld, int P, int C,
                               void* stream) {
    if (!dy || !y || !dx || P <= 0 || C <= 0 || (C % 4) || (dy_ld % 4) || (y_ld % 4) || (dx_ld % 4)) return SEMSEG_EINVAL;
    SEMSEG_LAUNCH_BODY((relu_bwd_kernel_body), dim3(stream_blocks((size_t)P * (C / 4))), 0, (hipStream_t)stream, dy, dy_ld,
                       y, y_ld, dx, dx_ld, P, C);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// nn.ReLU6's upper clamp (mobilenet.py:26,34; the lower clamp runs fused in the BN kernels): y = min(x, cap);
// backward dx = dy * (y < cap) -- hardtanh's open interval, as torch's ReLU6 backward
__global__ void clamp_max_kernel(const float* __restrict__ x, int x_ld, float cap, float* __restrict__ y, int y_ld, int P, int C) {
    const int qpr = C / 4;
    const size_t total = (size_t)P * qpr;
    GRID_STRIDE(i, total) {
        const int p = (int)(i / qpr);
        const int c = (int)(i - (size_t)p * qpr) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)p * x_ld + c);
        *reinterpret_cast<float4*>(y + (size_t)p * y_ld + c) =
            make_float4(fminf(v.x, cap), fminf(v.y, cap), fminf(v.z, cap), fminf(v.w, cap));
    }
}

__global__ void clamp_max_bwd_kernel(const float* __restrict__ dy, int dy_ld, const float* __restrict__ y, int y_ld, float cap,
                                     float* __restrict__ dx, int dx_ld, int P, int C) {
    const int qpr = C / 4;
    const size_t total = (size_t)P * qpr;
    GRID_STRIDE(i, total) {
        const int p = (int)(i / qpr);
        const int c = (int)(i - (size_t)p * qpr) * 4;
        const float4 g = *reinterpret_cast<const float4*>(dy + (size_t)p * dy_ld + c);
        const float4 v = *reinterpret_cast<const float4*>(y + (size_t)p * y_ld + c);
        *reinterpret_cast<float4*>(dx + (size_t)p * dx_ld + c) =
            make_float4(v.x < cap ? g.x : 0.f, v.y < cap ? g.y : 0.f, v.z < cap ? g.z : 0.f, v.w < cap ? g.w : 0.f);
    }
}

extern "C" int semseg_clamp_max(const float* x, int x_ld, float cap, float* y, int y_ld, int P, int C, void* stream) {
    if (!x || !y || P <= 0 || C <= 0 || (C % 4) || (x_ld % 4) || (y_ld % 4) || x_ld < C || y_ld < C) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(clamp_max_kernel, dim3(stream_blocks((size_t)P * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, x_ld, cap,
                       y, y_ld, P, C);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

extern "C" int semseg_clamp_max_bwd(const float* dy, int dy_ld, const float* y, int y_ld, float cap, float* dx, int dx_ld, int P,
                                    int C, void* stream) {
    if (!dy || !y || !dx || P <= 0 || C <= 0 || (C % 4) || (dy_ld % 4) || (y_ld % 4) || (dx_ld % 4)) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(clamp_max_bwd_kernel, dim3(stream_blocks((size_t)P * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, dy_ld,
                       y, y_ld, cap, dx, dx_ld, P, C);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// nn.Dropout2d's per-(n, c) Bernoulli keep mask (models.py:460,464): mask[i] = keep ? 1/(1-p) : 0 from a counter-based hash of
// (seed, launch counter, i).  `state` = {seed, counter} in DEVICE memory; the kernel advances the counter itself, so a captured
// launch draws a fresh mask on every hipGraph replay.  One block (the mask has N*C <= a few thousand entries).
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ mask, int n, float p, float keep_scale,
                                                           unsigned long long* __restrict__ state) {
    const uint64_t seed = state[0], counter = state[1];
    const uint64_t base = mix64(seed ^ mix64(counter));
    for (int i = threadIdx.x; i < n; i += 256) {
        const uint32_t r = (uint32_t)(mix64(base + (uint64_t)i) >> 40);           // 24 uniform bits
        const float u = (float)r * (1.0f / 16777216.0f);                          // [0, 1), exactly representable
        mask[i] = u >= p ? keep_scale : 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) state[1] = counter + 1;
}

extern "C" int semseg_dropout_mask(float* mask, int n, float p, void* state, void* stream) {
    if (!mask || !state || n <= 0 || !(p >= 0.f) || !(p < 1.f)) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mask, n, p, 1.0f / (1.0f - p),
                       (unsigned long long*)state);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

template <bool ACC, bool VEC>
struct copy2d_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ src, int src_ld, float* __restrict__ dst, int dst_ld, int P, int C) {
    if (VEC) {
        const int qpr = C / 4;
        const size_t total = (size_t)P * qpr;
        GRID_STRIDE(i, total) {
            const int p = (int)(i / qpr);
            const int c = (int)(i - (size_t)p * qpr) * 4;
            float4 v = *reinterpret_cast<const float4*>(src + (size_t)p * src_ld + c);
            float4* d = reinterpret_cast<float4*>(dst + (size_t)p * dst_ld + c);
            if (ACC) { const float4 o = *d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *d = v;
        }
    } else {
        const size_t total = (size_t)P * C;
        GRID_STRIDE(i, total) {
            const int p = (int)(i / C);
            const int c = (int)(i - (size_t)p * C);
            float v = src[(size_t)p * src_ld + c];
            float* d = dst + (size_t)p * dst_ld + c;
            if (ACC) v += *d;
            *d = v;
        }
    }
    }
};

extern "C" int semseg_copy2d(const float* src, int src_ld, float* dst, int dst_ld, int P, int C, int accumulate,
                             void* stream) {
    if (!src || !dst || P <= 0 || C <= 0 || src_ld < C || dst_ld < C) return SEMSEG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (C % 4 == 0) && (src_ld % 4 == 0) && (dst_ld % 4 == 0) && aligned16(src) && aligned16(dst);
    const int blocks = stream_blocks(vec ? (size_t)P * (C / 4) : (size_t)P * C);
#define LAUNCH(A, V) SEMSEG_LAUNCH_BODY((copy2d_kernel_body<A, V>), dim3(blocks), 0, st, src, src_ld, dst, dst_ld, P, C)
    if (accumulate) { if (vec) LAUNCH(true, true); else LAUNCH(true, false); }
    else            { if (vec) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

__global__ void scale_nc_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ y, int N,
                                int HW, int C) {
    const int qpr = C / 4;
    const size_t total = (size_t)N * HW * qpr;
    GRID_STRIDE(i, total) {
        const size_t p = i / qpr;
        const int c = (int)(i - p * qpr) * 4;
        const int n = (int)(p / HW);
        const float4 v = *reinterpret_cast<const float4*>(x + p * C + c);
        const float4 m = *reinterpret_cast<const float4*>(mask + (size_t)n * C + c);
        *reinterpret_cast<float4*>(y + p * C + c) = make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w);
    }
}

extern "C" int semseg_scale_nc(const float* x, const float* mask, float* y, int N, int HW, int C, void* stream) {
    if (!x || !mask || !y || N <= 0 || HW <= 0 || C <= 0 || (C % 4)) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(scale_nc_kernel, dim3(stream_blocks((size_t)N * HW * (C / 4))), dim3(256), 0, (hipStream_t)stream, x,
                       mask, y, N, HW, C);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// NCHW <-> NHWC: per image a [C][HW] <-> [HW][C] transpose through a 32x33 LDS tile
__global__ void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int cols) {
    // x: [batch][rows][cols] -> y: [batch][cols][rows]
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? x[base + (size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) y[base + (size_t)c * rows + r] = tile[tx][i];
    }
}

extern "C" int semseg_nchw_to_nhwc(const float* x, float* y, int N, int C, int HW, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || HW <= 0) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(HW, 32), ceil_div(C, 32), N), dim3(256), 0, (hipStream_t)stream, x, y, C, HW);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}
extern "C" int semseg_nhwc_to_nchw(const float* x, float* y, int N, int C, int HW, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || HW <= 0) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(C, 32), ceil_div(HW, 32), N), dim3(256), 0, (hipStream_t)stream, x, y, HW, C);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ max pool 3x3 s2 p1 ----------
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx, int N, int H,
                                   int W, int C, int OH, int OW) {
    const int qpr = C / 4;
    const size_t total = (size_t)N * OH * OW * qpr;
    GRID_STRIDE(i, total) {
        const size_t op = i / qpr;
        const int c = (int)(i - op * qpr) * 4;
        const int ow = (int)(op % OW);
        const int oh = (int)((op / OW) % OH);
        const int n = (int)(op / ((size_t)OW * OH));
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
        bool first = true;
        for (int r = 0; r < 3; ++r) {
            const int ih = oh * 2 - 1 + r;
            if (ih < 0 || ih >= H) continue;
            for (int s = 0; s < 3; ++s) {
                const int iw = ow * 2 - 1 + s;
                if (iw < 0 || iw >= W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)(n * H + ih) * W + iw) * C + c);
                const int t = r * 3 + s;
                // torch CPU max_pool2d: take val if (val > max) || isnan(val); first visited wins ties
                if (first || v.x > best.x || v.x != v.x) { best.x = v.x; b0 = t; }
                if (first || v.y > best.y || v.y != v.y) { best.y = v.y; b1 = t; }
                if (first || v.z > best.z || v.z != v.z) { best.z = v.z; b2 = t; }
                if (first || v.w > best.w || v.w != v.w) { best.w = v.w; b3 = t; }
                first = false;
            }
        }
        *reinterpret_cast<float4*>(y + op * C + c) = best;
        *reinterpret_cast<uchar4*>(idx + op * C + c) = make_uchar4((unsigned char)b0, (unsigned char)b1, (unsigned char)b2, (unsigned char)b3);
    }
}

__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx, int N,
                                   int H, int W, int C, int OH, int OW) {
    const int qpr = C / 4;
    const size_t total = (size_t)N * H * W * qpr;
    GRID_STRIDE(i, total) {
        const size_t ip = i / qpr;
        const int c = (int)(i - ip * qpr) * 4;
        const int iw = (int)(ip % W);
        const int ih = (int)((ip / W) % H);
        const int n = (int)(ip / ((size_t)W * H));
        float4 acc = f4zero();
        // windows (oh, r) with oh*2-1+r == ih
        for (int r = 0; r < 3; ++r) {
            const int t = ih + 1 - r;
            if (t < 0 || (t & 1)) continue;
            const int oh = t >> 1;
            if (oh >= OH) continue;
            for (int s = 0; s < 3; ++s) {
                const int u = iw + 1 - s;
                if (u < 0 || (u & 1)) continue;
                const int ow = u >> 1;
                if (ow >= OW) continue;
                const size_t op = ((size_t)(n * OH + oh) * OW + ow);
                const uchar4 k = *reinterpret_cast<const uchar4*>(idx + op * C + c);
                const float4 g = *reinterpret_cast<const float4*>(dy + op * C + c);
                const int tt = r * 3 + s;
                if (k.x == tt) acc.x += g.x;
                if (k.y == tt) acc.y += g.y;
                if (k.z == tt) acc.z += g.z;
                if (k.w == tt) acc.w += g.w;
            }
        }
        *reinterpret_cast<float4*>(dx + ip * C + c) = acc;
    }
}

extern "C" int semseg_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream) {
    if (!x || !y || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4)) return SEMSEG_EINVAL;
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(stream_blocks((size_t)N * OH * OW * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       x, y, idx, N, H, W, C, OH, OW);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}
extern "C" int semseg_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int N, int H, int W, int C, void* stream) {
    if (!dy || !dx || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4)) return SEMSEG_EINVAL;
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(stream_blocks((size_t)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy,
                       idx, dx, N, H, W, C, OH, OW);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ adaptive average pool -------
__device__ __forceinline__ int bin_start(int i, int in, int out) { return (i * in) / out; }
__device__ __forceinline__ int bin_end(int i, int in, int out) { return ((i + 1) * in + out - 1) / out; }

// block = 16 channel quads (64 channels) x 16 pixel lanes; one output bin per block.x, channel chunk per block.y
__global__ __launch_bounds__(256) void adaptive_avgpool_fwd_kernel(const float* __restrict__ x, int x_ld, float* __restrict__ y,
                                                                   int N, int H, int W, int C, int OH, int OW) {
    __shared__ float4 red[16][16];
    const int tq = threadIdx.x & 15, tp = threadIdx.x >> 4;
    const int c = (blockIdx.y * 16 + tq) * 4;
    const int bin = blockIdx.x;
    const int ow = bin % OW, oh = (bin / OW) % OH, n = bin / (OW * OH);
    const int h0 = bin_start(oh, H, OH), h1 = bin_end(oh, H, OH);
    const int w0 = bin_start(ow, W, OW), w1 = bin_end(ow, W, OW);
    const int bw = w1 - w0, cnt = (h1 - h0) * bw;
    float4 s = f4zero();
    if (c < C) {
        for (int k = tp; k < cnt; k += 16) {
            const int ih = h0 + k / bw, iw = w0 + k % bw;
            const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)(n * H + ih) * W + iw) * x_ld + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[tp][tq] = s;
    __syncthreads();
    if (tp == 0 && c < C) {
        float4 a = f4zero();
        for (int k = 0; k < 16; ++k) { const float4 v = red[k][tq]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        const float inv = 1.0f / (float)cnt;
        *reinterpret_cast<float4*>(y + (size_t)bin * C + c) = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
    }
}

template <bool ACC>
__global__ void adaptive_avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int dx_ld, int N, int H,
                                            int W, int C, int OH, int OW) {
    const int qpr = C / 4;
    const size_t total = (size_t)N * H * W * qpr;
    GRID_STRIDE(i, total) {
        const size_t ip = i / qpr;
        const int c = (int)(i - ip * qpr) * 4;
        const int iw = (int)(ip % W);
        const int ih = (int)((ip / W) % H);
        const int n = (int)(ip / ((size_t)W * H));
        float4 acc = f4zero();
        // bins containing ih: oh in [floor(ih*OH/H) - 1, ...] -- scan the (at most 2-3) candidates
        int oh_lo = (ih * OH) / H; if (oh_lo > 0) --oh_lo;
        int ow_lo = (iw * OW) / W; if (ow_lo > 0) --ow_lo;
        for (int oh = oh_lo; oh < OH && bin_start(oh, H, OH) <= ih; ++oh) {
            const int h0 = bin_start(oh, H, OH), h1 = bin_end(oh, H, OH);
            if (ih < h0 || ih >= h1) continue;
            for (int ow = ow_lo; ow < OW && bin_start(ow, W, OW) <= iw; ++ow) {
                const int w0 = bin_start(ow, W, OW), w1 = bin_end(ow, W, OW);
                if (iw < w0 || iw >= w1) continue;
                const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
                const float4 g = *reinterpret_cast<const float4*>(dy + ((size_t)(n * OH + oh) * OW + ow) * C + c);
                acc.x += g.x * inv; acc.y += g.y * inv; acc.z += g.z * inv; acc.w += g.w * inv;
            }
        }
        float4* d = reinterpret_cast<float4*>(dx + ip * dx_ld + c);
        if (ACC) { const float4 o = *d; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        *d = acc;
    }
}

extern "C" int semseg_adaptive_avgpool_fwd(const float* x, int x_ld, float* y, int N, int H, int W, int C, int OH, int OW,
                                           void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) || (x_ld % 4) || OH <= 0 || OW <= 0) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(adaptive_avgpool_fwd_kernel, dim3(N * OH * OW, ceil_div(C, 64)), dim3(256), 0, (hipStream_t)stream, x, x_ld,
                       y, N, H, W, C, OH, OW);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}
extern "C" int semseg_adaptive_avgpool_bwd(const float* dy, float* dx, int dx_ld, int accumulate, int N, int H, int W, int C,
                                           int OH, int OW, void* stream) {
    if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) || (dx_ld % 4) || OH <= 0 || OW <= 0) return SEMSEG_EINVAL;
    const int blocks = stream_blocks((size_t)N * H * W * (C / 4));
    if (accumulate) hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, dx, dx_ld, N, H, W, C, OH, OW);
    else            hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, dx, dx_ld, N, H, W, C, OH, OW);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ pyramid pooling: all scales in one pass
// PPM / UPerNet pool the SAME map to several square grids (models.py:447-450,511-514: scales 1, 2, 3, 6).  Per-scale
// kernels read the map once per scale and the four backward results are summed by three more passes; here
//   forward : pass 1 -- one thread per (n, row, channel quad) walks the row once and keeps the column-bin sums of every scale
//             in registers (<= MP_MAXCOLS bins in total: 12 for 1+2+3+6) -> rowsum[n][h][col][c];
//             pass 2 -- per output bin, sum its rows of rowsum in row order and scale by 1/count.  Fixed summation order.
//   backward: one pass writes dx = sum over scales and bins containing the pixel of g/count.
constexpr int MP_MAXSCALES = 4;
constexpr int MP_MAXCOLS = 16;
struct MultiPool {
    float* y[MP_MAXSCALES];          // forward outputs / backward incoming gradients, [N][s][s][C] dense
    int s[MP_MAXSCALES];
    int col0[MP_MAXSCALES];          // first column-bin slot of the scale
    int nscale, ncols;
};

__global__ __launch_bounds__(256) void multipool_finish_kernel(const float* __restrict__ rowsum, MultiPool mp, int N, int H, int W,
                                                               int C, int bins_total) {
    const int qpr = C / 4;
    const size_t total = (size_t)N * bins_total * qpr;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % qpr) * 4;
        size_t b = i / qpr;
        const int n = (int)(b / bins_total);
        int bin = (int)(b - (size_t)n * bins_total);
        int sc = 0;
        while (sc + 1 < mp.nscale && bin >= mp.s[sc] * mp.s[sc]) { bin -= mp.s[sc] * mp.s[sc]; ++sc; }
        const int S = mp.s[sc], oh = bin / S, ow = bin % S;
        const int h0 = bin_start(oh, H, S), h1 = bin_end(oh, H, S);
        const int cnt = (h1 - h0) * (bin_end(ow, W, S) - bin_start(ow, W, S));
        float4 a = f4zero();
        // 8 rows in flight per round (rows past the bin: clamped loads, values unused); additions in row order
        for (int hb = h0; hb < h1; hb += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int h = min(hb + u, h1 - 1);
                v[u] = *reinterpret_cast<const float4*>(rowsum + (((size_t)n * H + h) * mp.ncols + mp.col0[sc] + ow) * C + c);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool live = hb + u < h1;
                a.x = live ? a.x + v[u].x : a.x; a.y = live ? a.y + v[u].y : a.y;
                a.z = live ? a.z + v[u].z : a.z; a.w = live ? a.w + v[u].w : a.w;
            }
        }
        const float inv = 1.0f / (float)cnt;
        *reinterpret_cast<float4*>(mp.y[sc] + (((size_t)n * S + oh) * S + ow) * C + c) = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
    }
}

__global__ __launch_bounds__(256) void multipool_bwd_kernel(MultiPool mp, float* __restrict__ dx, int dx_ld, int N, int H, int W,
                                                            int C) {
    const int qpr = C / 4;
    const size_t total = (size_t)N * H * W * qpr;
    GRID_STRIDE(i, total) {
        const size_t ip = i / qpr;
        const int c = (int)(i - ip * qpr) * 4;
        const int iw = (int)(ip % W);
        const int ih = (int)((ip / W) % H);
        const int n = (int)(ip / ((size_t)W * H));
        float4 acc = f4zero();
        for (int sc = 0; sc < mp.nscale; ++sc) {
            const int S = mp.s[sc];
            const float* dy = mp.y[sc];
            int oh_lo = (ih * S) / H; if (oh_lo > 0) --oh_lo;
            int ow_lo = (iw * S) / W; if (ow_lo > 0) --ow_lo;
            for (int oh = oh_lo; oh < S && bin_start(oh, H, S) <= ih; ++oh) {
                const int h0 = bin_start(oh, H, S), h1 = bin_end(oh, H, S);
                if (ih < h0 || ih >= h1) continue;
                for (int ow = ow_lo; ow < S && bin_start(ow, W, S) <= iw; ++ow) {
                    const int w0 = bin_start(ow, W, S), w1 = bin_end(ow, W, S);
                    if (iw < w0 || iw >= w1) continue;
                    const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
                    const float4 g = *reinterpret_cast<const float4*>(dy + ((size_t)(n * S + oh) * S + ow) * C + c);
                    acc.x += g.x * inv; acc.y += g.y * inv; acc.z += g.z * inv; acc.w += g.w * inv;
                }
            }
        }
        *reinterpret_cast<float4*>(dx + ip * dx_ld + c) = acc;
    }
}

// ---- the two streaming passes.  A first generation evaluated the bin bounds -- integer divisions by run-time values -- per
// (pixel, slot) in the forward row walk and per (pixel, channel quad) in the backward pass and was ALU-bound (rocprof r2s:
// 123 us / 111 us for a 2x64x64x2048 map whose 67 MB stream in ~15 us; multipool_bwd_kernel above is what is left of it, for
// maps with fewer than two pixels per bin).  Here the bounds are computed once per thread (forward: 16 slots before the row walk; backward: per pixel, reused for every channel quad
// of the thread) and the forward walk keeps 8 row loads in flight.  Same summation order, bit-identical results.
__global__ __launch_bounds__(256) void multipool_rowsum2_kernel(const float* __restrict__ x, int x_ld, float* __restrict__ rowsum,
                                                                MultiPool mp, int N, int H, int W, int C) {
    int lo[MP_MAXCOLS], hi[MP_MAXCOLS];
#pragma unroll
    for (int k = 0; k < MP_MAXCOLS; ++k) {
        int sc = 0;
#pragma unroll
        for (int t = 1; t < MP_MAXSCALES; ++t) sc += (t < mp.nscale && k >= mp.col0[t]) ? 1 : 0;
        const int j = k - mp.col0[sc];
        const bool live = k < mp.ncols;
        lo[k] = live ? bin_start(j, W, mp.s[sc]) : 0;
        hi[k] = live ? bin_end(j, W, mp.s[sc]) : 0;
    }
    const int qpr = C / 4;
    const size_t total = (size_t)N * H * qpr;
    GRID_STRIDE(i, total) {
        const size_t row = i / qpr;                 // n * H + h
        const int c = (int)(i - row * qpr) * 4;
        float4 acc[MP_MAXCOLS];
#pragma unroll
        for (int k = 0; k < MP_MAXCOLS; ++k) acc[k] = f4zero();
        const float* src = x + row * W * (size_t)x_ld + c;
        for (int w0 = 0; w0 < W; w0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = (w0 + u < W) ? *reinterpret_cast<const float4*>(src + (size_t)(w0 + u) * x_ld) : f4zero();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int w = w0 + u;
                if (w < W) {
#pragma unroll
                    for (int k = 0; k < MP_MAXCOLS; ++k) {
                        const bool in = (w >= lo[k]) && (w < hi[k]);
                        acc[k].x += in ? v[u].x : 0.f; acc[k].y += in ? v[u].y : 0.f;
                        acc[k].z += in ? v[u].z : 0.f; acc[k].w += in ? v[u].w : 0.f;
                    }
                }
            }
        }
        float* dst = rowsum + row * (size_t)mp.ncols * C + c;
#pragma unroll
        for (int k = 0; k < MP_MAXCOLS; ++k)
            if (k < mp.ncols) *reinterpret_cast<float4*>(dst + (size_t)k * C) = acc[k];
    }
}

// block = 4 pixels x 64 channel-quad lanes.  Requires H >= 2 S and W >= 2 S for every scale (checked by the launcher):
// then a pixel lies in at most two bins per axis.
__global__ __launch_bounds__(256) void multipool_bwd2_kernel(MultiPool mp, float* __restrict__ dx, int dx_ld, int N, int H, int W,
                                                             int C) {
    const int lane = threadIdx.x & 63;
    const unsigned P = (unsigned)N * H * W;
    const int qpr = C / 4;
    for (unsigned ip = blockIdx.x * 4 + (threadIdx.x >> 6); ip < P; ip += gridDim.x * 4) {
        const int iw = (int)(ip % (unsigned)W);
        const unsigned t = ip / (unsigned)W;
        const int ih = (int)(t % (unsigned)H);
        const int n = (int)(t / (unsigned)H);
        // per scale: first bin and number of bins (1 or 2) containing the pixel on each axis, 1/count of the 2x2 candidates
        int base[MP_MAXSCALES], nh[MP_MAXSCALES], nw[MP_MAXSCALES];
        float inv[MP_MAXSCALES][2][2];
#pragma unroll
        for (int sc = 0; sc < MP_MAXSCALES; ++sc) {
            base[sc] = 0; nh[sc] = 0; nw[sc] = 0;
            if (sc < mp.nscale) {
                const int S = mp.s[sc];
                int oh = (ih * S) / H; if (oh > 0) --oh;
                if (!(ih >= bin_start(oh, H, S) && ih < bin_end(oh, H, S))) ++oh;        // the first bin containing ih
                int ow = (iw * S) / W; if (ow > 0) --ow;
                if (!(iw >= bin_start(ow, W, S) && iw < bin_end(ow, W, S))) ++ow;
                const int hl0 = bin_end(oh, H, S) - bin_start(oh, H, S), wl0 = bin_end(ow, W, S) - bin_start(ow, W, S);
                int hl1 = 0, wl1 = 0;
                nh[sc] = 1; nw[sc] = 1;
                if (oh + 1 < S && bin_start(oh + 1, H, S) <= ih) { nh[sc] = 2; hl1 = bin_end(oh + 1, H, S) - bin_start(oh + 1, H, S); }
                if (ow + 1 < S && bin_start(ow + 1, W, S) <= iw) { nw[sc] = 2; wl1 = bin_end(ow + 1, W, S) - bin_start(ow + 1, W, S); }
                base[sc] = (n * S + oh) * S + ow;
                inv[sc][0][0] = 1.0f / (float)(hl0 * wl0);
                inv[sc][0][1] = 1.0f / (float)(hl0 * (wl1 > 0 ? wl1 : 1));
                inv[sc][1][0] = 1.0f / (float)((hl1 > 0 ? hl1 : 1) * wl0);
                inv[sc][1][1] = 1.0f / (float)((hl1 > 0 ? hl1 : 1) * (wl1 > 0 ? wl1 : 1));
            }
        }
        for (int q = lane; q < qpr; q += 64) {
            const int c = q * 4;
            float4 acc = f4zero();
#pragma unroll
            for (int sc = 0; sc < MP_MAXSCALES; ++sc) {
                if (sc < mp.nscale) {
                    const int S = mp.s[sc];
                    const float* dy = mp.y[sc];
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            if (a < nh[sc] && b < nw[sc]) {
                                const float4 g = *reinterpret_cast<const float4*>(dy + (size_t)(base[sc] + a * S + b) * C + c);
                                const float f = inv[sc][a][b];
                                acc.x += g.x * f; acc.y += g.y * f; acc.z += g.z * f; acc.w += g.w * f;
                            }
                }
            }
            *reinterpret_cast<float4*>(dx + (size_t)ip * dx_ld + c) = acc;
        }
    }
}

static int multipool_setup(MultiPool& mp, void* const* ys_host, const int* sizes_host, int nscale) {
    if (!ys_host || !sizes_host || nscale <= 0 || nscale > MP_MAXSCALES) return SEMSEG_EINVAL;
    mp.nscale = nscale;
    mp.ncols = 0;
    for (int i = 0; i < MP_MAXSCALES; ++i) { mp.y[i] = nullptr; mp.s[i] = 1; mp.col0[i] = 0; }
    for (int i = 0; i < nscale; ++i) {
        if (!ys_host[i] || sizes_host[i] <= 0) return SEMSEG_EINVAL;
        mp.y[i] = (float*)ys_host[i];
        mp.s[i] = sizes_host[i];
        mp.col0[i] = mp.ncols;
        mp.ncols += sizes_host[i];
    }
    return mp.ncols <= MP_MAXCOLS ? 0 : SEMSEG_EINVAL;
}

extern "C" size_t semseg_adaptive_avgpool_multi_workspace_bytes(int N, int H, int C, const int* sizes_host, int nscale) {
    size_t cols = 0;
    for (int i = 0; i < nscale; ++i) cols += (size_t)sizes_host[i];
    return (size_t)N * H * cols * C * sizeof(float);
}

extern "C" int semseg_adaptive_avgpool_multi_fwd(const float* x, int x_ld, int N, int H, int W, int C, int nscale,
                                                 const int* sizes_host, void* const* ys_host, void* workspace,
                                                 size_t workspace_bytes, void* stream) {
    if (!x || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) || (x_ld % 4) || x_ld < C) return SEMSEG_EINVAL;
    MultiPool mp;
    const int rc = multipool_setup(mp, ys_host, sizes_host, nscale);
    if (rc) return rc;
    const size_t need = (size_t)N * H * mp.ncols * C * sizeof(float);
    if (!workspace || workspace_bytes < need) return SEMSEG_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* rowsum = (float*)workspace;
    hipLaunchKernelGGL(multipool_rowsum2_kernel, dim3(stream_blocks((size_t)N * H * (C / 4))), dim3(256), 0, st, x, x_ld, rowsum,
                       mp, N, H, W, C);
    SEMSEG_LAUNCH_CHECK();
    int bins = 0;
    for (int i = 0; i < nscale; ++i) bins += sizes_host[i] * sizes_host[i];
    hipLaunchKernelGGL(multipool_finish_kernel, dim3(stream_blocks((size_t)N * bins * (C / 4))), dim3(256), 0, st,
                       (const float*)rowsum, mp, N, H, W, C, bins);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

extern "C" int semseg_adaptive_avgpool_multi_bwd(void* const* dys_host, const int* sizes_host, int nscale, float* dx, int dx_ld,
                                                 int N, int H, int W, int C, void* stream) {
    if (!dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) || (dx_ld % 4) || dx_ld < C) return SEMSEG_EINVAL;
    MultiPool mp;
    const int rc = multipool_setup(mp, dys_host, sizes_host, nscale);
    if (rc) return rc;
    bool two_bins = (size_t)N * H * W < ((size_t)1 << 31);
    for (int i = 0; i < nscale; ++i) two_bins = two_bins && H >= 2 * sizes_host[i] && W >= 2 * sizes_host[i];
    if (two_bins) {
        size_t blocks = ceil_div_sz((size_t)N * H * W, 4);
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(multipool_bwd2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mp, dx, dx_ld, N, H, W, C);
    } else {
        hipLaunchKernelGGL(multipool_bwd_kernel, dim3(stream_blocks((size_t)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                           mp, dx, dx_ld, N, H, W, C);
    }
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ bilinear, align_corners=False
// torch area_pixel_compute_source_index: src = max(0, scale*(dst+0.5)-0.5), scale = in/out (float);
// i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1.
__device__ __forceinline__ void bl_coord(int d, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    float src = scale * ((float)d + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

template <int V> struct VecT;
template <> struct VecT<4> { typedef float4 type; };
template <> struct VecT<2> { typedef float2 type; };
template <> struct VecT<1> { typedef float type; };

// V = floats per lane (4 when C % 4 == 0; 2 / 1 for the 150-class logits of the inference branch)
template <bool ACC, bool RELU, int V>
struct bilinear_fwd_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ x, int x_ld, float* __restrict__ y, int y_ld, int N, int IH,
                                    int IW, int OH, int OW, int C, float sh, float sw) {
    typedef typename VecT<V>::type vec;
    const int qpr = C / V;
    const size_t total = (size_t)N * OH * OW * qpr;
    GRID_STRIDE(i, total) {
        const size_t op = i / qpr;
        const int c = (int)(i - op * qpr) * V;
        const int ow = (int)(op % OW);
        const int oh = (int)((op / OW) % OH);
        const int n = (int)(op / ((size_t)OW * OH));
        int h0, h1, w0, w1;
        float lh0, lh1, lw0, lw1;
        bl_coord(oh, sh, IH, h0, h1, lh0, lh1);
        bl_coord(ow, sw, IW, w0, w1, lw0, lw1);
        const float* b = x + (size_t)n * IH * IW * x_ld + c;
        const vec v00 = *reinterpret_cast<const vec*>(b + ((size_t)h0 * IW + w0) * x_ld);
        const vec v01 = *reinterpret_cast<const vec*>(b + ((size_t)h0 * IW + w1) * x_ld);
        const vec v10 = *reinterpret_cast<const vec*>(b + ((size_t)h1 * IW + w0) * x_ld);
        const vec v11 = *reinterpret_cast<const vec*>(b + ((size_t)h1 * IW + w1) * x_ld);
        vec* d = reinterpret_cast<vec*>(y + op * y_ld + c);
        vec prev;
        if (ACC) prev = *d;
        vec o;
        // same association as torch upsample_bilinear2d: h0lambda*(w0lambda*v00 + w1lambda*v01) + h1lambda*(...)
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float a00 = reinterpret_cast<const float*>(&v00)[e], a01 = reinterpret_cast<const float*>(&v01)[e];
            const float a10 = reinterpret_cast<const float*>(&v10)[e], a11 = reinterpret_cast<const float*>(&v11)[e];
            float r = lh0 * (lw0 * a00 + lw1 * a01) + lh1 * (lw0 * a10 + lw1 * a11);
            if (ACC) r += reinterpret_cast<const float*>(&prev)[e];
            if (RELU) r = fmaxf(r, 0.f);
            reinterpret_cast<float*>(&o)[e] = r;
        }
        *d = o;
    }
    }
};

extern "C" int semseg_bilinear_fwd(const float* x, int x_ld, float* y, int y_ld, int accumulate, int relu, int N, int IH, int IW,
                                   int OH, int OW, int C, void* stream) {
    if (!x || !y || N <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || C <= 0 || x_ld < C || y_ld < C) return SEMSEG_EINVAL;
    const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
    int V = 1;
    if (C % 4 == 0 && x_ld % 4 == 0 && y_ld % 4 == 0 && aligned16(x) && aligned16(y)) V = 4;
    else if (C % 2 == 0 && x_ld % 2 == 0 && y_ld % 2 == 0 && (((uintptr_t)x | (uintptr_t)y) & 7) == 0) V = 2;
    const int blocks = stream_blocks((size_t)N * OH * OW * (C / V));
#define LAUNCH(A, R, VV) SEMSEG_LAUNCH_BODY((bilinear_fwd_kernel_body<A, R, VV>), dim3(blocks), 0, (hipStream_t)stream, x, x_ld, y, y_ld, N, IH, IW, OH, OW, C, sh, sw)
#define LAUNCH_V(A, R) do { if (V == 4) LAUNCH(A, R, 4); else if (V == 2) LAUNCH(A, R, 2); else LAUNCH(A, R, 1); } while (0)
    if (accumulate) { if (relu) LAUNCH_V(true, true); else LAUNCH_V(true, false); }
    else            { if (relu) LAUNCH_V(false, true); else LAUNCH_V(false, false); }
#undef LAUNCH_V
#undef LAUNCH
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// Inference head in ONE pass (models.py:480-484 / eval.py:66-71): out[n, oh, ow, :] (+)= weight * softmax_c(bilinear(logits)(oh, ow, :))
// -- the bilinear up-sampling of the class logits to segSize, the softmax over the classes and the multi-scale average
// `scores = scores + scores_tmp / len(imgSizes)`.  One wave per output pixel: lane l holds classes l, l+64, ... (K per lane),
// the four source pixels are read as contiguous class vectors, max / sum by wave shuffles, the probabilities are written as one
// contiguous class vector.  Versus bilinear_fwd + softmax_fwd + a scale and an add pass: the full-resolution class map is
// written once (read once more when accumulating) instead of written three times and read three times.
template <int K, bool ACC>
__global__ __launch_bounds__(256) void upsample_softmax_kernel(const float* __restrict__ x, int x_ld, float* __restrict__ out,
                                                               int out_ld, int N, int IH, int IW, int OH, int OW, int C, float sh,
                                                               float sw, float weight) {
    const int lane = threadIdx.x & 63;
    const size_t P = (size_t)N * OH * OW;
    for (size_t op = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); op < P; op += (size_t)gridDim.x * 4) {
        const int ow = (int)(op % OW);
        const int oh = (int)((op / OW) % OH);
        const int n = (int)(op / ((size_t)OW * OH));
        int h0, h1, w0, w1;
        float lh0, lh1, lw0, lw1;
        bl_coord(oh, sh, IH, h0, h1, lh0, lh1);
        bl_coord(ow, sw, IW, w0, w1, lw0, lw1);
        const float* b = x + (size_t)n * IH * IW * x_ld;
        const float* r00 = b + ((size_t)h0 * IW + w0) * x_ld;
        const float* r01 = b + ((size_t)h0 * IW + w1) * x_ld;
        const float* r10 = b + ((size_t)h1 * IW + w0) * x_ld;
        const float* r11 = b + ((size_t)h1 * IW + w1) * x_ld;
        float v[K];
        float m = -INFINITY;
        // All 4 K requests of the pixel leave together: unconditional loads at a clamped class index, then ONE point where they
        // are consumed (the empty asm statements: without them the compiler sinks each k's loads back into its `c < C` branch and
        // every k waits for a round trip of its own -- tools/isa_wait_lint.py).
        float a00[K], a01[K], a10[K], a11[K], prev[K];
        float* o = out + op * out_ld;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + 64 * k;
            const int cc = c < C ? c : C - 1;
            a00[k] = r00[cc]; a01[k] = r01[cc]; a10[k] = r10[cc]; a11[k] = r11[cc];
            prev[k] = ACC ? o[cc] : 0.f;               // the scores accumulated so far (eval.py:70), requested with the rest
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < K; ++k) {
            asm volatile("" : "+v"(a00[k]), "+v"(a01[k]), "+v"(a10[k]), "+v"(a11[k]));
            if (ACC) asm volatile("" : "+v"(prev[k]));
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + 64 * k;
            // the association of bilinear_fwd_kernel (= torch upsample_bilinear2d)
            v[k] = (c < C) ? lh0 * (lw0 * a00[k] + lw1 * a01[k]) + lh1 * (lw0 * a10[k] + lw1 * a11[k]) : -INFINITY;
            m = fmaxf(m, v[k]);
        }
        m = wave_max(m);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            v[k] = (lane + 64 * k < C) ? expf(v[k] - m) : 0.f;
            sum += v[k];
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + 64 * k;
            if (c < C) {
                const float pr = (v[k] * inv) * weight;        // softmax, then `/ len(imgSizes)` as eval.py:70 (weight = 1/n)
                o[c] = ACC ? prev[k] + pr : pr;
            }
        }
    }
}

extern "C" int semseg_upsample_softmax(const float* logits, int x_ld, float* out, int out_ld, int accumulate, float weight, int N,
                                       int IH, int IW, int OH, int OW, int C, void* stream) {
    if (!logits || !out || N <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || C <= 0 || C > 1024 || x_ld < C || out_ld < C)
        return SEMSEG_EINVAL;
    const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
    const size_t P = (size_t)N * OH * OW;
    const int blocks = (int)(ceil_div_sz(P, 4) < 16384 ? ceil_div_sz(P, 4) : 16384);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(KK)                                                                                                          \
    do {                                                                                                                    \
        if (accumulate) hipLaunchKernelGGL((upsample_softmax_kernel<KK, true>), dim3(blocks), dim3(256), 0, st, logits, x_ld, out, \
                                           out_ld, N, IH, IW, OH, OW, C, sh, sw, weight);                                   \
        else hipLaunchKernelGGL((upsample_softmax_kernel<KK, false>), dim3(blocks), dim3(256), 0, st, logits, x_ld, out, out_ld, N, \
                                IH, IW, OH, OW, C, sh, sw, weight);                                                          \
    } while (0)
    const int K = ceil_div(C, 64);
    if (K <= 1) LAUNCH(1);
    else if (K <= 2) LAUNCH(2);
    else if (K <= 3) LAUNCH(3);
    else if (K <= 4) LAUNCH(4);
    else if (K <= 8) LAUNCH(8);
    else LAUNCH(16);
#undef LAUNCH
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// gather-form backward: block.x = one input pixel, block.y = a chunk of 4*ql channels; the 256/ql pixel lanes sweep the
// window of output pixels that can touch this input pixel, fixed-order LDS combine.  ql = 16 (64 channels, 16 pixel
// lanes) for ordinary maps; ql = 4 (16 channels, 64 pixel lanes) when the input map is tiny and every input pixel
// gathers from thousands of output pixels (PPM: 1x1 .. 6x6 -> 64x64 took 85 us per scale with 16 lanes).
template <bool ACC>
struct bilinear_bwd_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ dy, int dy_ld, float* __restrict__ dx,
                                                           int dx_ld, int N, int IH, int IW, int OH, int OW, int C, float sh,
                                                           float sw, int ql) {
    __shared__ float4 red[256];
    const int pl = 256 / ql;
    const int tq = threadIdx.x % ql, tp = threadIdx.x / ql;
    const int c = (blockIdx.y * ql + tq) * 4;
    const int ip = blockIdx.x;
    const int iw = ip % IW, ih = (ip / IW) % IH, n = ip / (IW * IH);
    // candidate output rows: src in (ih-1, ih+1)  =>  d in ((ih-0.5)/s - 0.5, (ih+1.5)/s - 0.5); widen by 1, clamp
    int oh_lo = (int)floorf(((float)ih - 0.5f) / sh - 0.5f) - 1; if (oh_lo < 0) oh_lo = 0;
    int oh_hi = (int)ceilf(((float)ih + 1.5f) / sh - 0.5f) + 1; if (oh_hi > OH - 1) oh_hi = OH - 1;
    int ow_lo = (int)floorf(((float)iw - 0.5f) / sw - 0.5f) - 1; if (ow_lo < 0) ow_lo = 0;
    int ow_hi = (int)ceilf(((float)iw + 1.5f) / sw - 0.5f) + 1; if (ow_hi > OW - 1) ow_hi = OW - 1;
    if (ih == 0) oh_lo = 0;            // clamped-at-zero source rows all map to i0 = 0
    if (iw == 0) ow_lo = 0;
    if (ih == IH - 1) oh_hi = OH - 1;
    if (iw == IW - 1) ow_hi = OW - 1;
    const int nh = oh_hi - oh_lo + 1, nw = ow_hi - ow_lo + 1;
    float4 s = f4zero();
    if (c < C) {
        // 4 window positions per round: their loads are issued together (a tiny input map makes this a long serial walk:
        // 1x1 -> 64x64 is 64 rounds per lane), the accumulation keeps the order of the window walk
        const int nwin = nh * nw;
        for (int k0 = tp; k0 < nwin; k0 += 4 * pl) {
            float wgt[4];
            float4 g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * pl;
                wgt[u] = 0.f;
                g[u] = f4zero();
                if (k < nwin) {
                    const int oh = oh_lo + k / nw, ow = ow_lo + k % nw;
                    int h0, h1, w0, w1;
                    float lh0, lh1, lw0, lw1;
                    bl_coord(oh, sh, IH, h0, h1, lh0, lh1);
                    bl_coord(ow, sw, IW, w0, w1, lw0, lw1);
                    float wh = 0.f, ww = 0.f;
                    if (h0 == ih) wh += lh0;
                    if (h1 == ih) wh += lh1;
                    if (w0 == iw) ww += lw0;
                    if (w1 == iw) ww += lw1;
                    wgt[u] = wh * ww;
                    if (wgt[u] != 0.f) g[u] = *reinterpret_cast<const float4*>(dy + ((size_t)(n * OH + oh) * OW + ow) * dy_ld + c);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (wgt[u] != 0.f) { s.x += wgt[u] * g[u].x; s.y += wgt[u] * g[u].y; s.z += wgt[u] * g[u].z; s.w += wgt[u] * g[u].w; }
        }
    }
    red[tp * ql + tq] = s;
    __syncthreads();
    if (tp == 0 && c < C) {
        float4 a = f4zero();
        for (int k = 0; k < pl; ++k) { const float4 v = red[k * ql + tq]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        float4* d = reinterpret_cast<float4*>(dx + (size_t)ip * dx_ld + c);
        if (ACC) { const float4 o = *d; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
        *d = a;
    }
    }
};

extern "C" int semseg_bilinear_bwd(const float* dy, int dy_ld, float* dx, int dx_ld, int accumulate, int N, int IH, int IW,
                                   int OH, int OW, int C, void* stream) {
    if (!dy || !dx || N <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || C <= 0 || (C % 4) || (dy_ld % 4) || (dx_ld % 4))
        return SEMSEG_EINVAL;
    const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
    // tiny input maps: few blocks, huge gather windows -> more pixel lanes per block and more (narrower) blocks
    const int ql = ((long)N * IH * IW * ceil_div(C, 64) < 512 && (long)OH * OW >= 16L * IH * IW) ? 4 : 16;
    dim3 grid(N * IH * IW, ceil_div(C, 4 * ql));
    if (accumulate) SEMSEG_LAUNCH_BODY((bilinear_bwd_kernel_body<true>), grid, 0, (hipStream_t)stream, dy, dy_ld, dx, dx_ld, N, IH, IW, OH, OW, C, sh, sw, ql);
    else            SEMSEG_LAUNCH_BODY((bilinear_bwd_kernel_body<false>), grid, 0, (hipStream_t)stream, dy, dy_ld, dx, dx_ld, N, IH, IW, OH, OW, C, sh, sw, ql);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}
