// The first convolution of every backbone (resnet.py:100 / hrnet.py:278: 3x3, stride 2, 3 input channels -> 64) as dedicated
// streaming kernels.  With C = 3 the implicit-GEMM path pads the reduction to 32 channels (10x redundant MFMA work, a split
// pass over the image, 33 MB of padded traffic in the weight gradient: 35 us forward + 84 us weight gradient at 2 x 512 x 512,
// profiles/r3j_tuned_plans_h2.json) for a layer whose arithmetic is 27 multiply-adds per output -- it is bound by the 33 MB of
// its output / output gradient, i.e. ~10 us.  Here:
//   forward : one thread = 4 output channels of one output pixel, 27 x C fp32 FMAs from the fp32 image (no split planes of the
//             image at all), weights tap-major in LDS;
//   wgrad   : dw[k][tap][c] = sum_p dy[p][k] x[p @ tap][c], dy read from the h2 planes the BN backward kernel wrote (dz has no
//             fp32 copy); one block = a run of pixels, thread = one k x one of 4 pixel lanes, the C x 9 input values of a pixel
//             read by all k-threads from the same addresses (broadcast); fp32 partial sums over <= 64 pixels, fp64 across lanes
//             / blocks in a fixed order.
// Arithmetic is exact fp32 FMA (tighter than the h2 products).  No data gradient: the image needs none.
#include "common.h"
#include "split_layout.h"

namespace {

constexpr int STEM_MAX_C = 4;
constexpr int STEM_TAPS = 9;

struct StemGeom {
    int N, H, W, C, K, stride, pad, dil, OH, OW, M;
};

__device__ __forceinline__ void stem_pixel(const StemGeom& g, int p, int& n, int& oh, int& ow) {
    ow = p % g.OW;
    const int t = p / g.OW;
    oh = t % g.OH;
    n = t / g.OH;
}

// w: [K][9][C] fp32 (KRSC)
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, int x_ld, const float* __restrict__ w,
                                                       float* __restrict__ y, int y_ld, StemGeom g) {
    extern __shared__ __align__(16) float ws[];   // [9*C][K]: tap-major, k contiguous
    const int KT = STEM_TAPS * g.C;
    for (int i = threadIdx.x; i < g.K * KT; i += blockDim.x) {
        const int k = i / KT, tc = i - k * KT;
        ws[tc * g.K + k] = w[i];
    }
    __syncthreads();
    const int KQ = g.K >> 2;
    const size_t total = (size_t)g.M * KQ;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(idx / KQ);
        const int k = (int)(idx - (size_t)p * KQ) << 2;
        int n, oh, ow;
        stem_pixel(g, p, n, oh, ow);
        float4 acc = f4zero();
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ih = oh * g.stride - g.pad + r * g.dil;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int iw = ow * g.stride - g.pad + s * g.dil;
                if (ih < 0 || ih >= g.H || iw < 0 || iw >= g.W) continue;
                const float* xp = x + ((size_t)(n * g.H + ih) * g.W + iw) * x_ld;
                for (int c = 0; c < g.C; ++c) {
                    const float xv = xp[c];
                    const float4 wv = *reinterpret_cast<const float4*>(ws + ((r * 3 + s) * g.C + c) * g.K + k);
                    acc.x = fmaf(xv, wv.x, acc.x); acc.y = fmaf(xv, wv.y, acc.y);
                    acc.z = fmaf(xv, wv.z, acc.z); acc.w = fmaf(xv, wv.w, acc.w);
                }
            }
        }
        *reinterpret_cast<float4*>(y + (size_t)p * y_ld + k) = acc;
    }
}

// partial[block][k][9*C] (double): sums over the block's pixels, in the scaled domain of the dy planes.
// thread = (k, one of 4 pixel lanes); a lane walks its own pixels (p0 + lane, += 4): the 9 x C input values of a pixel are read by
// the 64 k-threads of the wave from the SAME addresses (one broadcast transaction per load, L1-resident image rows), dy[p][k] is
// coalesced over k.  No barrier and no LDS inside the loop (a first version staged the inputs through LDS with two barriers per
// four pixels and was 2x slower than the padded implicit GEMM it replaces).
constexpr int WG_LANES = 4;                       // pixel lanes per block (threads = K x lanes, K <= 64)
template <int C>
__global__ __launch_bounds__(256) void stem_wgrad_partial_kernel(const float* __restrict__ x, int x_ld,
                                                                 const uint16_t* __restrict__ dyp, size_t dy_plane, int dy_pitch,
                                                                 double* __restrict__ partial, StemGeom g, int px_per_block) {
    __shared__ double red[WG_LANES][64];
    constexpr int KT = STEM_TAPS * C;
    const int k = threadIdx.x % g.K, lane = threadIdx.x / g.K;       // blockDim.x == K * WG_LANES
    const int p0 = blockIdx.x * px_per_block;
    const int p1 = min(g.M, p0 + px_per_block);
    float acc[KT];
#pragma unroll
    for (int i = 0; i < KT; ++i) acc[i] = 0.f;
    const _Float16* d0 = reinterpret_cast<const _Float16*>(dyp);
    const _Float16* d1 = reinterpret_cast<const _Float16*>(dyp + dy_plane);
    for (int p = p0 + lane; p < p1; p += WG_LANES) {
        int n, oh, ow;
        stem_pixel(g, p, n, oh, ow);
        const size_t o = (size_t)p * dy_pitch + k;
        const float d = (float)d0[o] + (float)d1[o];
        const float* img = x + (size_t)n * g.H * g.W * x_ld;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ih = oh * g.stride - g.pad + r * g.dil;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int iw = ow * g.stride - g.pad + s * g.dil;
                const bool ok = ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
                const float* xp = img + ((size_t)(ok ? ih : 0) * g.W + (ok ? iw : 0)) * x_ld;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float xv = ok ? xp[c] : 0.f;
                    acc[(r * 3 + s) * C + c] = fmaf(d, xv, acc[(r * 3 + s) * C + c]);
                }
            }
        }
    }
    // lanes -> one row per (k, tap-channel) in fp64, fixed order
#pragma unroll
    for (int i = 0; i < KT; ++i) {
        __syncthreads();
        red[lane][k] = (double)acc[i];
        __syncthreads();
        if (lane == 0) {
            double s = red[0][k];
#pragma unroll
            for (int l = 1; l < WG_LANES; ++l) s += red[l][k];
            partial[((size_t)blockIdx.x * g.K + k) * KT + i] = s;
        }
    }
}

// dw[k][tap][c] = 2^-e * sum_b partial[b][k][tap*C + c]
__global__ __launch_bounds__(256) void stem_wgrad_finish_kernel(const double* __restrict__ partial, int nblocks, int total,
                                                                const int* __restrict__ dy_exp, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * total + i];
    dw[i] = (float)ldexp(s, -dy_exp[0]);
}

bool stem_geom(StemGeom& g, int N, int H, int W, int C, int K, int stride, int pad, int dil) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C > STEM_MAX_C || K <= 0 || (K % 4) || K > 64 || stride <= 0 || dil <= 0 || pad < 0)
        return false;
    g.N = N; g.H = H; g.W = W; g.C = C; g.K = K; g.stride = stride; g.pad = pad; g.dil = dil;
    g.OH = (H + 2 * pad - dil * 2 - 1) / stride + 1;
    g.OW = (W + 2 * pad - dil * 2 - 1) / stride + 1;
    if (g.OH <= 0 || g.OW <= 0 || (size_t)N * g.OH * g.OW >= ((size_t)1 << 31)) return false;
    g.M = N * g.OH * g.OW;
    return true;
}

int stem_wgrad_blocks(int M) {
    int b = ceil_div(M, 256);                     // >= 256 pixels per block: 64 per lane
    return b > 1024 ? 1024 : b;
}

}  // namespace

extern "C" int semseg_stem_conv3x3_supported(int C, int K) { return C > 0 && C <= STEM_MAX_C && K > 0 && K <= 64 && (K % 4) == 0; }

extern "C" int semseg_stem_conv3x3_fwd(const float* x, int x_ld, const float* w, float* y, int y_ld, int N, int H, int W, int C, int K,
                                       int stride, int pad, int dil, void* stream) {
    StemGeom g;
    if (!x || !w || !y || x_ld < C || y_ld < K || (y_ld % 4) || !aligned16(y) || !stem_geom(g, N, H, W, C, K, stride, pad, dil))
        return SEMSEG_EINVAL;
    size_t blocks = ceil_div_sz((size_t)g.M * (K / 4), 256);
    if (blocks > 2048) blocks = 2048;           // the block's copy of the weights in LDS is loaded once per block: keep blocks long-lived
    const size_t smem = (size_t)STEM_TAPS * C * K * sizeof(float);
    hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream, x, x_ld, w, y, y_ld, g);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t semseg_stem_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int stride, int pad, int dil) {
    StemGeom g;
    if (!stem_geom(g, N, H, W, C, K, stride, pad, dil)) return 0;
    return (size_t)stem_wgrad_blocks(g.M) * K * STEM_TAPS * C * sizeof(double);
}

// dys: the h2 split planes of dy ([N*OH*OW] rows x K channels), as the BN backward kernel writes them
extern "C" int semseg_stem_conv3x3_wgrad_h2(const float* x, int x_ld, const void* dys, float* dw, int N, int H, int W, int C, int K,
                                            int stride, int pad, int dil, void* workspace, size_t workspace_bytes, void* stream) {
    StemGeom g;
    if (!x || !dys || !dw || x_ld < C || !aligned16(dys) || !stem_geom(g, N, H, W, C, K, stride, pad, dil)) return SEMSEG_EINVAL;
    const int nb = stem_wgrad_blocks(g.M);
    const int total = K * STEM_TAPS * C;
    if (!workspace || workspace_bytes < (size_t)nb * total * sizeof(double)) return SEMSEG_EWORKSPACE;
    const int px = ceil_div(g.M, nb);
    hipStream_t st = (hipStream_t)stream;
#define STEM_WG(CC)                                                                                                        \
    hipLaunchKernelGGL(stem_wgrad_partial_kernel<CC>, dim3(nb), dim3(K * WG_LANES), 0, st, x, x_ld, (const uint16_t*)dys, \
                       h2_plane_elems((size_t)g.M, K), split_pitch(K), (double*)workspace, g, px)
    switch (C) {
        case 1: STEM_WG(1); break;
        case 2: STEM_WG(2); break;
        case 3: STEM_WG(3); break;
        default: STEM_WG(4); break;
    }
#undef STEM_WG
    SEMSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(stem_wgrad_finish_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, (const double*)workspace, nb, total, h2_exp_ptr(dys, (size_t)g.M, K), dw);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}
