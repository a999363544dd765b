// Winograd F(2x2, 3x3) forward convolution on h2 split planes (stride 1, pad == dilation; the taps of a dilated 3x3 form a
// plain 3x3 on each of the dil x dil phase images).  2.25x fewer MFMA products than the direct implicit GEMM:
//   V = B^T d B   (input transform, fp32, one 4x4 input tile -> 16 "frequencies")        -> h2 planes, rows (f, tile)
//   U = G g G^T   (weight transform, fp32, once per optimiser step: weights_prep.hip)      -> h2 planes, rows (f, k)
//   M[f] = V[f] U[f]^T   16 independent [tiles x C] x [C x K] GEMMs = ONE batched launch of the LDS-DMA igemm kernels
//   z = A^T M A   (output transform, fp32, 2x2 outputs per tile)
// The exponent of the V planes needs no pass over V: |V| <= 4 max|x| (B^T has rows of at most two +-1), and x comes
// with a bound (BN outputs of the fused chain; max over the inputs of a concat).  Accuracy: same error class as the direct
// h2 path and torch's CPU fp32 convolution (tools/studies/winograd_h2_accuracy.py, tests/test_gpu_ops.py).
// Replaces nn.Conv2d forward at the 3x3 stride-1 call sites (resnet.py:61-66 dilated by models.py:209-251; models.py:163,
// 456-457) when ops enables it.
#include "common.h"
#include "batch.h"
#include "split_layout.h"

struct WinoGeom {
    int N, H, W, C, K, dil;
    int TH, TW;          // 2x2-output tiles per phase image (same grid for every phase; surplus tiles are empty)
    int tiles;           // N * dil * dil * TH * TW
};

static WinoGeom wino_geom(int N, int H, int W, int C, int K, int dil) {
    WinoGeom g;
    g.N = N; g.H = H; g.W = W; g.C = C; g.K = K; g.dil = dil;
    g.TH = (ceil_div(H, dil) + 1) / 2;
    g.TW = (ceil_div(W, dil) + 1) / 2;
    g.tiles = N * dil * dil * g.TH * g.TW;
    return g;
}

extern "C" int semseg_winograd_tiles(int N, int H, int W, int dil) {
    if (N <= 0 || H <= 0 || W <= 0 || dil <= 0) return 0;
    return wino_geom(N, H, W, 1, 1, dil).tiles;
}

// tile id -> (n, phase, tile coordinates)
__device__ __forceinline__ void wino_tile(const WinoGeom& g, int t, int& n, int& ph, int& pw, int& ty, int& tx) {
    tx = t % g.TW; t /= g.TW;
    ty = t % g.TH; t /= g.TH;
    pw = t % g.dil; t /= g.dil;
    ph = t % g.dil;
    n = t / g.dil;
}

constexpr int WINO_MAX_BOUNDS = 8;
struct WinoBounds {
    const float* p[WINO_MAX_BOUNDS];
    int n;
};

// one thread = 4 channels of one tile: 16 float4 loads, 32 + 32 adds per channel, 16 rows x 8 B per plane
struct wino_input_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ x, int x_ld, uint16_t* __restrict__ planes,
                                                         size_t plane, int pitch, int* __restrict__ hdr, WinoGeom g, int Cp,
                                                         WinoBounds bounds) {
    // exponent from the bound 4 * max(bounds of x)
    float b = 0.f;
    bool bad = false;
    for (int i = 0; i < bounds.n; ++i) {
        const float v = bounds.p[i][0];
        bad = bad || !(v == v);
        b = fmaxf(b, v);
    }
    const uint32_t bits = bad ? 0x7fc00000u : __float_as_uint(4.0f * b);
    const int ex = h2_exponent(bits);
    const float sc = pow2i(ex);
    if (blockIdx.x == 0) {
        if (threadIdx.x < SPLIT_ZERO_TAIL_BYTES / 16) reinterpret_cast<uint4*>(planes + H2_NP * plane)[threadIdx.x] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) hdr[0] = ex;
    }
    const int G4 = Cp >> 2;
    const size_t total = (size_t)g.tiles * G4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx / G4);
        const int c = (int)(idx - (size_t)t * G4) << 2;
        int n, ph, pw, ty, tx;
        wino_tile(g, t, n, ph, pw, ty, tx);
        float4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int qh = 2 * ty - 1 + a;                    // row in the phase image
            const int h = qh * g.dil + ph;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const int qw = 2 * tx - 1 + bb;
                const int w = qw * g.dil + pw;
                const bool ok = (c < g.C) && (qh >= 0) && (qw >= 0) && (h < g.H) && (w < g.W);
                d[a][bb] = ok ? *reinterpret_cast<const float4*>(x + ((size_t)(n * g.H + h) * g.W + w) * x_ld + c) : f4zero();
            }
        }
        // B^T d (columns), then (.) B (rows)
        float4 v[4][4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const float4 d0 = d[0][bb], d1 = d[1][bb], d2 = d[2][bb], d3 = d[3][bb];
            v[0][bb] = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
            v[1][bb] = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
            v[2][bb] = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
            v[3][bb] = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 t0 = v[a][0], t1 = v[a][1], t2 = v[a][2], t3 = v[a][3];
            const float4 o[4] = {make_float4(t0.x - t2.x, t0.y - t2.y, t0.z - t2.z, t0.w - t2.w),
                                 make_float4(t1.x + t2.x, t1.y + t2.y, t1.z + t2.z, t1.w + t2.w),
                                 make_float4(t2.x - t1.x, t2.y - t1.y, t2.z - t1.z, t2.w - t1.w),
                                 make_float4(t1.x - t3.x, t1.y - t3.y, t1.z - t3.z, t1.w - t3.w)};
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const int f = a * 4 + bb;
                const float e[4] = {o[bb].x * sc, o[bb].y * sc, o[bb].z * sc, o[bb].w * sc};
                f16x4 p0, p1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    _Float16 hi, lo;
                    h2_split_of(e[q], hi, lo);
                    p0[q] = hi;
                    p1[q] = lo;
                }
                const size_t off = ((size_t)f * g.tiles + t) * pitch + c;
                *reinterpret_cast<f16x4*>(planes + off) = p0;
                *reinterpret_cast<f16x4*>(planes + plane + off) = p1;
            }
        }
    }
    }
};

extern "C" int semseg_winograd_input_h2(const float* x, int x_ld, const float* const* bounds_host, int nbounds, void* v_planes,
                                        int N, int H, int W, int C, int dil, void* stream) {
    if (!x || !v_planes || !bounds_host || nbounds <= 0 || nbounds > WINO_MAX_BOUNDS || N <= 0 || H <= 0 || W <= 0 || C <= 0 ||
        (C % 4) || (x_ld % 4) || x_ld < C || dil <= 0 || !aligned16(x) || !aligned16(v_planes))
        return SEMSEG_EINVAL;
    const WinoGeom g = wino_geom(N, H, W, C, 1, dil);
    const size_t rows = (size_t)16 * g.tiles;
    const int Cp = round_up32(C), pitch = split_pitch(C);
    const size_t plane = h2_plane_elems(rows, C);
    if ((size_t)2 * H2_NP * plane + SPLIT_ZERO_TAIL_BYTES >= ((size_t)1 << 31)) return SEMSEG_EINVAL;     // 32-bit DMA offsets
    WinoBounds b;
    b.n = nbounds;
    for (int i = 0; i < WINO_MAX_BOUNDS; ++i) b.p[i] = i < nbounds ? bounds_host[i] : nullptr;
    for (int i = 0; i < nbounds; ++i)
        if (!b.p[i]) return SEMSEG_EINVAL;
    int* hdr = const_cast<int*>(h2_exp_ptr(v_planes, rows, C));
    size_t blocks = ceil_div_sz((size_t)g.tiles * (Cp / 4), 256);
    if (blocks > 16384) blocks = 16384;
    SEMSEG_LAUNCH_BODY((wino_input_kernel_body), dim3((unsigned)blocks), 0, (hipStream_t)stream, x, x_ld, (uint16_t*)v_planes,
                       plane, pitch, hdr, g, Cp, b);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// The same input transform with the SOURCE given as h2 planes (the data gradient in the Winograd domain: its input is dz, which
// the BN backward kernel writes as planes only).  p0 + p1 is the value to 22 bits in the scaled domain of the planes;
// |B^T d B| <= 4 max|d|, so the transform is formed with a factor 1/4 in that domain (exact in fp32) and the exponent of V is
// the exponent of the source minus 2 -- no bound, no rescaling.
struct wino_input_planes_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const uint16_t* __restrict__ src, size_t src_plane, int src_pitch,
                                                                const int* __restrict__ src_hdr, uint16_t* __restrict__ planes,
                                                                size_t plane, int pitch, int* __restrict__ hdr, WinoGeom g, int Cp) {
    if (blockIdx.x == 0) {
        if (threadIdx.x < SPLIT_ZERO_TAIL_BYTES / 16) reinterpret_cast<uint4*>(planes + H2_NP * plane)[threadIdx.x] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) hdr[0] = src_hdr[0] - 2;
    }
    const int G4 = Cp >> 2;
    const size_t total = (size_t)g.tiles * G4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx / G4);
        const int c = (int)(idx - (size_t)t * G4) << 2;
        int n, ph, pw, ty, tx;
        wino_tile(g, t, n, ph, pw, ty, tx);
        float4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int qh = 2 * ty - 1 + a;
            const int h = qh * g.dil + ph;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const int qw = 2 * tx - 1 + bb;
                const int w = qw * g.dil + pw;
                const bool ok = (c < g.C) && (qh >= 0) && (qw >= 0) && (h < g.H) && (w < g.W);
                float4 v = f4zero();
                if (ok) {
                    const size_t off = ((size_t)(n * g.H + h) * g.W + w) * src_pitch + c;
                    const f16x4 p0 = *reinterpret_cast<const f16x4*>(src + off);
                    const f16x4 p1 = *reinterpret_cast<const f16x4*>(src + src_plane + off);
                    v = make_float4(((float)p0[0] + (float)p1[0]) * 0.25f, ((float)p0[1] + (float)p1[1]) * 0.25f,
                                    ((float)p0[2] + (float)p1[2]) * 0.25f, ((float)p0[3] + (float)p1[3]) * 0.25f);
                }
                d[a][bb] = v;
            }
        }
        float4 v[4][4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const float4 d0 = d[0][bb], d1 = d[1][bb], d2 = d[2][bb], d3 = d[3][bb];
            v[0][bb] = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
            v[1][bb] = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
            v[2][bb] = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
            v[3][bb] = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 t0 = v[a][0], t1 = v[a][1], t2 = v[a][2], t3 = v[a][3];
            const float4 o[4] = {make_float4(t0.x - t2.x, t0.y - t2.y, t0.z - t2.z, t0.w - t2.w),
                                 make_float4(t1.x + t2.x, t1.y + t2.y, t1.z + t2.z, t1.w + t2.w),
                                 make_float4(t2.x - t1.x, t2.y - t1.y, t2.z - t1.z, t2.w - t1.w),
                                 make_float4(t1.x - t3.x, t1.y - t3.y, t1.z - t3.z, t1.w - t3.w)};
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const float e[4] = {o[bb].x, o[bb].y, o[bb].z, o[bb].w};
                f16x4 p0, p1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    _Float16 hi, lo;
                    h2_split_of(e[q], hi, lo);
                    p0[q] = hi;
                    p1[q] = lo;
                }
                const size_t off = ((size_t)(a * 4 + bb) * g.tiles + t) * pitch + c;
                *reinterpret_cast<f16x4*>(planes + off) = p0;
                *reinterpret_cast<f16x4*>(planes + plane + off) = p1;
            }
        }
    }
    }
};

// src_planes: h2 planes of a [N*H*W rows] x [C channels] tensor (e.g. dz of a conv whose data gradient runs in the Winograd domain)
extern "C" int semseg_winograd_input_planes_h2(const void* src_planes, void* v_planes, int N, int H, int W, int C, int dil,
                                               void* stream) {
    if (!src_planes || !v_planes || N <= 0 || H <= 0 || W <= 0 || C <= 0 || dil <= 0 || !aligned16(src_planes) || !aligned16(v_planes))
        return SEMSEG_EINVAL;
    const WinoGeom g = wino_geom(N, H, W, C, 1, dil);
    const size_t P = (size_t)N * H * W, rows = (size_t)16 * g.tiles;
    const int Cp = round_up32(C), pitch = split_pitch(C);
    const size_t src_plane = h2_plane_elems(P, C), plane = h2_plane_elems(rows, C);
    if ((size_t)2 * H2_NP * plane + SPLIT_ZERO_TAIL_BYTES >= ((size_t)1 << 31)) return SEMSEG_EINVAL;
    const int* src_hdr = h2_exp_ptr(src_planes, P, C);
    int* hdr = const_cast<int*>(h2_exp_ptr(v_planes, rows, C));
    size_t blocks = ceil_div_sz((size_t)g.tiles * (Cp / 4), 256);
    if (blocks > 16384) blocks = 16384;
    SEMSEG_LAUNCH_BODY((wino_input_planes_kernel_body), dim3((unsigned)blocks), 0, (hipStream_t)stream, (const uint16_t*)src_planes,
                       src_plane, pitch, src_hdr, (uint16_t*)v_planes, plane, pitch, hdr, g, Cp);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// one thread = 4 output channels of one tile: 16 float4 loads of M, 2x2 outputs
struct wino_output_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ M, float* __restrict__ z, int z_ld, WinoGeom g) {
    const int K4 = g.K >> 2;
    const size_t total = (size_t)g.tiles * K4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx / K4);
        const int k = (int)(idx - (size_t)t * K4) << 2;
        int n, ph, pw, ty, tx;
        wino_tile(g, t, n, ph, pw, ty, tx);
        float4 m[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
                m[a][bb] = *reinterpret_cast<const float4*>(M + ((size_t)(a * 4 + bb) * g.tiles + t) * g.K + k);
        // A^T m (columns): s0 = m0 + m1 + m2, s1 = m1 - m2 - m3; then (.) A (rows)
        float4 s[2][4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const float4 m0 = m[0][bb], m1 = m[1][bb], m2 = m[2][bb], m3 = m[3][bb];
            s[0][bb] = make_float4(m0.x + m1.x + m2.x, m0.y + m1.y + m2.y, m0.z + m1.z + m2.z, m0.w + m1.w + m2.w);
            s[1][bb] = make_float4(m1.x - m2.x - m3.x, m1.y - m2.y - m3.y, m1.z - m2.z - m3.z, m1.w - m2.w - m3.w);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 s0 = s[i][0], s1 = s[i][1], s2 = s[i][2], s3 = s[i][3];
            const float4 y0 = make_float4(s0.x + s1.x + s2.x, s0.y + s1.y + s2.y, s0.z + s1.z + s2.z, s0.w + s1.w + s2.w);
            const float4 y1 = make_float4(s1.x - s2.x - s3.x, s1.y - s2.y - s3.y, s1.z - s2.z - s3.z, s1.w - s2.w - s3.w);
            const int h = (2 * ty + i) * g.dil + ph;
            const int w0 = (2 * tx) * g.dil + pw, w1 = (2 * tx + 1) * g.dil + pw;
            if (h < g.H) {
                if (w0 < g.W) *reinterpret_cast<float4*>(z + ((size_t)(n * g.H + h) * g.W + w0) * z_ld + k) = y0;
                if (w1 < g.W) *reinterpret_cast<float4*>(z + ((size_t)(n * g.H + h) * g.W + w1) * z_ld + k) = y1;
            }
        }
    }
    }
};

extern "C" int semseg_winograd_output(const float* M, float* z, int z_ld, int N, int H, int W, int K, int dil, void* stream) {
    if (!M || !z || N <= 0 || H <= 0 || W <= 0 || K <= 0 || (K % 4) || (z_ld % 4) || z_ld < K || dil <= 0 || !aligned16(M) ||
        !aligned16(z))
        return SEMSEG_EINVAL;
    const WinoGeom g = wino_geom(N, H, W, 1, K, dil);
    size_t blocks = ceil_div_sz((size_t)g.tiles * (K / 4), 256);
    if (blocks > 16384) blocks = 16384;
    SEMSEG_LAUNCH_BODY((wino_output_kernel_body), dim3((unsigned)blocks), 0, (hipStream_t)stream, M, z, z_ld, g);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Winograd weight gradient (same layers):  dU[f] = dM[f]^T V[f]  with  dM = A dY A^T  (the adjoint of the output
// transform; 2x2 output-gradient tile -> 16 frequencies) and V the input transform kept from the forward; then
// dg = G^T dU G (the adjoint of the weight transform).  dY is read from the h2 planes the BN backward kernel wrote
// (dz has no fp32 copy); |dM| <= 4 max|dz|, so the planes of dM use the exponent of dz minus 2 and the values are formed in
// the scaled domain: 0.25 * (+-p +- p ...) of the fp16 parts (exact in fp32) -- no bound pass, no rescaling.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino_dm_kernel(const uint16_t* __restrict__ dzp, size_t dz_plane, int dz_pitch,
                                                      const int* __restrict__ dz_hdr, uint16_t* __restrict__ planes, size_t plane,
                                                      int pitch, int* __restrict__ hdr, WinoGeom g, int Kp) {
    if (blockIdx.x == 0) {
        if (threadIdx.x < SPLIT_ZERO_TAIL_BYTES / 16) reinterpret_cast<uint4*>(planes + H2_NP * plane)[threadIdx.x] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) hdr[0] = dz_hdr[0] - 2;
    }
    const int G4 = Kp >> 2;
    const size_t total = (size_t)g.tiles * G4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx / G4);
        const int k = (int)(idx - (size_t)t * G4) << 2;
        int n, ph, pw, ty, tx;
        wino_tile(g, t, n, ph, pw, ty, tx);
        float y[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int h = (2 * ty + i) * g.dil + ph, w = (2 * tx + j) * g.dil + pw;
                const bool ok = (k < g.K) && (h < g.H) && (w < g.W);
                f16x4 p0, p1;
                if (ok) {
                    const size_t off = ((size_t)(n * g.H + h) * g.W + w) * dz_pitch + k;
                    p0 = *reinterpret_cast<const f16x4*>(dzp + off);
                    p1 = *reinterpret_cast<const f16x4*>(dzp + dz_plane + off);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) y[i][j][e] = ok ? ((float)p0[e] + (float)p1[e]) * 0.25f : 0.f;
            }
        // A y (rows): m0 = y0, m1 = y0 + y1, m2 = y0 - y1, m3 = -y1; then (.) A^T (columns)
        float a[4][2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y0 = y[0][j][e], y1 = y[1][j][e];
                a[0][j][e] = y0; a[1][j][e] = y0 + y1; a[2][j][e] = y0 - y1; a[3][j][e] = -y1;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float m[4][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a0 = a[i][0][e], a1 = a[i][1][e];
                m[0][e] = a0; m[1][e] = a0 + a1; m[2][e] = a0 - a1; m[3][e] = -a1;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f16x4 q0, q1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 hi, lo;
                    h2_split_of(m[j][e], hi, lo);
                    q0[e] = hi;
                    q1[e] = lo;
                }
                const size_t off = ((size_t)(i * 4 + j) * g.tiles + t) * pitch + k;
                *reinterpret_cast<f16x4*>(planes + off) = q0;
                *reinterpret_cast<f16x4*>(planes + plane + off) = q1;
            }
        }
    }
}

extern "C" int semseg_winograd_dm_h2(const void* dz_planes, void* dm_planes, int N, int H, int W, int K, int dil, void* stream) {
    if (!dz_planes || !dm_planes || N <= 0 || H <= 0 || W <= 0 || K <= 0 || dil <= 0 || !aligned16(dz_planes) ||
        !aligned16(dm_planes))
        return SEMSEG_EINVAL;
    const WinoGeom g = wino_geom(N, H, W, 1, K, dil);
    const size_t P = (size_t)N * H * W, rows = (size_t)16 * g.tiles;
    const int Kp = round_up32(K), pitch = split_pitch(K);
    const size_t dz_plane = h2_plane_elems(P, K), plane = h2_plane_elems(rows, K);
    if ((size_t)2 * H2_NP * plane + SPLIT_ZERO_TAIL_BYTES >= ((size_t)1 << 31)) return SEMSEG_EINVAL;
    const int* dz_hdr = h2_exp_ptr(dz_planes, P, K);
    int* hdr = const_cast<int*>(h2_exp_ptr(dm_planes, rows, K));
    size_t blocks = ceil_div_sz((size_t)g.tiles * (Kp / 4), 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(wino_dm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dz_planes,
                       dz_plane, pitch, dz_hdr, (uint16_t*)dm_planes, plane, pitch, hdr, g, Kp);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// dw[k][r][s][c] = (G^T dU G)[r][s]:  one thread = 4 channels of one k
__global__ __launch_bounds__(256) void wino_dg_kernel(const float* __restrict__ dU, float* __restrict__ dw, int K, int C) {
    const int C4 = C >> 2;
    const size_t total = (size_t)K * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx / C4);
        const int c = (int)(idx - (size_t)k * C4) << 2;
        float4 u[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) u[i][j] = *reinterpret_cast<const float4*>(dU + ((size_t)(i * 4 + j) * K + k) * C + c);
        // G^T u (rows): r0 = u0 + (u1 + u2)/2, r1 = (u1 - u2)/2, r2 = (u1 + u2)/2 + u3; then (.) G (columns)
        float4 r[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 u0 = u[0][j], u1 = u[1][j], u2 = u[2][j], u3 = u[3][j];
            r[0][j] = make_float4(u0.x + 0.5f * (u1.x + u2.x), u0.y + 0.5f * (u1.y + u2.y), u0.z + 0.5f * (u1.z + u2.z), u0.w + 0.5f * (u1.w + u2.w));
            r[1][j] = make_float4(0.5f * (u1.x - u2.x), 0.5f * (u1.y - u2.y), 0.5f * (u1.z - u2.z), 0.5f * (u1.w - u2.w));
            r[2][j] = make_float4(0.5f * (u1.x + u2.x) + u3.x, 0.5f * (u1.y + u2.y) + u3.y, 0.5f * (u1.z + u2.z) + u3.z, 0.5f * (u1.w + u2.w) + u3.w);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float4 a0 = r[i][0], a1 = r[i][1], a2 = r[i][2], a3 = r[i][3];
            const float4 o0 = make_float4(a0.x + 0.5f * (a1.x + a2.x), a0.y + 0.5f * (a1.y + a2.y), a0.z + 0.5f * (a1.z + a2.z), a0.w + 0.5f * (a1.w + a2.w));
            const float4 o1 = make_float4(0.5f * (a1.x - a2.x), 0.5f * (a1.y - a2.y), 0.5f * (a1.z - a2.z), 0.5f * (a1.w - a2.w));
            const float4 o2 = make_float4(0.5f * (a1.x + a2.x) + a3.x, 0.5f * (a1.y + a2.y) + a3.y, 0.5f * (a1.z + a2.z) + a3.z, 0.5f * (a1.w + a2.w) + a3.w);
            float* d = dw + ((size_t)k * 9 + i * 3) * C + c;
            *reinterpret_cast<float4*>(d) = o0;
            *reinterpret_cast<float4*>(d + C) = o1;
            *reinterpret_cast<float4*>(d + 2 * (size_t)C) = o2;
        }
    }
}

extern "C" int semseg_winograd_dg(const float* dU, float* dw, int K, int C, void* stream) {
    if (!dU || !dw || K <= 0 || C <= 0 || (C % 4) || !aligned16(dU) || !aligned16(dw)) return SEMSEG_EINVAL;
    size_t blocks = ceil_div_sz((size_t)K * (C / 4), 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wino_dg_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dU, dw, K, C);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}
