// Batch-norm (training + eval) forward/backward for NHWC fp32 activations, gfx950.
// Replaces F.batch_norm at reference lib/nn/modules/batchnorm.py:58-61 (+ReLU / residual add of
// resnet.py:76-90, hrnet.py:45-61) and its autograd backward.  All kernels are HBM-bound streaming
// passes: float4 (16 B/lane) coalesced loads along the channel axis, fp32 per-thread strips, fp64 for
// every cross-thread / cross-block combine (so E[x^2]-E[x]^2 keeps fp32-class accuracy and the result
// is independent of the launch geometry up to one fp64 rounding).
//
// The statistics are exposed as raw sums (sum, sum of squares, count) in fp64 so a data-parallel caller
// can all-reduce them between `stats` and `finalize` -- this is what replaces the reference's
// SyncMaster/SlavePipe rendezvous (batchnorm.py:63-117, comm.py).
#include "common.h"
#include "peer_dev.h"
#include "split_layout.h"
#include "batch.h"
#include <stdlib.h>

// block = 256 threads arranged as cx channel-quads x py row lanes; grid = (quad groups, row chunks)
struct ColGeom {
    int cx, py, gx, gy, rows_per_block;
};
static ColGeom col_geom(int P, int C) {
    ColGeom g;
    const int quads = C / 4;
    g.cx = quads < 64 ? quads : 64;
    g.py = 256 / g.cx;
    g.gx = ceil_div(quads, g.cx);
    // ~512 blocks in total (2 per CU; 256 / 1024 / 2048 measured +0.05 ... +0.25 ms per step); every thread walks >= 8 rows
    const int target = 512;
    int gy = ceil_div(target, g.gx);
    const int min_rows = g.py * 8;
    if (gy > ceil_div(P, min_rows)) gy = ceil_div(P, min_rows);
    if (gy < 1) gy = 1;
    g.rows_per_block = ceil_div(P, gy);
    g.gy = ceil_div(P, g.rows_per_block);
    return g;
}

extern "C" size_t semseg_bn_workspace_bytes(int P, int C) {
    if (C % 4) return 0;
    const ColGeom g = col_geom(P, C);
    return (size_t)g.gy * 2 * C * sizeof(double);
}

// partial[by][0][c] = sum z, partial[by][1][c] = sum z^2 over the block's rows
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ z, int P, int C, int cx, int py,
                                                               int rows_per_block, double* __restrict__ partial) {
    extern __shared__ double red[];   // [py][cx][8]
    const int tx = threadIdx.x % cx, ty = threadIdx.x / cx;
    const int quad = blockIdx.x * cx + tx;
    const int c = quad * 4;
    const int row0 = blockIdx.y * rows_per_block;
    const int row1 = min(P, row0 + rows_per_block);
    // fp64 accumulation with EXACT squares: E[x^2]-E[x]^2 must survive var << mean^2 (the PPM scale-1 branch
    // normalises 2 pooled values per channel, models.py:447-450, where var ~ 1e-5 * mean^2)
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    const bool active = (ty < py) && (c < C);
    if (active) {
        for (int p = row0 + ty; p < row1; p += py) {
            const float4 v = *reinterpret_cast<const float4*>(z + (size_t)p * C + c);
            const double x = v.x, y = v.y, zz = v.z, w = v.w;
            a0 += x; a1 += y; a2 += zz; a3 += w;
            q0 = fma(x, x, q0); q1 = fma(y, y, q1); q2 = fma(zz, zz, q2); q3 = fma(w, w, q3);
        }
    }
    if (ty < py) {
        double* r = red + ((size_t)ty * cx + tx) * 8;
        r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3;
        r[4] = q0; r[5] = q1; r[6] = q2; r[7] = q3;
    }
    __syncthreads();
    if (ty == 0 && c < C) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int y = 0; y < py; ++y) {
            const double* r = red + ((size_t)y * cx + tx) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += r[e];
        }
        double* o = partial + (size_t)blockIdx.y * 2 * C;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[c + e] = a[e];
            o[C + c + e] = a[4 + e];
        }
    }
}

// out[j] = sum_by partial[by][j] for j in [0,2C); out[2C] = count (if count >= 0).
// block = 16 columns x 16 partial-lanes (a serial loop over hundreds of partials per thread was 80 us per BN layer)
__device__ __forceinline__ double colsum_block(const double* __restrict__ partial, int nparts, int C2, int j, int lane,
                                               double (*red)[17]) {
    double s = 0.0;
    if (j < C2) {
        // 8 loads in flight per round (rows past the end: clamped loads, values unused); additions in the order of the
        // plain loop
        for (int b = lane; b < nparts; b += 8 * 16) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)min(b + 16 * u, nparts - 1) * C2 + j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) s = (b + 16 * u < nparts) ? s + v[u] : s;
        }
    }
    red[lane][threadIdx.x & 15] = s;
    __syncthreads();
    double t = 0.0;
    if (lane == 0)
        for (int i = 0; i < 16; ++i) t += red[i][threadIdx.x & 15];
    return t;
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const double* __restrict__ partial, int nparts, int C2,
                                                            double* __restrict__ out, double count) {
    __shared__ double red[16][17];
    const int j = blockIdx.x * 16 + (threadIdx.x & 15);
    const int lane = threadIdx.x >> 4;
    const double t = colsum_block(partial, nparts, C2, j, lane, red);
    if (lane == 0 && j < C2) out[j] = t;
    if (blockIdx.x == 0 && threadIdx.x == 0 && count >= 0.0) out[C2] = count;
}

extern "C" int semseg_bn_stats(const float* z, int P, int C, double* stats, void* workspace, size_t workspace_bytes,
                               void* stream) {
    if (!z || !stats || P <= 0 || C <= 0 || (C % 4) || !aligned16(z)) return SEMSEG_EINVAL;
    const ColGeom g = col_geom(P, C);
    const size_t need = (size_t)g.gy * 2 * C * sizeof(double);
    if (!workspace || workspace_bytes < need) return SEMSEG_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t smem = (size_t)g.py * g.cx * 8 * sizeof(double);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(g.gx, g.gy), dim3(256), smem, st, z, P, C, g.cx, g.py,
                       g.rows_per_block, (double*)workspace);
    SEMSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(2 * C, 16)), dim3(256), 0, st, (const double*)workspace, g.gy,
                       2 * C, stats, (double)P);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float momentum, float eps, float* __restrict__ mean,
                                   float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift,
                                   int64_t* __restrict__ num_batches_tracked) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (c == 0 && num_batches_tracked) num_batches_tracked[0] += 1;
    const double n = stats[2 * C];
    const double mu = stats[c] / n;
    double var = stats[C + c] / n - mu * mu;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float muf = (float)mu;
    mean[c] = muf;
    invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - muf * sc;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * muf;
    if (running_var) {
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

extern "C" int semseg_bn_finalize(const double* stats, int C, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, int64_t* num_batches_tracked, float momentum, float eps, float* mean,
                                  float* invstd, float* scale, float* shift, void* stream) {
    if (!stats || !gamma || !beta || !mean || !invstd || !scale || !shift || C <= 0) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, stats, C, gamma, beta,
                       running_mean, running_var, momentum, eps, mean, invstd, scale, shift, num_batches_tracked);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps, int C,
                                      float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                      float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.0f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * is;
    mean[c] = rm[c];
    invstd[c] = is;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

extern "C" int semseg_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                     const float* running_var, float eps, int C, float* mean, float* invstd,
                                     float* scale, float* shift, void* stream) {
    if (!gamma || !beta || !running_mean || !running_var || C <= 0) return SEMSEG_EINVAL;
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, C, mean, invstd, scale, shift);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// y = act(z*scale + shift (+res)); one float4 per thread-iteration, grid-stride over P*C/4 quads
template <bool RES, bool RELU>
struct bn_apply_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ z, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ res,
                                                       int res_ld, float* __restrict__ y, int y_ld, int P, int C) {
    const int qpr = C / 4;
    const size_t total = (size_t)P * qpr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / qpr);
        const int c = (int)(i - (size_t)p * qpr) * 4;
        const float4 v = *reinterpret_cast<const float4*>(z + (size_t)p * C + c);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c);
        const float4 sh = *reinterpret_cast<const float4*>(shift + c);
        float4 o;
        o.x = fmaf(v.x, sc.x, sh.x); o.y = fmaf(v.y, sc.y, sh.y);
        o.z = fmaf(v.z, sc.z, sh.z); o.w = fmaf(v.w, sc.w, sh.w);
        if (RES) {
            const float4 r = *reinterpret_cast<const float4*>(res + (size_t)p * res_ld + c);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (RELU) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        *reinterpret_cast<float4*>(y + (size_t)p * y_ld + c) = o;
    }
    }
};

static inline int stream_blocks(size_t items) {
    size_t b = ceil_div_sz(items, 256);
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int semseg_bn_apply(const float* z, const float* scale, const float* shift, const float* residual, int res_ld,
                               int relu, float* y, int y_ld, int P, int C, void* stream) {
    if (!z || !scale || !shift || !y || P <= 0 || C <= 0 || (C % 4) || (y_ld % 4) || y_ld < C || !aligned16(z) || !aligned16(y))
        return SEMSEG_EINVAL;
    if (residual && ((res_ld % 4) || res_ld < C || !aligned16(residual))) return SEMSEG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = stream_blocks((size_t)P * (C / 4));
#define LAUNCH(R, A) SEMSEG_LAUNCH_BODY((bn_apply_kernel_body<R, A>), dim3(blocks), 0, st, z, scale, shift, residual, res_ld, y, y_ld, P, C)
    if (residual) { if (relu) LAUNCH(true, true); else LAUNCH(true, false); }
    else          { if (relu) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// partial sums of g and g*xhat, g = dy * (relu ? y>0 : 1)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, int dy_ld,
                                                             const float* __restrict__ y, int y_ld,
                                                             const float* __restrict__ z, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, int relu, int P, int C,
                                                             int cx, int py, int rows_per_block,
                                                             double* __restrict__ partial) {
    extern __shared__ double red[];
    const int tx = threadIdx.x % cx, ty = threadIdx.x / cx;
    const int c = (blockIdx.x * cx + tx) * 4;
    const int row0 = blockIdx.y * rows_per_block;
    const int row1 = min(P, row0 + rows_per_block);
    float4 s = f4zero(), sx = f4zero();
    if (ty < py && c < C) {
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        for (int p = row0 + ty; p < row1; p += py) {
            float4 g = *reinterpret_cast<const float4*>(dy + (size_t)p * dy_ld + c);
            if (relu) {
                const float4 yy = *reinterpret_cast<const float4*>(y + (size_t)p * y_ld + c);
                g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
                g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
            }
            const float4 v = *reinterpret_cast<const float4*>(z + (size_t)p * C + c);
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            sx.x += g.x * ((v.x - mu.x) * is.x); sx.y += g.y * ((v.y - mu.y) * is.y);
            sx.z += g.z * ((v.z - mu.z) * is.z); sx.w += g.w * ((v.w - mu.w) * is.w);
        }
    }
    if (ty < py) {
        double* r = red + ((size_t)ty * cx + tx) * 8;
        r[0] = s.x; r[1] = s.y; r[2] = s.z; r[3] = s.w;
        r[4] = sx.x; r[5] = sx.y; r[6] = sx.z; r[7] = sx.w;
    }
    __syncthreads();
    if (ty == 0 && c < C) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int yy = 0; yy < py; ++yy) {
            const double* r = red + ((size_t)yy * cx + tx) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += r[e];
        }
        double* o = partial + (size_t)blockIdx.y * 2 * C;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[c + e] = a[e];
            o[C + c + e] = a[4 + e];
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finish_kernel(const double* __restrict__ partial, int nparts, int C,
                                                            double* __restrict__ sums, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
    __shared__ double red[16][17];
    const int j = blockIdx.x * 16 + (threadIdx.x & 15);
    const int lane = threadIdx.x >> 4;
    const double s = colsum_block(partial, nparts, 2 * C, j, lane, red);
    if (lane != 0 || j >= 2 * C) return;
    sums[j] = s;
    if (j < C) { if (dbeta) dbeta[j] = (float)s; }
    else       { if (dgamma) dgamma[j - C] = (float)s; }
}

extern "C" int semseg_bn_bwd_reduce(const float* dy, int dy_ld, const float* y, int y_ld, const float* z,
                                    const float* mean, const float* invstd, int relu, int P, int C, double* sums,
                                    float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy || !z || !mean || !invstd || !sums || P <= 0 || C <= 0 || (C % 4) || (dy_ld % 4) || dy_ld < C) return SEMSEG_EINVAL;
    if (relu && (!y || (y_ld % 4) || y_ld < C)) return SEMSEG_EINVAL;
    const ColGeom g = col_geom(P, C);
    const size_t need = (size_t)g.gy * 2 * C * sizeof(double);
    if (!workspace || workspace_bytes < need) return SEMSEG_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t smem = (size_t)g.py * g.cx * 8 * sizeof(double);
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(g.gx, g.gy), dim3(256), smem, st, dy, dy_ld, y, y_ld, z, mean, invstd, relu,
                       P, C, g.cx, g.py, g.rows_per_block, (double*)workspace);
    SEMSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(ceil_div(2 * C, 16)), dim3(256), 0, st, (const double*)workspace, g.gy, C,
                       sums, dgamma, dbeta);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

template <bool TRAIN, bool RELU, bool DRES>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, int dy_ld,
                                                           const float* __restrict__ y, int y_ld,
                                                           const float* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const double* __restrict__ sums,
                                                           const double* __restrict__ count, float* __restrict__ dz,
                                                           float* __restrict__ dres, int P, int C) {
    const int qpr = C / 4;
    const size_t total = (size_t)P * qpr;
    const float inv_n = TRAIN ? (float)(1.0 / count[0]) : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / qpr);
        const int c = (int)(i - (size_t)p * qpr) * 4;
        float4 g = *reinterpret_cast<const float4*>(dy + (size_t)p * dy_ld + c);
        if (RELU) {
            const float4 yy = *reinterpret_cast<const float4*>(y + (size_t)p * y_ld + c);
            g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
            g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
        }
        if (DRES) *reinterpret_cast<float4*>(dres + (size_t)p * C + c) = g;
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
        float4 o;
        if (TRAIN) {
            const float4 v = *reinterpret_cast<const float4*>(z + (size_t)p * C + c);
            const float4 mu = *reinterpret_cast<const float4*>(mean + c);
            const float m0 = (float)sums[c] * inv_n, m1 = (float)sums[c + 1] * inv_n;
            const float m2 = (float)sums[c + 2] * inv_n, m3 = (float)sums[c + 3] * inv_n;
            const float x0 = (float)sums[C + c] * inv_n, x1 = (float)sums[C + c + 1] * inv_n;
            const float x2 = (float)sums[C + c + 2] * inv_n, x3 = (float)sums[C + c + 3] * inv_n;
            o.x = ga.x * is.x * (g.x - m0 - (v.x - mu.x) * is.x * x0);
            o.y = ga.y * is.y * (g.y - m1 - (v.y - mu.y) * is.y * x1);
            o.z = ga.z * is.z * (g.z - m2 - (v.z - mu.z) * is.z * x2);
            o.w = ga.w * is.w * (g.w - m3 - (v.w - mu.w) * is.w * x3);
        } else {
            o.x = ga.x * is.x * g.x; o.y = ga.y * is.y * g.y; o.z = ga.z * is.z * g.z; o.w = ga.w * is.w * g.w;
        }
        *reinterpret_cast<float4*>(dz + (size_t)p * C + c) = o;
    }
}

extern "C" int semseg_bn_bwd_apply(const float* dy, int dy_ld, const float* y, int y_ld, const float* z,
                                   const float* mean, const float* invstd, const float* gamma, const double* sums,
                                   const double* stats_count, int training, int relu, float* dz, float* dres, int P, int C,
                                   void* stream) {
    if (!dy || !invstd || !gamma || !dz || P <= 0 || C <= 0 || (C % 4) || (dy_ld % 4) || dy_ld < C) return SEMSEG_EINVAL;
    if (training && (!z || !mean || !sums || !stats_count)) return SEMSEG_EINVAL;
    if (relu && (!y || (y_ld % 4) || y_ld < C)) return SEMSEG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = stream_blocks((size_t)P * (C / 4));
#define LAUNCH(T, R, D) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, R, D>), dim3(blocks), dim3(256), 0, st, dy, dy_ld, y, y_ld, z, mean, invstd, gamma, sums, stats_count, dz, dres, P, C)
    const int key = (training ? 4 : 0) | (relu ? 2 : 0) | (dres ? 1 : 0);
    switch (key) {
        case 0: LAUNCH(false, false, false); break;
        case 1: LAUNCH(false, false, true); break;
        case 2: LAUNCH(false, true, false); break;
        case 3: LAUNCH(false, true, true); break;
        case 4: LAUNCH(true, false, false); break;
        case 5: LAUNCH(true, false, true); break;
        case 6: LAUNCH(true, true, false); break;
        default: LAUNCH(true, true, true); break;
    }
#undef LAUNCH
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// ================================================================================================
// BN kernels of the fused conv -> BN -> (ReLU) -> conv chain on the h2 path (ops.ConvBNActFn).
//
// The h2 split of a tensor needs a per-tensor exponent BEFORE the first element is written.  For tensors that BN
// produces the exponent does not need a pass over the result: BN is a per-channel affine map, so the per-channel
// min/max of z (gathered by the statistics pass that reads z anyway) give a RIGOROUS bound of |y|, and the sums of the
// backward reduction bound |dz|.  h2_exponent accepts any upper bound (split_layout.h), hence
//   forward : stats(+min/max) -> finalize(+bound -> exponent) -> apply writes y AND its split planes
//   backward: reduce(+max|g|) -> bound -> apply writes the split planes of dz ONLY (dz feeds nothing but the conv
//             gradients, which read planes)
// which removes the absmax + split passes (2 launches, 12 B/element) on both sides of every conv -> BN pair.
// ================================================================================================

// ReLU gate of a BN WITHOUT residual recomputed from z: y = max(fmaf(z, scale, shift), 0), so y > 0 <=> the very same fmaf
// > 0 -- bit-identical to the forward decision; saves the 4 B/element read of y in both backward passes.
__device__ __forceinline__ float4 relu_gate_from_z(const float4 v, const float* __restrict__ scale,
                                                   const float* __restrict__ shift) {
    const float4 sc = *reinterpret_cast<const float4*>(scale);
    const float4 sh = *reinterpret_cast<const float4*>(shift);
    return make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
}

extern "C" size_t semseg_bn_mm_workspace_bytes(int P, int C) {
    if (C % 4) return 0;
    const ColGeom g = col_geom(P, C);
    return (size_t)g.gy * 2 * C * (sizeof(double) + sizeof(float));
}

// as bn_stats_partial_kernel + per-channel min / max of z:  mm[by][0][c] = min, mm[by][1][c] = max
constexpr int ROWS_IN_FLIGHT = 8;
struct bn_stats_mm_partial_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ z, int P, int C, int cx, int py,
                                                                  int rows_per_block, double* __restrict__ partial,
                                                                  float* __restrict__ mm, float* __restrict__ zero_word) {
    extern __shared__ double red[];   // [py][cx][8] doubles, then [py][cx][8] floats
    // the fused finish kernel max-reduces its blocks' bounds into this word with atomics: start it at +0
    if (zero_word && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) zero_word[0] = 0.f;
    float* redf = reinterpret_cast<float*>(red + (size_t)py * cx * 8);
    const int tx = threadIdx.x % cx, ty = threadIdx.x / cx;
    const int quad = blockIdx.x * cx + tx;
    const int c = quad * 4;
    const int row0 = blockIdx.y * rows_per_block;
    const int row1 = min(P, row0 + rows_per_block);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    float4 lo = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    float4 hi = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    const bool active = (ty < py) && (c < C);
    if (active) {
        auto acc = [&](const float4 v) {
            const double x = v.x, y = v.y, zz = v.z, w = v.w;
            a0 += x; a1 += y; a2 += zz; a3 += w;
            q0 = fma(x, x, q0); q1 = fma(y, y, q1); q2 = fma(zz, zz, q2); q3 = fma(w, w, q3);
            lo.x = fminf(lo.x, v.x); lo.y = fminf(lo.y, v.y); lo.z = fminf(lo.z, v.z); lo.w = fminf(lo.w, v.w);
            hi.x = fmaxf(hi.x, v.x); hi.y = fmaxf(hi.y, v.y); hi.z = fmaxf(hi.z, v.z); hi.w = fmaxf(hi.w, v.w);
        };
        // ROWS_IN_FLIGHT loads are issued before the first is consumed (a thread walks only ~8 rows: one load per
        // iteration makes the pass latency-bound); the accumulation order is that of the plain loop
        int p = row0 + ty;
        for (; p + (ROWS_IN_FLIGHT - 1) * py < row1; p += ROWS_IN_FLIGHT * py) {
            float4 v[ROWS_IN_FLIGHT];
#pragma unroll
            for (int u = 0; u < ROWS_IN_FLIGHT; ++u) v[u] = *reinterpret_cast<const float4*>(z + (size_t)(p + u * py) * C + c);
#pragma unroll
            for (int u = 0; u < ROWS_IN_FLIGHT; ++u) acc(v[u]);
        }
        for (; p < row1; p += py) acc(*reinterpret_cast<const float4*>(z + (size_t)p * C + c));
    }
    if (ty < py) {
        double* r = red + ((size_t)ty * cx + tx) * 8;
        r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3;
        r[4] = q0; r[5] = q1; r[6] = q2; r[7] = q3;
        float* f = redf + ((size_t)ty * cx + tx) * 8;
        f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w;
        f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
    }
    __syncthreads();
    if (ty == 0 && c < C) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        float mn[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int y = 0; y < py; ++y) {
            const double* r = red + ((size_t)y * cx + tx) * 8;
            const float* f = redf + ((size_t)y * cx + tx) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += r[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mn[e] = fminf(mn[e], f[e]);
                mx[e] = fmaxf(mx[e], f[4 + e]);
            }
        }
        double* o = partial + (size_t)blockIdx.y * 2 * C;
        float* of = mm + (size_t)blockIdx.y * 2 * C;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[c + e] = a[e];
            o[C + c + e] = a[4 + e];
            of[c + e] = mn[e];
            of[C + c + e] = mx[e];
        }
    }
    }
};

// column sums of the fp64 partials (as colsum_finish_kernel) + column min (j < C) / max (j >= C) of the float partials
__global__ __launch_bounds__(256) void bn_stats_mm_finish_kernel(const double* __restrict__ partial,
                                                                 const float* __restrict__ mm, int nparts, int C,
                                                                 double* __restrict__ out, float* __restrict__ zmm,
                                                                 double count) {
    __shared__ double red[16][17];
    __shared__ float redf[16][17];
    const int C2 = 2 * C;
    const int j = blockIdx.x * 16 + (threadIdx.x & 15);
    const int lane = threadIdx.x >> 4;
    const bool is_min = j < C;
    float m = is_min ? INFINITY : -INFINITY;
    if (j < C2) {
#pragma unroll 8
        for (int b = lane; b < nparts; b += 16) {        // min / max: order-free, the compiler may batch the loads
            const float v = mm[(size_t)b * C2 + j];
            m = is_min ? fminf(m, v) : fmaxf(m, v);
        }
    }
    redf[lane][threadIdx.x & 15] = m;
    const double t = colsum_block(partial, nparts, C2, j, lane, red);     // contains the __syncthreads
    if (lane == 0 && j < C2) {
        out[j] = t;
        float r = redf[0][threadIdx.x & 15];
        for (int i = 1; i < 16; ++i) {
            const float v = redf[i][threadIdx.x & 15];
            r = is_min ? fminf(r, v) : fmaxf(r, v);
        }
        zmm[j] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && count >= 0.0) out[C2] = count;
}

extern "C" int semseg_bn_stats_mm(const float* z, int P, int C, double* stats, float* zmm, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    if (!z || !stats || !zmm || P <= 0 || C <= 0 || (C % 4) || !aligned16(z)) return SEMSEG_EINVAL;
    const ColGeom g = col_geom(P, C);
    const size_t need = (size_t)g.gy * 2 * C * (sizeof(double) + sizeof(float));
    if (!workspace || workspace_bytes < need) return SEMSEG_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double* partial = (double*)workspace;
    float* mm = reinterpret_cast<float*>(partial + (size_t)g.gy * 2 * C);
    const size_t smem = (size_t)g.py * g.cx * 8 * (sizeof(double) + sizeof(float));
    SEMSEG_LAUNCH_BODY((bn_stats_mm_partial_kernel_body), dim3(g.gx, g.gy), smem, st, z, P, C, g.cx, g.py,
                       g.rows_per_block, partial, mm, (float*)nullptr);
    SEMSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_stats_mm_finish_kernel, dim3(ceil_div(2 * C, 16)), dim3(256), 0, st, (const double*)partial,
                       (const float*)mm, g.gy, C, stats, zmm, (double)P);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// the sweep half of semseg_bn_stats_mm / semseg_bn_fwd_stats_fused alone: partial sums into `workspace`, nothing else touched.
// What a launch plan that splits the conv's reduction costs on top of the conv (the epilogue cannot gather statistics from
// split-K slabs): mit_semseg.tuner times conv + this for such candidates against conv-with-epilogue-statistics for the others.
extern "C" int semseg_bn_stats_mm_partial(const float* z, int P, int C, void* workspace, size_t workspace_bytes, void* stream) {
    if (!z || P <= 0 || C <= 0 || (C % 4) || !aligned16(z)) return SEMSEG_EINVAL;
    const ColGeom g = col_geom(P, C);
    const size_t need = (size_t)g.gy * 2 * C * (sizeof(double) + sizeof(float));
    if (!workspace || workspace_bytes < need) return SEMSEG_EWORKSPACE;
    double* partial = (double*)workspace;
    float* mm = reinterpret_cast<float*>(partial + (size_t)g.gy * 2 * C);
    const size_t smem = (size_t)g.py * g.cx * 8 * (sizeof(double) + sizeof(float));
    SEMSEG_LAUNCH_BODY((bn_stats_mm_partial_kernel_body), dim3(g.gx, g.gy), smem, (hipStream_t)stream, z, P, C, g.cx, g.py,
                       g.rows_per_block, partial, mm, (float*)nullptr);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// bn_finalize_kernel + the bound of |y| and the exponent of y's h2 planes.  One block (C <= a few thousand).
//   y = act(fmaf(z, scale, shift) (+ res)): fmaf and the fp32 add are monotone, so per channel the extremes of the BN
//   part are attained at zmin / zmax EXACTLY as bn_apply computes them, and |res| <= res_absmax.
__global__ __launch_bounds__(256) void bn_finalize_mm_kernel(const double* __restrict__ stats, const float* __restrict__ zmm,
                                                             int C, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, float momentum, float eps,
                                                             int relu, const float* __restrict__ res_absmax,
                                                             float* __restrict__ mean, float* __restrict__ invstd,
                                                             float* __restrict__ scale, float* __restrict__ shift,
                                                             int64_t* __restrict__ num_batches_tracked,
                                                             float* __restrict__ absmax_out, int* __restrict__ planes_hdr) {
    if (threadIdx.x == 0 && num_batches_tracked) num_batches_tracked[0] += 1;
    const double n = stats[2 * C];
    const float rmax = res_absmax ? res_absmax[0] : 0.f;
    float bound = 0.f;
    bool bad = false;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double mu = stats[c] / n;
        double var = stats[C + c] / n - mu * mu;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        const float muf = (float)mu;
        mean[c] = muf;
        invstd[c] = is;
        const float sc = gamma[c] * is;
        const float sh = beta[c] - muf * sc;
        scale[c] = sc;
        shift[c] = sh;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * muf;
        if (running_var) {
            const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
        const float e0 = fmaf(zmm[c], sc, sh), e1 = fmaf(zmm[C + c], sc, sh);
        const float hi = fmaxf(e0, e1) + rmax, lo = fminf(e0, e1) - rmax;
        const float b = relu ? fmaxf(hi, 0.f) : fmaxf(fabsf(hi), fabsf(lo));
        bad = bad || !(b == b) || !(e0 == e0) || !(e1 == e1);
        bound = fmaxf(bound, b);
    }
    uint32_t bits = bad ? 0x7fc00000u : __float_as_uint(bound);       // a NaN anywhere: NaN bound -> exponent 0
    bits = block_max_u32(bits);
    if (threadIdx.x == 0) {
        if (absmax_out) absmax_out[0] = __uint_as_float(bits);
        if (planes_hdr) planes_hdr[0] = h2_exponent(bits);
    }
}

extern "C" int semseg_bn_finalize_mm(const double* stats, const float* zmm, int C, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                     float eps, int relu, const float* res_absmax, float* mean, float* invstd, float* scale,
                                     float* shift, float* absmax_out, void* y_planes, int P, void* stream) {
    if (!stats || !zmm || !gamma || !beta || !mean || !invstd || !scale || !shift || C <= 0 || P <= 0) return SEMSEG_EINVAL;
    int* hdr = y_planes ? const_cast<int*>(h2_exp_ptr(y_planes, (size_t)P, C)) : nullptr;
    hipLaunchKernelGGL(bn_finalize_mm_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, zmm, C, gamma, beta,
                       running_mean, running_var, momentum, eps, relu, res_absmax, mean, invstd, scale, shift,
                       num_batches_tracked, absmax_out, hdr);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// Exponent of an h2 plane buffer inside the kernel that writes the planes: either the header word (set by an earlier
// bn_finalize_mm / bn_bwd_bound launch), or -- single-rank fused path -- the maximum of the per-block bounds that the
// merged finish kernel left in `blockbound` (bit patterns; every block redoes this <= 1 KB reduction, block 0 publishes
// the exponent word and the bound itself).
__device__ __forceinline__ int h2_exponent_from(int* __restrict__ hdr, const uint32_t* __restrict__ blockbound, int nbound,
                                                float* __restrict__ absmax_out, const bool first_block) {
    if (!blockbound) return hdr[0];
    uint32_t m = 0;
    for (int i = threadIdx.x; i < nbound; i += blockDim.x) m = max(m, blockbound[i]);
    m = block_max_u32(m);
    const int ex = h2_exponent(m);
    if (first_block && threadIdx.x == 0) {
        hdr[0] = ex;
        if (absmax_out) absmax_out[0] = __uint_as_float(m);
    }
    return ex;
}

// y = act(z*scale + shift (+res)) written as fp32 AND as h2 split planes (exponent from the header, set by
// bn_finalize_mm).  One thread = 8 channels of one pixel: 2 x 16 B fp32 stores + 2 x 16 B plane stores.
template <bool RES, bool RELU>
struct bn_apply_h2_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ z, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ res,
                                                          int res_ld, float* __restrict__ y, uint16_t* __restrict__ planes,
                                                          size_t plane, int pitch, int* __restrict__ hdr, int P, int C,
                                                          int Cp, const uint32_t* __restrict__ blockbound, int nbound,
                                                          float* __restrict__ absmax_out, uint8_t* __restrict__ gate) {
    // a thread keeps ONE group of 8 channels (scale / shift loaded once) and walks down the rows; consecutive threads own
    // consecutive channel groups of a row, then the next row -- the coalescing of a flat index
    // (32-bit index arithmetic: the host keeps gridDim.x * 256 below 2^31; a 64-bit division is a software loop on this part)
    const unsigned G = (unsigned)Cp >> 3;
    const unsigned threads = gridDim.x * blockDim.x;
    const unsigned rows_step = threads / G;
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = rows_step != 0 && tid < rows_step * G;
    const unsigned trow = tid / G;
    const int c = in_range ? (int)(tid - trow * G) << 3 : 0;
    const int prow = in_range ? (int)trow : 0;
    const bool live = in_range && c < C;
    // Everything that does not depend on the plane exponent is REQUESTED before the exponent's reduction (blockbound loads, two
    // barriers): scale / shift and the thread's first row.  The grid is one wave of blocks that all start together, so behind the
    // reduction each of these was a memory latency of its own at the head of a 5 - 10 us kernel.
    float4 sc0 = f4zero(), sc1 = f4zero(), sh0 = f4zero(), sh1 = f4zero();
    float4 v_first[2] = {f4zero(), f4zero()}, r_first[2] = {f4zero(), f4zero()};
    auto load_row = [&](size_t p, float4 (&v)[2], float4 (&r)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v[h] = *reinterpret_cast<const float4*>(z + p * C + c + 4 * h);
            if (RES) r[h] = *reinterpret_cast<const float4*>(res + p * res_ld + c + 4 * h);
        }
    };
    const bool has_first = live && (size_t)prow < (size_t)P;
    if (live) {
        sc0 = *reinterpret_cast<const float4*>(scale + c); sc1 = *reinterpret_cast<const float4*>(scale + c + 4);
        sh0 = *reinterpret_cast<const float4*>(shift + c); sh1 = *reinterpret_cast<const float4*>(shift + c + 4);
        if (has_first) load_row((size_t)prow, v_first, r_first);
    }
    const int ex = h2_exponent_from(hdr, blockbound, nbound, absmax_out, blockIdx.x == 0);
    const float sc2 = pow2i(ex);
    if (blockIdx.x == 0 && threadIdx.x < SPLIT_ZERO_TAIL_BYTES / 16)
        reinterpret_cast<uint4*>(planes + H2_NP * plane)[threadIdx.x] = make_uint4(0, 0, 0, 0);
    if (!in_range) return;
    if (c >= C) {                                  // padding groups of the plane pitch: zeros
        f16x8 zf;
#pragma unroll
        for (int e = 0; e < 8; ++e) zf[e] = (_Float16)0.f;
        for (size_t p = prow; p < (size_t)P; p += rows_step) {
            const size_t po = p * pitch + c;
            *reinterpret_cast<f16x8*>(planes + po) = zf;
            *reinterpret_cast<f16x8*>(planes + plane + po) = zf;
        }
        return;
    }
    auto finish_row = [&](size_t p, const float4 (&v)[2], const float4 (&r)[2]) {
        float o[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 sc = h ? sc1 : sc0, sh = h ? sh1 : sh0;
            float4 t;
            t.x = fmaf(v[h].x, sc.x, sh.x); t.y = fmaf(v[h].y, sc.y, sh.y);
            t.z = fmaf(v[h].z, sc.z, sh.z); t.w = fmaf(v[h].w, sc.w, sh.w);
            if (RES) { t.x += r[h].x; t.y += r[h].y; t.z += r[h].z; t.w += r[h].w; }
            if (RELU) {
                t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
            }
            if (y) *reinterpret_cast<float4*>(y + p * C + c + 4 * h) = t;      // y == nullptr: the planes are the only consumer
            o[4 * h] = t.x; o[4 * h + 1] = t.y; o[4 * h + 2] = t.z; o[4 * h + 3] = t.w;
        }
        if (RELU && gate) {                     // the ReLU decisions of these 8 channels, one bit each (y > 0): backward's gate
            unsigned bits = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) bits |= (o[e] > 0.f ? 1u : 0u) << e;
            gate[p * (size_t)(C >> 3) + (c >> 3)] = (uint8_t)bits;
        }
        f16x8 p0, p1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 a, r2;
            h2_split_of(o[e] * sc2, a, r2);
            p0[e] = a;
            p1[e] = r2;
        }
        const size_t po = p * pitch + c;
        *reinterpret_cast<f16x8*>(planes + po) = p0;
        *reinterpret_cast<f16x8*>(planes + plane + po) = p1;
    };
    if (has_first) finish_row((size_t)prow, v_first, r_first);
#pragma unroll 4
    for (size_t p = (size_t)prow + rows_step; p < (size_t)P; p += rows_step) {
        float4 v[2], r[2] = {f4zero(), f4zero()};
        load_row(p, v, r);
        finish_row(p, v, r);
    }
    }
};

static int bn_apply_h2_impl(const float* z, const float* scale, const float* shift, const float* residual, int res_ld,
                            int relu, float* y, void* y_planes, int P, int C, const void* blockbound, float* absmax_out,
                            void* stream, uint8_t* gate) {
    // y == nullptr (not with a gate bitmask: that form's y is the next block's shortcut): only the planes are written
    if (!z || !scale || !shift || (!y && gate) || !y_planes || P <= 0 || C <= 0 || (C % 8) || !aligned16(z) || (y && !aligned16(y)) ||
        !aligned16(y_planes))
        return SEMSEG_EINVAL;
    if (residual && ((res_ld % 4) || res_ld < C || !aligned16(residual))) return SEMSEG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int Cp = round_up32(C), pitch = split_pitch(C);
    const size_t plane = h2_plane_elems((size_t)P, C);
    int* hdr = const_cast<int*>(h2_exp_ptr(y_planes, (size_t)P, C));
    int blocks = stream_blocks((size_t)P * (Cp / 8));
    if (blockbound && blocks > 2048) blocks = 2048;      // every block redoes the bound reduction: keep them long-lived
    {   // 4 rows per thread amortise the per-channel loads (never below 1024 blocks = 4 per CU)
        const int quarter = (int)ceil_div_sz((size_t)P * (Cp / 8), 4 * 256);
        const int target = quarter > 1024 ? quarter : 1024;
        if (blocks > target) blocks = target;
    }
    const int nbound = ceil_div(C, 16);
#define LAUNCH(R, A) SEMSEG_LAUNCH_BODY((bn_apply_h2_kernel_body<R, A>), dim3(blocks), 0, st, z, scale, shift, residual, res_ld, y, (uint16_t*)y_planes, plane, pitch, hdr, P, C, Cp, (const uint32_t*)blockbound, nbound, absmax_out, gate)
    if (residual) { if (relu) LAUNCH(true, true); else LAUNCH(true, false); }
    else          { if (relu) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

extern "C" int semseg_bn_apply_h2(const float* z, const float* scale, const float* shift, const float* residual, int res_ld,
                                  int relu, float* y, void* y_planes, int P, int C, const void* blockbound, float* absmax_out,
                                  void* stream) {
    return bn_apply_h2_impl(z, scale, shift, residual, res_ld, relu, y, y_planes, P, C, blockbound, absmax_out, stream, nullptr);
}

// + the ReLU decisions as a bitmask (P x C/8 bytes, bit e of byte [p][c/8] = y[p][c + e] > 0): the backward kernels take it in place
// of y (y_ld = 0), 1/32 of the bytes
extern "C" int semseg_bn_apply_h2_gate(const float* z, const float* scale, const float* shift, const float* residual, int res_ld,
                                       int relu, float* y, void* y_planes, int P, int C, const void* blockbound,
                                       float* absmax_out, void* stream, unsigned char* gate) {
    if (!gate || !relu) return SEMSEG_EINVAL;
    return bn_apply_h2_impl(z, scale, shift, residual, res_ld, relu, y, y_planes, P, C, blockbound, absmax_out, stream, gate);
}

// ---- backward ----------------------------------------------------------------------------------
// as bn_bwd_partial_kernel + per-channel max |g|:  gm[by][c]
// GATE: 0 = no ReLU, 1 = ReLU gate recomputed from z (gscale/gshift), 2 = ReLU gate read from y, 3 = from the forward's bitmask
// (`y` then points to P x C/8 bytes, semseg_bn_apply_h2_gate)
template <int GATE>
struct bn_bwd_mm_partial_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ dy, int dy_ld,
                                                                const float* __restrict__ y, int y_ld,
                                                                const float* __restrict__ z, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, int relu, int P, int C,
                                                                int cx, int py, int rows_per_block,
                                                                double* __restrict__ partial, float* __restrict__ gm,
                                                                const float* __restrict__ gscale,
                                                                const float* __restrict__ gshift,
                                                                const float* __restrict__ dy2, int dy2_ld) {
    // dy2 != nullptr: the incoming gradient is the SUM dy + dy2 of the two gradients that meet at a fork (ops.ForkFn: block
    // output -> next block's first conv + its shortcut), added here -- one fp32 add per element, the value the add kernel would
    // have written -- instead of being materialised by a launch of its own and read back
    extern __shared__ double red[];   // [py][cx][8] doubles, then [py][cx][4] floats
    float* redf = reinterpret_cast<float*>(red + (size_t)py * cx * 8);
    const int tx = threadIdx.x % cx, ty = threadIdx.x / cx;
    const int c = (blockIdx.x * cx + tx) * 4;
    const int row0 = blockIdx.y * rows_per_block;
    const int row1 = min(P, row0 + rows_per_block);
    float4 s = f4zero(), sx = f4zero(), gx = f4zero();
    if (ty < py && c < C) {
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        constexpr bool gate_z = GATE == 1, gate_y = GATE == 2, gate_m = GATE == 3;
        const uint8_t* mask = reinterpret_cast<const uint8_t*>(y) + (c >> 3);
        const int mshift = c & 4, mld = C >> 3;
        auto gate_of = [&](int p) {               // the 4 decisions of this thread's channels as a positive / zero float4
            if (gate_y) return *reinterpret_cast<const float4*>(y + (size_t)p * y_ld + c);
            if (gate_m) {
                const unsigned b = (unsigned)mask[(size_t)p * mld] >> mshift;
                return make_float4((float)(b & 1u), (float)(b >> 1 & 1u), (float)(b >> 2 & 1u), (float)(b >> 3 & 1u));
            }
            return f4zero();
        };
        auto acc = [&](float4 g, const float4 v, const float4 yin) {
            if (GATE != 0) {
                const float4 yy = gate_z ? relu_gate_from_z(v, gscale + c, gshift + c) : yin;
                g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
                g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
            }
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            sx.x += g.x * ((v.x - mu.x) * is.x); sx.y += g.y * ((v.y - mu.y) * is.y);
            sx.z += g.z * ((v.z - mu.z) * is.z); sx.w += g.w * ((v.w - mu.w) * is.w);
            // NaN-propagating max (fmaxf would drop a NaN gradient)
            gx.x = (fabsf(g.x) > gx.x || g.x != g.x) ? fabsf(g.x) : gx.x;
            gx.y = (fabsf(g.y) > gx.y || g.y != g.y) ? fabsf(g.y) : gx.y;
            gx.z = (fabsf(g.z) > gx.z || g.z != g.z) ? fabsf(g.z) : gx.z;
            gx.w = (fabsf(g.w) > gx.w || g.w != g.w) ? fabsf(g.w) : gx.w;
        };
        // 4 rows (8-12 loads) in flight, accumulation order of the plain loop (see bn_stats_mm_partial_kernel)
        constexpr int U = 4;
        int p = row0 + ty;
        for (; p + (U - 1) * py < row1; p += U * py) {
            float4 g[U], v[U], yy[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                g[u] = *reinterpret_cast<const float4*>(dy + (size_t)(p + u * py) * dy_ld + c);
                v[u] = *reinterpret_cast<const float4*>(z + (size_t)(p + u * py) * C + c);
                yy[u] = gate_of(p + u * py);
            }
            if (dy2) {                    // block-uniform
                float4 h[U];
#pragma unroll
                for (int u = 0; u < U; ++u) h[u] = *reinterpret_cast<const float4*>(dy2 + (size_t)(p + u * py) * dy2_ld + c);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < U; ++u) { g[u].x += h[u].x; g[u].y += h[u].y; g[u].z += h[u].z; g[u].w += h[u].w; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) acc(g[u], v[u], yy[u]);
        }
        for (; p < row1; p += py) {
            float4 g1 = *reinterpret_cast<const float4*>(dy + (size_t)p * dy_ld + c);
            if (dy2) {
                const float4 h = *reinterpret_cast<const float4*>(dy2 + (size_t)p * dy2_ld + c);
                g1.x += h.x; g1.y += h.y; g1.z += h.z; g1.w += h.w;
            }
            acc(g1, *reinterpret_cast<const float4*>(z + (size_t)p * C + c), gate_of(p));
        }
    }
    if (ty < py) {
        double* r = red + ((size_t)ty * cx + tx) * 8;
        r[0] = s.x; r[1] = s.y; r[2] = s.z; r[3] = s.w;
        r[4] = sx.x; r[5] = sx.y; r[6] = sx.z; r[7] = sx.w;
        float* f = redf + ((size_t)ty * cx + tx) * 4;
        f[0] = gx.x; f[1] = gx.y; f[2] = gx.z; f[3] = gx.w;
    }
    __syncthreads();
    if (ty == 0 && c < C) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t m[4] = {0, 0, 0, 0};             // bit patterns: monotone for |g| >= 0, NaN sorts above inf
        for (int yy = 0; yy < py; ++yy) {
            const double* r = red + ((size_t)yy * cx + tx) * 8;
            const float* f = redf + ((size_t)yy * cx + tx) * 4;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += r[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = max(m[e], absbits(f[e]));
        }
        double* o = partial + (size_t)blockIdx.y * 2 * C;
        float* of = gm + (size_t)blockIdx.y * C;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[c + e] = a[e];
            o[C + c + e] = a[4 + e];
            of[c + e] = __uint_as_float(m[e]);
        }
    }
    }
};

__global__ __launch_bounds__(256) void bn_bwd_mm_finish_kernel(const double* __restrict__ partial, const float* __restrict__ gm,
                                                               int nparts, int C, double* __restrict__ sums,
                                                               float* __restrict__ gmax, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
    __shared__ double red[16][17];
    __shared__ uint32_t redu[16][17];
    const int j = blockIdx.x * 16 + (threadIdx.x & 15);
    const int lane = threadIdx.x >> 4;
    uint32_t m = 0;
    if (j < C) {
#pragma unroll 8
        for (int b = lane; b < nparts; b += 16) m = max(m, absbits(gm[(size_t)b * C + j]));
    }
    redu[lane][threadIdx.x & 15] = m;
    const double s = colsum_block(partial, nparts, 2 * C, j, lane, red);   // contains the __syncthreads
    if (lane != 0 || j >= 2 * C) return;
    sums[j] = s;
    if (j < C) {
        if (dbeta) dbeta[j] = (float)s;
        uint32_t r = 0;
        for (int i = 0; i < 16; ++i) r = max(r, redu[i][threadIdx.x & 15]);
        gmax[j] = __uint_as_float(r);
    } else if (dgamma) {
        dgamma[j - C] = (float)s;
    }
}

extern "C" int semseg_bn_bwd_reduce_mm(const float* dy, int dy_ld, const float* y, int y_ld, const float* z,
                                       const float* mean, const float* invstd, int relu, int P, int C, double* sums,
                                       float* gmax, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    if (!dy || !z || !mean || !invstd || !sums || !gmax || P <= 0 || C <= 0 || (C % 4) || (dy_ld % 4) || dy_ld < C)
        return SEMSEG_EINVAL;
    if (relu && (!y || (y_ld % 4) || y_ld < C)) return SEMSEG_EINVAL;
    const ColGeom g = col_geom(P, C);
    const size_t need = (size_t)g.gy * 2 * C * (sizeof(double) + sizeof(float));
    if (!workspace || workspace_bytes < need) return SEMSEG_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double* partial = (double*)workspace;
    float* gm = reinterpret_cast<float*>(partial + (size_t)g.gy * 2 * C);
    const size_t smem = (size_t)g.py * g.cx * (8 * sizeof(double) + 4 * sizeof(float));
    if (relu)
        SEMSEG_LAUNCH_BODY((bn_bwd_mm_partial_kernel_body<2>), dim3(g.gx, g.gy), smem, st, dy, dy_ld, y, y_ld, z, mean, invstd,
                           relu, P, C, g.cx, g.py, g.rows_per_block, partial, gm, (const float*)nullptr, (const float*)nullptr, nullptr, 0);
    else
        SEMSEG_LAUNCH_BODY((bn_bwd_mm_partial_kernel_body<0>), dim3(g.gx, g.gy), smem, st, dy, dy_ld, y, y_ld, z, mean, invstd,
                           relu, P, C, g.cx, g.py, g.rows_per_block, partial, gm, (const float*)nullptr, (const float*)nullptr, nullptr, 0);
    SEMSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_mm_finish_kernel, dim3(ceil_div(2 * C, 16)), dim3(256), 0, st, (const double*)partial,
                       (const float*)gm, g.gy, C, sums, gmax, dgamma, dbeta);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// exponent of the dz planes from a bound of |dz| = |gamma*invstd*(g - m - xhat*x)|:
//   |dz| <= |gamma| invstd (max|g| + |m| + max|xhat| |x|),  m = sums[c]/n, x = sums[C+c]/n,
//   max|xhat| = max(|zmin-mean|, |zmax-mean|) invstd.   The bound is evaluated in fp32 with a 2^-10 margin for the
//   roundings of bn_bwd_apply (the exponent leaves a further factor 2 of headroom below the fp16 maximum).
__global__ __launch_bounds__(256) void bn_bwd_bound_kernel(const double* __restrict__ sums, const double* __restrict__ count,
                                                           const float* __restrict__ gmax, const float* __restrict__ zmm,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, int C, int training,
                                                           int* __restrict__ planes_hdr) {
    const float inv_n = training ? (float)(1.0 / count[0]) : 0.f;
    uint32_t bits = 0;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float is = invstd[c];
        float b = gmax[c];
        if (training) {
            const float m = fabsf((float)sums[c] * inv_n), x = fabsf((float)sums[C + c] * inv_n);
            const float xh = fmaxf(fabsf(zmm[c] - mean[c]), fabsf(zmm[C + c] - mean[c])) * is;
            b = b + m + xh * x;
        }
        b = fabsf(gamma[c]) * is * b * 1.0009765625f;
        bits = max(bits, absbits(b));             // NaN sorts above inf -> exponent 0
    }
    bits = block_max_u32(bits);
    if (threadIdx.x == 0) planes_hdr[0] = h2_exponent(bits);
}

extern "C" int semseg_bn_bwd_bound(const double* sums, const double* stats_count, const float* gmax, const float* zmm,
                                   const float* mean, const float* invstd, const float* gamma, int C, int training,
                                   void* dz_planes, int P, void* stream) {
    if (!gmax || !invstd || !gamma || !dz_planes || C <= 0 || P <= 0) return SEMSEG_EINVAL;
    if (training && (!sums || !stats_count || !zmm || !mean)) return SEMSEG_EINVAL;
    int* hdr = const_cast<int*>(h2_exp_ptr(dz_planes, (size_t)P, C));
    hipLaunchKernelGGL(bn_bwd_bound_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, stats_count, gmax, zmm, mean,
                       invstd, gamma, C, training, hdr);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// bn_bwd_apply_kernel writing the h2 split planes of dz instead of fp32 dz (dres stays fp32)
template <bool TRAIN, bool RELU, bool DRES>
struct bn_bwd_apply_h2_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ dy, int dy_ld,
                                                              const float* __restrict__ y, int y_ld,
                                                              const float* __restrict__ z, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma,
                                                              const double* __restrict__ sums,
                                                              const double* __restrict__ count,
                                                              uint16_t* __restrict__ planes, size_t plane, int pitch,
                                                              int* __restrict__ hdr, float* __restrict__ dres, int P,
                                                              int C, int Cp, const float* __restrict__ gscale,
                                                              const float* __restrict__ gshift,
                                                              const uint32_t* __restrict__ blockbound, int nbound,
                                                              const float* __restrict__ dy2, int dy2_ld) {
    // dy2 != nullptr: the incoming gradient is dy + dy2 (see bn_bwd_mm_partial_kernel)
    // A thread keeps ONE group of 8 channels and walks down the rows: the per-channel terms (1/std, gamma, mean, the two batch
    // means, the gate's scale / shift: 26 loads) are fetched once instead of once per 8 outputs, and only dy / z (/ y) stream.
    // Consecutive threads own consecutive channel groups of a row, then the next row -- the coalescing of a flat index.
    const unsigned G = (unsigned)Cp >> 3;                 // 32-bit index arithmetic, as bn_apply_h2_kernel
    const unsigned threads = gridDim.x * blockDim.x;
    const unsigned rows_step = threads / G;               // rows covered per sweep; the last threads % G threads idle
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = rows_step != 0 && tid < rows_step * G;
    const unsigned trow = tid / G;
    const int c = in_range ? (int)(tid - trow * G) << 3 : 0;
    const int prow = in_range ? (int)trow : 0;
    const bool live = in_range && c < C;
    const bool gate_z = RELU && gscale;
    // The per-channel terms and the thread's first row are REQUESTED before the reduction that yields the plane exponent (as
    // bn_apply_h2_kernel: one wave of blocks that start together; each dependent load was a latency at the head of the kernel).
    struct Row { float4 g[2], h[2], v[2], y[2]; unsigned b; };
    auto load_row = [&](size_t p, Row& r) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            r.g[h] = *reinterpret_cast<const float4*>(dy + p * dy_ld + c + 4 * h);
            if (dy2) r.h[h] = *reinterpret_cast<const float4*>(dy2 + p * dy2_ld + c + 4 * h);      // grid-uniform
            if (TRAIN || gate_z) r.v[h] = *reinterpret_cast<const float4*>(z + p * C + c + 4 * h);
            if (RELU && !gate_z && y_ld) r.y[h] = *reinterpret_cast<const float4*>(y + p * y_ld + c + 4 * h);
        }
        // the forward's bitmask in place of y (semseg_bn_apply_h2_gate)
        if (RELU && !gate_z && !y_ld) r.b = reinterpret_cast<const uint8_t*>(y)[p * (size_t)(C >> 3) + (c >> 3)];
    };
    float is[8], ga[8], mu[8], m[8], x[8], gs[8], gh[8];      // m / x: the sums as floats until 1 / n is known (below)
    Row first;
    first.b = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        first.g[h] = f4zero(); first.h[h] = f4zero(); first.v[h] = f4zero(); first.y[h] = f4zero();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            is[4 * h + e] = 0.f; ga[4 * h + e] = 0.f; mu[4 * h + e] = 0.f; gs[4 * h + e] = 0.f; gh[4 * h + e] = 0.f;
            m[4 * h + e] = 0.f; x[4 * h + e] = 0.f;
        }
    }
    const bool has_first = live && (size_t)prow < (size_t)P;
    if (live) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cc = c + 4 * h;
            const float4 is4 = *reinterpret_cast<const float4*>(invstd + cc);
            const float4 ga4 = *reinterpret_cast<const float4*>(gamma + cc);
            is[4 * h] = is4.x; is[4 * h + 1] = is4.y; is[4 * h + 2] = is4.z; is[4 * h + 3] = is4.w;
            ga[4 * h] = ga4.x; ga[4 * h + 1] = ga4.y; ga[4 * h + 2] = ga4.z; ga[4 * h + 3] = ga4.w;
            if (TRAIN) {
                const float4 mu4 = *reinterpret_cast<const float4*>(mean + cc);
                mu[4 * h] = mu4.x; mu[4 * h + 1] = mu4.y; mu[4 * h + 2] = mu4.z; mu[4 * h + 3] = mu4.w;
                const double2 sa = *reinterpret_cast<const double2*>(sums + cc), sb = *reinterpret_cast<const double2*>(sums + cc + 2);
                const double2 xa = *reinterpret_cast<const double2*>(sums + C + cc), xb = *reinterpret_cast<const double2*>(sums + C + cc + 2);
                m[4 * h] = (float)sa.x; m[4 * h + 1] = (float)sa.y; m[4 * h + 2] = (float)sb.x; m[4 * h + 3] = (float)sb.y;
                x[4 * h] = (float)xa.x; x[4 * h + 1] = (float)xa.y; x[4 * h + 2] = (float)xb.x; x[4 * h + 3] = (float)xb.y;
            }
            if (RELU && gscale) {
                const float4 s4 = *reinterpret_cast<const float4*>(gscale + cc), h4 = *reinterpret_cast<const float4*>(gshift + cc);
                gs[4 * h] = s4.x; gs[4 * h + 1] = s4.y; gs[4 * h + 2] = s4.z; gs[4 * h + 3] = s4.w;
                gh[4 * h] = h4.x; gh[4 * h + 1] = h4.y; gh[4 * h + 2] = h4.z; gh[4 * h + 3] = h4.w;
            }
        }
        if (has_first) load_row((size_t)prow, first);
    }
    const double n_count = TRAIN ? count[0] : 1.0;
    const int ex = h2_exponent_from(hdr, blockbound, nbound, nullptr, blockIdx.x == 0);
    const float sc2 = pow2i(ex);
    if (blockIdx.x == 0 && threadIdx.x < SPLIT_ZERO_TAIL_BYTES / 16)
        reinterpret_cast<uint4*>(planes + H2_NP * plane)[threadIdx.x] = make_uint4(0, 0, 0, 0);
    if (!in_range) return;
    if (c >= C) {                                          // padding groups of the plane pitch: zeros
        f16x8 zf;
#pragma unroll
        for (int e = 0; e < 8; ++e) zf[e] = (_Float16)0.f;
        for (size_t p = prow; p < (size_t)P; p += rows_step) {
            const size_t po = p * pitch + c;
            *reinterpret_cast<f16x8*>(planes + po) = zf;
            *reinterpret_cast<f16x8*>(planes + plane + po) = zf;
        }
        return;
    }
    const float inv_n = TRAIN ? (float)(1.0 / n_count) : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { m[e] *= inv_n; x[e] *= inv_n; }        // (float)sum * inv_n, as before
    auto finish_row = [&](size_t p, const Row& r) {
        float g[8], v[8], yy[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 g4 = r.g[h];
            if (dy2) { g4.x += r.h[h].x; g4.y += r.h[h].y; g4.z += r.h[h].z; g4.w += r.h[h].w; }
            g[4 * h] = g4.x; g[4 * h + 1] = g4.y; g[4 * h + 2] = g4.z; g[4 * h + 3] = g4.w;
            const float4 v4 = (TRAIN || gate_z) ? r.v[h] : f4zero();
            v[4 * h] = v4.x; v[4 * h + 1] = v4.y; v[4 * h + 2] = v4.z; v[4 * h + 3] = v4.w;
            yy[4 * h] = r.y[h].x; yy[4 * h + 1] = r.y[h].y; yy[4 * h + 2] = r.y[h].z; yy[4 * h + 3] = r.y[h].w;
        }
        if (RELU && !gate_z && !y_ld) {
#pragma unroll
            for (int e = 0; e < 8; ++e) yy[e] = (float)(r.b >> e & 1u);
        }
        f16x8 p0, p1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float ge = g[e];
            if (RELU) {
                const float ye = gate_z ? fmaf(v[e], gs[e], gh[e]) : yy[e];        // relu_gate_from_z: the forward's own fmaf
                ge = ye > 0.f ? ge : 0.f;
            }
            g[e] = ge;
            const float t = TRAIN ? ga[e] * is[e] * (ge - m[e] - (v[e] - mu[e]) * is[e] * x[e]) : ga[e] * is[e] * ge;
            _Float16 a, rr;
            h2_split_of(t * sc2, a, rr);
            p0[e] = a;
            p1[e] = rr;
        }
        if (DRES) {
            *reinterpret_cast<float4*>(dres + p * C + c) = make_float4(g[0], g[1], g[2], g[3]);
            *reinterpret_cast<float4*>(dres + p * C + c + 4) = make_float4(g[4], g[5], g[6], g[7]);
        }
        const size_t po = p * pitch + c;
        *reinterpret_cast<f16x8*>(planes + po) = p0;
        *reinterpret_cast<f16x8*>(planes + plane + po) = p1;
    };
    if (has_first) finish_row((size_t)prow, first);
#pragma unroll 4
    for (size_t p = (size_t)prow + rows_step; p < (size_t)P; p += rows_step) {
        Row r;
        r.b = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) { r.h[h] = f4zero(); r.v[h] = f4zero(); r.y[h] = f4zero(); }
        load_row(p, r);
        finish_row(p, r);
    }
    }
};

static int bn_bwd_apply_h2_impl(const float* dy, int dy_ld, const float* dy2, int dy2_ld, const float* y, int y_ld, const float* z,
                                const float* mean, const float* invstd, const float* gamma, const double* sums,
                                const double* stats_count, int training, int relu, void* dz_planes, float* dres, int P,
                                int C, const float* gate_scale, const float* gate_shift, const void* blockbound,
                                void* stream) {
    if (dy2 && ((dy2_ld % 4) || dy2_ld < C || !aligned16(dy2))) return SEMSEG_EINVAL;
    if (!dy || !invstd || !gamma || !dz_planes || P <= 0 || C <= 0 || (C % 8) || (dy_ld % 4) || dy_ld < C ||
        !aligned16(dy) || !aligned16(dz_planes))
        return SEMSEG_EINVAL;
    if (training && (!z || !mean || !sums || !stats_count)) return SEMSEG_EINVAL;
    if (relu && gate_scale && (!gate_shift || !z)) return SEMSEG_EINVAL;
    if (relu && !gate_scale && (!y || (y_ld != 0 && ((y_ld % 4) || y_ld < C)))) return SEMSEG_EINVAL;     // y_ld 0: bitmask
    hipStream_t st = (hipStream_t)stream;
    const int Cp = round_up32(C), pitch = split_pitch(C);
    const size_t plane = h2_plane_elems((size_t)P, C);
    int* hdr = const_cast<int*>(h2_exp_ptr(dz_planes, (size_t)P, C));
    int blocks = stream_blocks((size_t)P * (Cp / 8));
    if (blockbound && blocks > 2048) blocks = 2048;
    // a thread keeps its channel group: 4 rows per thread amortise its per-channel loads (never below 1024 blocks = 4 per CU)
    const int quarter = (int)ceil_div_sz((size_t)P * (Cp / 8), 4 * 256);
    const int target = quarter > 1024 ? quarter : 1024;
    if (blocks > target) blocks = target;
    const int nbound = ceil_div(C, 16);
#define LAUNCH(T, R, D) SEMSEG_LAUNCH_BODY((bn_bwd_apply_h2_kernel_body<T, R, D>), dim3(blocks), 0, st, dy, dy_ld, y, y_ld, z, mean, invstd, gamma, sums, stats_count, (uint16_t*)dz_planes, plane, pitch, hdr, dres, P, C, Cp, gate_scale, gate_shift, (const uint32_t*)blockbound, nbound, dy2, dy2_ld)
    const int key = (training ? 4 : 0) | (relu ? 2 : 0) | (dres ? 1 : 0);
    switch (key) {
        case 0: LAUNCH(false, false, false); break;
        case 1: LAUNCH(false, false, true); break;
        case 2: LAUNCH(false, true, false); break;
        case 3: LAUNCH(false, true, true); break;
        case 4: LAUNCH(true, false, false); break;
        case 5: LAUNCH(true, false, true); break;
        case 6: LAUNCH(true, true, false); break;
        default: LAUNCH(true, true, true); break;
    }
#undef LAUNCH
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

extern "C" int semseg_bn_bwd_apply_h2(const float* dy, int dy_ld, const float* y, int y_ld, const float* z,
                                      const float* mean, const float* invstd, const float* gamma, const double* sums,
                                      const double* stats_count, int training, int relu, void* dz_planes, float* dres, int P,
                                      int C, const float* gate_scale, const float* gate_shift, const void* blockbound,
                                      void* stream) {
    return bn_bwd_apply_h2_impl(dy, dy_ld, nullptr, 0, y, y_ld, z, mean, invstd, gamma, sums, stats_count, training, relu, dz_planes,
                                dres, P, C, gate_scale, gate_shift, blockbound, stream);
}

// ... with the incoming gradient as two addends, dy + dy2 (semseg_bn_bwd_reduce_fused_sum2)
extern "C" int semseg_bn_bwd_apply_h2_sum2(const float* dy, int dy_ld, const float* dy2, int dy2_ld, const float* y, int y_ld,
                                           const float* z, const float* mean, const float* invstd, const float* gamma,
                                           const double* sums, const double* stats_count, int training, int relu, void* dz_planes,
                                           float* dres, int P, int C, const float* gate_scale, const float* gate_shift,
                                           const void* blockbound, void* stream) {
    if (!dy2) return SEMSEG_EINVAL;
    return bn_bwd_apply_h2_impl(dy, dy_ld, dy2, dy2_ld, y, y_ld, z, mean, invstd, gamma, sums, stats_count, training, relu, dz_planes,
                                dres, P, C, gate_scale, gate_shift, blockbound, stream);
}


// ================================================================================================
// Single-rank fused path: no all-reduce sits between the partial sums and their use, so the finish kernels also do the
// per-channel finalize / bound work and leave one bound per block (16 channels) for the apply kernel's prologue
// (h2_exponent_from): 3 launches per BN pass instead of 4.  Block = 16 channels x 16 partial lanes.
// ================================================================================================
// PEER (SyncBN over the one-node peer exchange, csrc/peer_dev.h): the per-channel [sum, sum^2] and the pixel count leave for the
// peers' inboxes as soon as the block has them and come back summed over the ranks -- the all-reduce of batchnorm.py:98-117
// INSIDE the finish kernel; min / max and the bounds stay rank-local (they only choose an exponent).
template <bool PEER>
struct bn_fwd_finish_fused_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const double* __restrict__ partial, const float* __restrict__ mm, int nparts, int C, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ running_mean,
    float* __restrict__ running_var, float momentum, float eps, int relu, const float* __restrict__ res_absmax,
    double* __restrict__ stats, float* __restrict__ zmm, float* __restrict__ mean, float* __restrict__ invstd,
    float* __restrict__ scale, float* __restrict__ shift, int64_t* __restrict__ num_batches_tracked,
    uint32_t* __restrict__ blockbound, float* __restrict__ absmax_out, semseg_peer::PeerArgs pa) {
    __shared__ double rs[16][17], rq[16][17];
    __shared__ float rlo[16][17], rhi[16][17];
    __shared__ double xs[3][16];
    __shared__ unsigned s_q;
    const int cl = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int C2 = 2 * C;
    double su = 0.0, sq = 0.0;
    float lo = INFINITY, hi = -INFINITY;
    if (PEER && threadIdx.x == 0) s_q = *pa.seq;
    // the per-channel terms of the thread that finishes channel c, requested BEFORE the partial rows: they were last touched a
    // step ago (HBM misses), and behind the reduction each of them was a memory latency of its own at the end of a ~5 us kernel
    const bool fin = lane == 0 && c < C;
    float p_gamma = 0.f, p_beta = 0.f, p_rmean = 0.f, p_rvar = 0.f, p_res = 0.f;
    if (fin) {
        p_gamma = gamma[c]; p_beta = beta[c];
        if (running_mean) p_rmean = running_mean[c];
        if (running_var) p_rvar = running_var[c];
        if (res_absmax) p_res = res_absmax[0];
    }
    if (c < C) {
        // 8 partial rows (32 loads) in flight per round; rows past the end are clamped loads whose values are not used.
        // Additions in the order of the plain loop (bit-identical sums).
        constexpr int U = 8;
        for (int b = lane; b < nparts; b += U * 16) {
            double ps[U], pq[U];
            float pl[U], ph[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t o = (size_t)min(b + 16 * u, nparts - 1) * C2;
                ps[u] = partial[o + c]; pq[u] = partial[o + C + c];
                pl[u] = mm[o + c]; ph[u] = mm[o + C + c];
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the loads together (the scheduler otherwise sinks each to its use)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = b + 16 * u < nparts;
                su = live ? su + ps[u] : su; sq = live ? sq + pq[u] : sq;
                lo = live ? fminf(lo, pl[u]) : lo; hi = live ? fmaxf(hi, ph[u]) : hi;
            }
        }
    }
    rs[lane][cl] = su; rq[lane][cl] = sq; rlo[lane][cl] = lo; rhi[lane][cl] = hi;
    __syncthreads();
    uint32_t bits = 0;
    double n = count;
    if (lane == 0 && c < C) {
        su = 0.0; sq = 0.0; lo = INFINITY; hi = -INFINITY;
        for (int i = 0; i < 16; ++i) {            // same summation order as colsum_block
            su += rs[i][cl]; sq += rq[i][cl];
            lo = fminf(lo, rlo[i][cl]); hi = fmaxf(hi, rhi[i][cl]);
        }
        if (PEER) { xs[0][cl] = su; xs[1][cl] = sq; }
    }
    if (PEER) {
        // threads (lane 0, cl) / (lane 1, cl) carry sum / sum^2 of channel cl through the exchange, thread (lane 2, 0) the pixel
        // count (pushed by block 0 only, gathered by every block)
        __syncthreads();
        const unsigned q = s_q;
        if (lane < 3 && (lane < 2 ? c < C : cl == 0)) {
            const int idx = lane == 2 ? C2 : lane * C + c;
            const double own = lane == 2 ? count : xs[lane][cl];
            if (lane < 2 || blockIdx.x == 0) semseg_peer::push(pa, q, idx, own);
            bool timed_out = false;
            const double tot = semseg_peer::gather_sum(pa, q, idx, own, wall_clock64(), timed_out);
            if (timed_out) atomicOr(pa.status, 1u);
            xs[lane][lane == 2 ? 0 : cl] = tot;
        }
        __syncthreads();
        n = xs[2][0];
    }
    if (lane == 0 && c < C) {
        if (PEER) { su = xs[0][cl]; sq = xs[1][cl]; }
        stats[c] = su; stats[C + c] = sq;
        zmm[c] = lo; zmm[C + c] = hi;
        const double mu = su / n;
        double var = sq / n - mu * mu;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        const float muf = (float)mu;
        mean[c] = muf;
        invstd[c] = is;
        const float sc = p_gamma * is;
        const float sh = p_beta - muf * sc;
        scale[c] = sc;
        shift[c] = sh;
        if (running_mean) running_mean[c] = (1.f - momentum) * p_rmean + momentum * muf;
        if (running_var) {
            const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
            running_var[c] = (1.f - momentum) * p_rvar + momentum * (float)unbiased;
        }
        const float rmax = p_res;
        const float e0 = fmaf(lo, sc, sh), e1 = fmaf(hi, sc, sh);
        const float bh = fmaxf(e0, e1) + rmax, bl = fminf(e0, e1) - rmax;
        const float b = relu ? fmaxf(bh, 0.f) : fmaxf(fabsf(bh), fabsf(bl));
        const bool bad = !(b == b) || !(e0 == e0) || !(e1 == e1);
        bits = bad ? 0x7fc00000u : __float_as_uint(b);
    }
    bits = block_max_u32(bits);
    if (threadIdx.x == 0) {
        blockbound[blockIdx.x] = bits;
        // bit patterns of non-negative floats order like the floats, the NaN pattern above all of them (zeroed by the partial kernel)
        if (absmax_out) atomicMax(reinterpret_cast<unsigned*>(absmax_out), bits);
        if (blockIdx.x == 0) {
            stats[C2] = n;
            if (num_batches_tracked) num_batches_tracked[0] += 1;
        }
        if (PEER) semseg_peer::advance(pa, s_q, gridDim.x);
    }
    }
};
// the PEER form as a kernel of its own: it waits for the other ranks inside the kernel, so it is never recorded, and the
// co-residency query (exchange_capacity) needs its symbol
__global__ __launch_bounds__(256) void bn_fwd_finish_fused_kernel_peer(const double* __restrict__ partial, const float* __restrict__ mm, int nparts, int C, double count, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps, int relu, const float* __restrict__ res_absmax, double* __restrict__ stats, float* __restrict__ zmm, float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift, int64_t* __restrict__ num_batches_tracked, uint32_t* __restrict__ blockbound, float* __restrict__ absmax_out, semseg_peer::PeerArgs pa) {
    bn_fwd_finish_fused_kernel_body<true>::run(semseg_batch::U3{blockIdx.x, blockIdx.y, blockIdx.z}, semseg_batch::U3{gridDim.x, gridDim.y, gridDim.z}, partial, mm, nparts, C, count, gamma, beta, running_mean, running_var, momentum, eps, relu, res_absmax, stats, zmm, mean, invstd, scale, shift, num_batches_tracked, blockbound, absmax_out, pa);
}

// Co-residency guard of the kernels that exchange with their peers while they run (csrc/peer_dev.h): block b of a rank spins
// until block b of every other rank has pushed, so a rank must be able to hold its whole exchanging grid on the device at once
// -- then no block ever waits for one that could not be dispatched, whatever order the hardware dispatches blocks in.  (With
// in-order dispatch the lowest unfinished block index is resident on every rank and progress follows anyway; this guard removes
// the reliance on that.)  ceil(C/16) blocks of 256 threads against CUs x resident blocks per CU: 256 blocks for the widest BN
// of the path (4096 channels) against >= 2048 on an MI355X.
template <class K>
static int exchange_capacity(K kernel) {         // resident blocks of `kernel` on the CURRENT device; -1: the query failed
    static int capacity[64];                     // per kernel instantiation AND device (round-3 advice: the cache was per process)
    static bool known[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return -1; }
    if (!known[dev]) {
        int cus = 0, per = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, 256, 0) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        capacity[dev] = cus * per;
        known[dev] = true;
    }
    return capacity[dev];
}
template <class K>
static bool exchange_grid_fits(K kernel, int blocks) { return blocks <= exchange_capacity(kernel); }

// `pre_parts` > 0: the partial sums exist already -- `workspace` holds pre_parts rows of [2C] fp64 sums followed by pre_parts rows
// of [2C] fp32 min / max, written by the convolution's epilogue (semseg_conv2d_fwd_stats_h2, which also zeroed the bound word)
// -- and only the finish kernel is launched; else the statistics sweep over z runs first.
static int bn_fwd_stats_fused_impl(const float* z, int P, int C, double* stats, float* zmm, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, float momentum, float eps, int relu,
                                   const float* res_absmax, float* mean, float* invstd, float* scale, float* shift,
                                   void* blockbound, void* workspace, size_t workspace_bytes, void* stream, void* peer,
                                   float* absmax_out, int pre_parts = 0) {
    semseg_peer::PeerArgs pa = {};
    if (peer && (!semseg_peer::peer_args(peer, &pa) || 2 * C + 1 > pa.cap)) return SEMSEG_EINVAL;
    // BEFORE anything is launched: a rank that cannot co-schedule the exchanging grid leaves with nothing in flight (and the ranks
    // have agreed on semseg_bn_peer_channel_capacity() when the exchange was built, comm.peer_init -- this is the backstop)
    if (peer && !exchange_grid_fits(bn_fwd_finish_fused_kernel_peer, ceil_div(C, 16))) return SEMSEG_EINVAL;
    if ((!z && pre_parts <= 0) || !stats || !zmm || !gamma || !beta || !mean || !invstd || !scale || !shift || !blockbound ||
        P <= 0 || C <= 0 || (C % 4) || (z && !aligned16(z)))
        return SEMSEG_EINVAL;
    ColGeom g = col_geom(P, C);
    if (pre_parts > 0) g.gy = pre_parts;
    const size_t need = (size_t)g.gy * 2 * C * (sizeof(double) + sizeof(float));
    if (!workspace || workspace_bytes < need) return SEMSEG_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double* partial = (double*)workspace;
    float* mm = reinterpret_cast<float*>(partial + (size_t)g.gy * 2 * C);
    if (pre_parts <= 0) {
        const size_t smem = (size_t)g.py * g.cx * 8 * (sizeof(double) + sizeof(float));
        SEMSEG_LAUNCH_BODY((bn_stats_mm_partial_kernel_body), dim3(g.gx, g.gy), smem, st, z, P, C, g.cx, g.py,
                           g.rows_per_block, partial, mm, absmax_out);
        SEMSEG_LAUNCH_CHECK();
    }
    if (peer)
        hipLaunchKernelGGL(bn_fwd_finish_fused_kernel_peer, dim3(ceil_div(C, 16)), dim3(256), 0, st, (const double*)partial,
                           (const float*)mm, g.gy, C, (double)P, gamma, beta, running_mean, running_var, momentum, eps, relu,
                           res_absmax, stats, zmm, mean, invstd, scale, shift, num_batches_tracked, (uint32_t*)blockbound,
                           absmax_out, pa);
    else
        SEMSEG_LAUNCH_BODY((bn_fwd_finish_fused_kernel_body<false>), dim3(ceil_div(C, 16)), 0, st, (const double*)partial,
                           (const float*)mm, g.gy, C, (double)P, gamma, beta, running_mean, running_var, momentum, eps, relu,
                           res_absmax, stats, zmm, mean, invstd, scale, shift, num_batches_tracked, (uint32_t*)blockbound,
                           absmax_out, pa);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

extern "C" int semseg_bn_fwd_stats_fused(const float* z, int P, int C, double* stats, float* zmm, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var,
                                         int64_t* num_batches_tracked, float momentum, float eps, int relu,
                                         const float* res_absmax, float* mean, float* invstd, float* scale, float* shift,
                                         void* blockbound, void* workspace, size_t workspace_bytes, void* stream) {
    return bn_fwd_stats_fused_impl(z, P, C, stats, zmm, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                                   relu, res_absmax, mean, invstd, scale, shift, blockbound, workspace, workspace_bytes, stream,
                                   nullptr, nullptr);
}

// + the bound of |y| itself in absmax_out (for outputs that are not written as planes: semseg_bn_apply follows, not _apply_h2)
extern "C" int semseg_bn_fwd_stats_fused_bound(const float* z, int P, int C, double* stats, float* zmm, const float* gamma,
                                               const float* beta, float* running_mean, float* running_var,
                                               int64_t* num_batches_tracked, float momentum, float eps, int relu,
                                               const float* res_absmax, float* mean, float* invstd, float* scale, float* shift,
                                               void* blockbound, void* workspace, size_t workspace_bytes, void* stream,
                                               float* absmax_out) {
    if (!absmax_out) return SEMSEG_EINVAL;
    return bn_fwd_stats_fused_impl(z, P, C, stats, zmm, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                                   relu, res_absmax, mean, invstd, scale, shift, blockbound, workspace, workspace_bytes, stream,
                                   nullptr, absmax_out);
}

extern "C" int semseg_bn_fwd_stats_fused_peer(const float* z, int P, int C, double* stats, float* zmm, const float* gamma,
                                              const float* beta, float* running_mean, float* running_var,
                                              int64_t* num_batches_tracked, float momentum, float eps, int relu,
                                              const float* res_absmax, float* mean, float* invstd, float* scale, float* shift,
                                              void* blockbound, void* workspace, size_t workspace_bytes, void* stream,
                                              void* peer, float* absmax_out) {
    if (!peer) return SEMSEG_EINVAL;
    return bn_fwd_stats_fused_impl(z, P, C, stats, zmm, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                                   relu, res_absmax, mean, invstd, scale, shift, blockbound, workspace, workspace_bytes, stream,
                                   peer, absmax_out);
}

// the finish half alone, on partial sums that the convolution's epilogue left in `partials` (semseg_conv2d_fwd_stats_h2: `parts`
// rows); `peer` / `absmax_out` optional as in the entry points above
extern "C" int semseg_bn_fwd_finish_fused(const void* partials, size_t partials_bytes, int parts, int P, int C, double* stats,
                                          float* zmm, const float* gamma, const float* beta, float* running_mean,
                                          float* running_var, int64_t* num_batches_tracked, float momentum, float eps, int relu,
                                          const float* res_absmax, float* mean, float* invstd, float* scale, float* shift,
                                          void* blockbound, void* stream, void* peer, float* absmax_out) {
    if (parts <= 0 || !partials) return SEMSEG_EINVAL;
    return bn_fwd_stats_fused_impl(nullptr, P, C, stats, zmm, gamma, beta, running_mean, running_var, num_batches_tracked, momentum,
                                   eps, relu, res_absmax, mean, invstd, scale, shift, blockbound, const_cast<void*>(partials),
                                   partials_bytes, stream, peer, absmax_out, parts);
}

// PEER: as bn_fwd_finish_fused_kernel -- [sum g, sum g xhat] summed over the ranks inside the kernel; dgamma / dbeta stay the
// rank's own sums (parameter gradients are reduced with the gradient buckets), max|g| stays rank-local.
template <bool PEER>
struct bn_bwd_finish_fused_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const double* __restrict__ partial, const float* __restrict__ gm, int nparts, int C, const double* __restrict__ count,
    const float* __restrict__ zmm, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, int training, double* __restrict__ sums, float* __restrict__ dgamma,
    float* __restrict__ dbeta, uint32_t* __restrict__ blockbound, semseg_peer::PeerArgs pa) {
    __shared__ double rs[16][17], rq[16][17];
    __shared__ uint32_t rg[16][17];
    __shared__ double xs[2][16];
    __shared__ unsigned s_q;
    const int cl = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int C2 = 2 * C;
    double su = 0.0, sq = 0.0;
    uint32_t gmx = 0;
    if (PEER && threadIdx.x == 0) s_q = *pa.seq;
    // per-channel terms of the finishing thread requested before the partial rows (as bn_fwd_finish_fused_kernel)
    const bool fin = lane == 0 && c < C;
    float p_is = 0.f, p_gamma = 0.f, p_zlo = 0.f, p_zhi = 0.f, p_mean = 0.f;
    double p_count = 1.0;
    if (fin) {
        p_is = invstd[c]; p_gamma = gamma[c];
        if (training) { p_count = count[0]; p_zlo = zmm[c]; p_zhi = zmm[C + c]; p_mean = mean[c]; }
    }
    if (c < C) {
        constexpr int U = 8;                   // as bn_fwd_finish_fused_kernel
        for (int b = lane; b < nparts; b += U * 16) {
            double ps[U], pq[U];
            float pg[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t r = (size_t)min(b + 16 * u, nparts - 1);
                ps[u] = partial[r * C2 + c]; pq[u] = partial[r * C2 + C + c];
                pg[u] = gm[r * C + c];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = b + 16 * u < nparts;
                su = live ? su + ps[u] : su; sq = live ? sq + pq[u] : sq;
                gmx = live ? max(gmx, absbits(pg[u])) : gmx;
            }
        }
    }
    rs[lane][cl] = su; rq[lane][cl] = sq; rg[lane][cl] = gmx;
    __syncthreads();
    uint32_t bits = 0;
    if (lane == 0 && c < C) {
        su = 0.0; sq = 0.0; gmx = 0;
        for (int i = 0; i < 16; ++i) {
            su += rs[i][cl]; sq += rq[i][cl];
            gmx = max(gmx, rg[i][cl]);
        }
        if (dbeta) dbeta[c] = (float)su;
        if (dgamma) dgamma[c] = (float)sq;
        if (PEER) { xs[0][cl] = su; xs[1][cl] = sq; }
    }
    if (PEER) {
        __syncthreads();
        const unsigned q = s_q;
        if (lane < 2 && c < C) {
            const int idx = lane * C + c;
            const double own = xs[lane][cl];
            semseg_peer::push(pa, q, idx, own);
            bool timed_out = false;
            const double tot = semseg_peer::gather_sum(pa, q, idx, own, wall_clock64(), timed_out);
            if (timed_out) atomicOr(pa.status, 1u);
            xs[lane][cl] = tot;
        }
        __syncthreads();
    }
    if (lane == 0 && c < C) {
        if (PEER) { su = xs[0][cl]; sq = xs[1][cl]; }
        sums[c] = su; sums[C + c] = sq;
        const float is = p_is;
        float b = __uint_as_float(gmx);
        if (training) {
            const float inv_n = (float)(1.0 / p_count);
            const float m = fabsf((float)su * inv_n), x = fabsf((float)sq * inv_n);
            const float xh = fmaxf(fabsf(p_zlo - p_mean), fabsf(p_zhi - p_mean)) * is;
            b = b + m + xh * x;
        }
        b = fabsf(p_gamma) * is * b * 1.0009765625f;
        bits = absbits(b);
    }
    bits = block_max_u32(bits);
    if (threadIdx.x == 0) {
        blockbound[blockIdx.x] = bits;
        if (PEER) semseg_peer::advance(pa, s_q, gridDim.x);
    }
    }
};
// the PEER form as a kernel of its own: it waits for the other ranks inside the kernel, so it is never recorded, and the
// co-residency query (exchange_capacity) needs its symbol
__global__ __launch_bounds__(256) void bn_bwd_finish_fused_kernel_peer(const double* __restrict__ partial, const float* __restrict__ gm, int nparts, int C, const double* __restrict__ count, const float* __restrict__ zmm, const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma, int training, double* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, uint32_t* __restrict__ blockbound, semseg_peer::PeerArgs pa) {
    bn_bwd_finish_fused_kernel_body<true>::run(semseg_batch::U3{blockIdx.x, blockIdx.y, blockIdx.z}, semseg_batch::U3{gridDim.x, gridDim.y, gridDim.z}, partial, gm, nparts, C, count, zmm, mean, invstd, gamma, training, sums, dgamma, dbeta, blockbound, pa);
}

static int bn_bwd_reduce_fused_impl(const float* dy, int dy_ld, const float* y, int y_ld, const float* z,
                                    const float* mean, const float* invstd, const float* gate_scale,
                                    const float* gate_shift, int relu, int P, int C, const double* stats_count,
                                    const float* zmm, const float* gamma, int training, double* sums, float* dgamma,
                                    float* dbeta, void* blockbound, void* workspace, size_t workspace_bytes,
                                    void* stream, void* peer, const float* dy2 = nullptr, int dy2_ld = 0) {
    if (dy2 && ((dy2_ld % 4) || dy2_ld < C || !aligned16(dy2))) return SEMSEG_EINVAL;
    semseg_peer::PeerArgs pa = {};
    if (peer && (!semseg_peer::peer_args(peer, &pa) || 2 * C > pa.cap)) return SEMSEG_EINVAL;
    if (peer && !exchange_grid_fits(bn_bwd_finish_fused_kernel_peer, ceil_div(C, 16))) return SEMSEG_EINVAL;      // before any launch
    if (!dy || !z || !mean || !invstd || !sums || !gamma || !blockbound || P <= 0 || C <= 0 || (C % 4) || (dy_ld % 4) ||
        dy_ld < C)
        return SEMSEG_EINVAL;
    if (training && (!stats_count || !zmm)) return SEMSEG_EINVAL;
    if (relu && gate_scale && !gate_shift) return SEMSEG_EINVAL;
    if (relu && !gate_scale && (!y || (y_ld != 0 && ((y_ld % 4) || y_ld < C)) || (y_ld == 0 && (C % 8)))) return SEMSEG_EINVAL;
    const ColGeom g = col_geom(P, C);
    const size_t need = (size_t)g.gy * 2 * C * (sizeof(double) + sizeof(float));
    if (!workspace || workspace_bytes < need) return SEMSEG_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double* partial = (double*)workspace;
    float* gm = reinterpret_cast<float*>(partial + (size_t)g.gy * 2 * C);
    const size_t smem = (size_t)g.py * g.cx * (8 * sizeof(double) + 4 * sizeof(float));
#define LAUNCH_PARTIAL(GATE)                                                                                              \
    SEMSEG_LAUNCH_BODY((bn_bwd_mm_partial_kernel_body<GATE>), dim3(g.gx, g.gy), smem, st, dy, dy_ld, y, y_ld, z, mean, \
                       invstd, relu, P, C, g.cx, g.py, g.rows_per_block, partial, gm, gate_scale, gate_shift, dy2, dy2_ld)
    if (!relu) LAUNCH_PARTIAL(0);
    else if (gate_scale) LAUNCH_PARTIAL(1);
    else if (y_ld == 0) LAUNCH_PARTIAL(3);          // `y` is the forward's ReLU bitmask
    else LAUNCH_PARTIAL(2);
#undef LAUNCH_PARTIAL
    SEMSEG_LAUNCH_CHECK();
    if (peer)
        hipLaunchKernelGGL(bn_bwd_finish_fused_kernel_peer, dim3(ceil_div(C, 16)), dim3(256), 0, st, (const double*)partial,
                           (const float*)gm, g.gy, C, stats_count, zmm, mean, invstd, gamma, training, sums, dgamma, dbeta,
                           (uint32_t*)blockbound, pa);
    else
        SEMSEG_LAUNCH_BODY((bn_bwd_finish_fused_kernel_body<false>), dim3(ceil_div(C, 16)), 0, st, (const double*)partial,
                           (const float*)gm, g.gy, C, stats_count, zmm, mean, invstd, gamma, training, sums, dgamma, dbeta,
                           (uint32_t*)blockbound, pa);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

extern "C" int semseg_bn_bwd_reduce_fused(const float* dy, int dy_ld, const float* y, int y_ld, const float* z,
                                          const float* mean, const float* invstd, const float* gate_scale,
                                          const float* gate_shift, int relu, int P, int C, const double* stats_count,
                                          const float* zmm, const float* gamma, int training, double* sums, float* dgamma,
                                          float* dbeta, void* blockbound, void* workspace, size_t workspace_bytes,
                                          void* stream) {
    return bn_bwd_reduce_fused_impl(dy, dy_ld, y, y_ld, z, mean, invstd, gate_scale, gate_shift, relu, P, C, stats_count, zmm, gamma,
                                    training, sums, dgamma, dbeta, blockbound, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int semseg_bn_bwd_reduce_fused_peer(const float* dy, int dy_ld, const float* y, int y_ld, const float* z,
                                               const float* mean, const float* invstd, const float* gate_scale,
                                               const float* gate_shift, int relu, int P, int C, const double* stats_count,
                                               const float* zmm, const float* gamma, int training, double* sums, float* dgamma,
                                               float* dbeta, void* blockbound, void* workspace, size_t workspace_bytes,
                                               void* stream, void* peer) {
    if (!peer) return SEMSEG_EINVAL;
    return bn_bwd_reduce_fused_impl(dy, dy_ld, y, y_ld, z, mean, invstd, gate_scale, gate_shift, relu, P, C, stats_count, zmm, gamma,
                                    training, sums, dgamma, dbeta, blockbound, workspace, workspace_bytes, stream, peer);
}

// the same with the incoming gradient given as TWO addends (dy + dy2: the two gradients that meet at a fork of the autograd graph,
// ops.ForkFn -- e.g. a block output that feeds the next block's first conv and its shortcut, resnet.py:84-92): the sum is formed
// where it is consumed instead of by an add launch of its own.  peer may be NULL (single rank).
extern "C" int semseg_bn_bwd_reduce_fused_sum2(const float* dy, int dy_ld, const float* dy2, int dy2_ld, const float* y, int y_ld,
                                               const float* z, const float* mean, const float* invstd, const float* gate_scale,
                                               const float* gate_shift, int relu, int P, int C, const double* stats_count,
                                               const float* zmm, const float* gamma, int training, double* sums, float* dgamma,
                                               float* dbeta, void* blockbound, void* workspace, size_t workspace_bytes,
                                               void* stream, void* peer) {
    if (!dy2) return SEMSEG_EINVAL;
    return bn_bwd_reduce_fused_impl(dy, dy_ld, y, y_ld, z, mean, invstd, gate_scale, gate_shift, relu, P, C, stats_count, zmm, gamma,
                                    training, sums, dgamma, dbeta, blockbound, workspace, workspace_bytes, stream, peer, dy2, dy2_ld);
}

// widest BN (channels) whose exchanging finish kernels -- ceil(C / 16) blocks that poll each other's peers inside the kernel -- are
// co-resident on the CURRENT device; 0 when the occupancy query fails.  comm.peer_init gathers this from every rank when the
// exchange is built and keeps the peer exchange only if every rank can hold the widest payload it was created for (round-3 advice:
// a rank that failed the check inside a step left its peers spinning until the timeout).
extern "C" int semseg_bn_peer_channel_capacity(void) {
    const int a = exchange_capacity(bn_fwd_finish_fused_kernel_peer), b = exchange_capacity(bn_bwd_finish_fused_kernel_peer);
    const int blocks = a < b ? a : b;
    return blocks > 0 ? (blocks > (1 << 26) ? (1 << 30) : blocks * 16) : 0;
}
