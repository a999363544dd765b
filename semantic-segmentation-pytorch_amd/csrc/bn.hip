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
    // ~512 blocks in total (2 per CU); every thread walks >= 8 rows
    int gy = ceil_div(512, g.gx);
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
    if (j < C2)
        for (int b = lane; b < nparts; b += 16) s += partial[(size_t)b * C2 + j];
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
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ scale,
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
#define LAUNCH(R, A) hipLaunchKernelGGL((bn_apply_kernel<R, A>), dim3(blocks), dim3(256), 0, st, z, scale, shift, residual, res_ld, y, y_ld, P, C)
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
