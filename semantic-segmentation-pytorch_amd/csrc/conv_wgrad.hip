// Weight gradient of the fp32 convolution on the f32-input MFMA (gfx950).
//   dw[k][t][c] = sum_m dy[m][k] * x[pix(m,t)][c]        (m over N*OH*OW output pixels)
// i.e. for every tap t a GEMM  (K x M) * (M x C)  whose reduction index (pixels) is the slow axis of
// both NHWC operands.  Tiles are staged pixel-major in LDS ([pixel][channel], channel contiguous, the
// way they sit in HBM: coalesced float4 rows) and the MFMA fragments are read with conflict-free
// ds_read_b32 (32 consecutive channels per half-wave).  The pixel range is split over grid.y; partial
// slabs are summed in a fixed order by wgrad_reduce_kernel (deterministic, no atomics).
// Replaces the autograd weight-gradient of nn.Conv2d (see conv_igemm.hip for call sites).
#include "common.h"
#include <stdlib.h>

struct WgradParams {
    const float* x;
    const float* dy;
    float* dw;        // [K][T][C] final (splits == 1)
    float* partial;   // [splits][K*T*C]
    int x_ld, dy_ld;
    int H, W, C;
    int OH, OW, K;
    int M;
    int S, T;
    int stride, pad, dil;
    int tiles_k, tiles_c;
    int m_per_split, splits;
};

template <int BM, int BN, int BKP, bool VEC>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int QA = BM / 4, QB = BN / 4;
    constexpr int RPA = 256 / QA, RPB = 256 / QB;
    constexpr int APASS = BKP / RPA, BPASS = BKP / RPB;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 32, FN = WN / 32;
    static_assert(APASS >= 1 && BPASS >= 1, "tile");

    extern __shared__ __align__(16) float smem[];
    float* As = smem;                     // [2][BKP][LDA]  dy tile, pixel-major
    float* Bs = smem + 2 * BKP * LDA;     // [2][BKP][LDB]  gathered x tile, pixel-major

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // tile id -> (tap, k tile, c tile); taps vary fastest among blocks of one XCD so that the 9 taps of a
    // (k,c) tile re-use the same dy rows / overlapping x rows from that XCD's L2
    const int ntiles = p.tiles_k * p.tiles_c * p.T;
    const WgradBlock tb = wgrad_block(blockIdx.x + gridDim.x * blockIdx.y, ntiles, gridDim.y);     // (tile, split) together: block_order.h
    const int tile = tb.tile;
    const int t = tile % p.T;
    const int tc = (tile / p.T) % p.tiles_c;
    const int tk = tile / (p.T * p.tiles_c);
    const int k0 = tk * BM, c0 = tc * BN;
    const int r = t / p.S, s = t - r * p.S;
    const int z = tb.z;
    const int m_begin = z * p.m_per_split;
    const int m_end = min(p.M, m_begin + p.m_per_split);

    const int qa = tid % QA, rowa = tid / QA;
    const int qb = tid % QB, rowb = tid / QB;
    const int ka = k0 + 4 * qa;
    const int na_valid = max(0, min(4, p.K - ka));
    const int cb = c0 + 4 * qb;
    const int nb_valid = max(0, min(4, p.C - cb));
    const int HWo = p.OH * p.OW;

    float4 ra[APASS], rb[BPASS];
    // branch-free loads (see conv_igemm.hip): unconditional load from a clamped offset, zero fill by select
    const float inv_hwo = 1.0f / (float)HWo, inv_ow = 1.0f / (float)p.OW;
    auto load_tile = [&](int mt) {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                const int m = mt + rowa + i * RPA;
                const bool ok = (m < m_end) && (na_valid > 0);
                const float4 v = *reinterpret_cast<const float4*>(p.dy + (ok ? (uint32_t)m * (uint32_t)p.dy_ld + ka : 0u));
                ra[i] = ok ? v : f4zero();
            }
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                const int m = min(mt + rowb + i * RPB, p.M - 1);
                // m -> (n, oh, ow) with float-reciprocal division + one correction step (exact for m < 2^24)
                int n = (int)((float)m * inv_hwo);
                int rem = m - n * HWo;
                if (rem < 0) { --n; rem += HWo; } else if (rem >= HWo) { ++n; rem -= HWo; }
                int oh = (int)((float)rem * inv_ow);
                int ow = rem - oh * p.OW;
                if (ow < 0) { --oh; ow += p.OW; } else if (ow >= p.OW) { ++oh; ow -= p.OW; }
                const int ih = oh * p.stride - p.pad + r * p.dil;
                const int iw = ow * p.stride - p.pad + s * p.dil;
                const bool ok = (mt + rowb + i * RPB < m_end) & (ih >= 0) & (iw >= 0) & (ih < p.H) & (iw < p.W) & (nb_valid > 0);
                const uint32_t addr = (uint32_t)((n * p.H + ih) * p.W + iw) * (uint32_t)p.x_ld + cb;
                const float4 v = *reinterpret_cast<const float4*>(p.x + (ok ? addr : 0u));
                rb[i] = ok ? v : f4zero();
            }
        } else {
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                const int m = mt + rowa + i * RPA;
                const bool ok = m < m_end;
                ra[i] = load4<false>(p.dy + (ok ? (size_t)m * p.dy_ld + ka : 0), ok ? na_valid : 0);
            }
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                const int m = mt + rowb + i * RPB;
                bool ok = m < m_end;
                size_t addr = 0;
                if (ok) {
                    const int n = m / HWo;
                    const int rem = m - n * HWo;
                    const int oh = rem / p.OW;
                    const int ow = rem - oh * p.OW;
                    const int ih = oh * p.stride - p.pad + r * p.dil;
                    const int iw = ow * p.stride - p.pad + s * p.dil;
                    ok = (ih >= 0) & (iw >= 0) & (ih < p.H) & (iw < p.W);
                    addr = ((size_t)(n * p.H + ih) * p.W + iw) * p.x_ld + cb;
                }
                rb[i] = load4<false>(p.x + (ok ? addr : 0), ok ? nb_valid : 0);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < APASS; ++i)
            *reinterpret_cast<float4*>(&As[(buf * BKP + rowa + i * RPA) * LDA + 4 * qa]) = ra[i];
#pragma unroll
        for (int i = 0; i < BPASS; ++i)
            *reinterpret_cast<float4*>(&Bs[(buf * BKP + rowb + i * RPB) * LDB + 4 * qb]) = rb[i];
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (m_begin < m_end) {
        load_tile(m_begin);
        store_tile(0);
    }
    __syncthreads();

    const int fcol = lane & 31;
    const int fk = lane >> 5;
    int it = 0;
    for (int mt = m_begin; mt < m_end; mt += BKP, ++it) {
        const int cur = it & 1;
        const bool more = (mt + BKP) < m_end;
        if (more) load_tile(mt + BKP);
        const float* Ab = As + (cur * BKP + fk) * LDA + wm * WM + fcol;
        const float* Bb = Bs + (cur * BKP + fk) * LDB + wn * WN + fcol;
#pragma unroll
        for (int ks = 0; ks < BKP / 2; ++ks) {
            float av[FM], bv[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) av[i] = Ab[ks * 2 * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < FN; ++j) bv[j] = Bb[ks * 2 * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    float* dst = (p.splits == 1) ? p.dw : p.partial + (size_t)z * p.K * p.T * p.C;
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int c = c0 + wn * WN + j * 32 + col_l;
        if (c >= p.C) continue;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + row_l;
                if (k < p.K) dst[((size_t)k * p.T + t) * p.C + c] = acc[i][j][e];
            }
        }
    }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, size_t total, int splits) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float s = partial[i];
        for (int z = 1; z < splits; ++z) s += partial[(size_t)z * total + i];
        dw[i] = s;
    }
}

// db[k] = sum_m dy[m][k] : one block per 16 channels, 16 row lanes, fp64 accumulation across rows
__global__ void bias_grad_kernel(const float* __restrict__ dy, int dy_ld, float* __restrict__ db, int M, int K) {
    __shared__ double red[16][17];
    const int kx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int k = blockIdx.x * 16 + kx;
    double s = 0.0;
    if (k < K)
        for (int m = ry; m < M; m += 16) s += (double)dy[(size_t)m * dy_ld + k];
    red[ry][kx] = s;
    __syncthreads();
    if (ry == 0 && k < K) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i][kx];
        db[k] = (float)t;
    }
}

struct WgradPlan {
    int BM, BN, tiles_k, tiles_c, splits, m_per_split;
};

// Same wave-quantisation model as plan_igemm: 128x128 tiles (2 blocks per CU -> 512 slots) or 64x64 (4 per CU),
// the pixel range split so that the grid is a whole number of waves and every block still loops >= 8 pixel tiles.
static WgradPlan plan_wgrad(int M, int K, int C, int T) {
    WgradPlan pl;
    const int mtiles = ceil_div(M, 32);
    const int cand[2] = {128, 64};
    const double tile_cost[2] = {1.0, 0.59};
    const int slots[2] = {512, 1024};
    int force_tile = -1, force_split = 0;
    {   // tuning overrides: SEMSEG_WGRAD_TILE=0 (128x128) | 1 (64x64), SEMSEG_WGRAD_SPLIT=n
        const char* v = getenv("SEMSEG_WGRAD_TILE");
        if (v && *v) force_tile = atoi(v);
        v = getenv("SEMSEG_WGRAD_SPLIT");
        if (v && *v) force_split = atoi(v);
    }
    double best = 1e30;
    int best_t = 1, best_s = 1;
    for (int t = 0; t < 2; ++t) {
        if (force_tile >= 0 && t != force_tile) continue;
        if (t == 0 && (K < 128 || C < 128) && force_tile < 0) continue;
        const long tiles = (long)ceil_div(K, cand[t]) * ceil_div(C, cand[t]) * T;
        for (int sp = 1; sp <= 64; ++sp) {
            if (force_split > 0 && sp != min(force_split, mtiles)) continue;
            if (force_split <= 0 && sp > 1 && mtiles / sp < 8) break;
            const int mps = ceil_div(mtiles, sp);
            if (ceil_div(mtiles, mps) != sp) continue;
            const long blocks = tiles * sp;
            const long rem = blocks % slots[t];
            double pw = 0.0;
            if (rem) {
                const int b = (int)((rem + 255) / 256), cs = slots[t] / 256;
                const double g4[4] = {0.0, 0.40, 0.65, 0.85};
                pw = b >= cs ? 1.0 : (cs == 2 ? 0.567 : g4[b]);
            }
            const double waves = (double)(blocks / slots[t]) + pw;
            double time = waves * (mps * tile_cost[t] + 4.0 * tile_cost[t]);
            if (sp > 1) time += (double)(sp + 1) * K * T * C * 4.0 / 5.0e12 / 4.2e-6;
            if (time < best) { best = time; best_t = t; best_s = sp; }
        }
    }
    pl.BM = pl.BN = cand[best_t];
    pl.tiles_k = ceil_div(K, pl.BM);
    pl.tiles_c = ceil_div(C, pl.BN);
    pl.m_per_split = ceil_div(mtiles, best_s) * 32;
    pl.splits = ceil_div(M, pl.m_per_split);
    return pl;
}

size_t wgrad_workspace_bytes(int M, int K, int C, int T) {
    const WgradPlan pl = plan_wgrad(M, K, C, T);
    return pl.splits > 1 ? (size_t)pl.splits * K * T * C * sizeof(float) : 0;
}

template <int BM, int BN, bool VEC>
static int launch_wgrad(const WgradParams& p, hipStream_t st) {
    constexpr int BKP = 32;
    constexpr size_t smem = (size_t)2 * BKP * (BM + 4 + BN + 4) * sizeof(float);
    static SmemAttrCache attr_cache;
    if (int e = ensure_smem_attr(attr_cache, (const void*)wgrad_kernel<BM, BN, BKP, VEC>, smem)) return e;
    dim3 grid(p.tiles_k * p.tiles_c * p.T, p.splits);
    hipLaunchKernelGGL((wgrad_kernel<BM, BN, BKP, VEC>), grid, dim3(256), smem, st, p);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

static inline int out_dim(int in, int k, int stride, int pad, int dil) {
    return (in + 2 * pad - dil * (k - 1) - 1) / stride + 1;
}

extern "C" int semseg_conv2d_wgrad(const float* x, int x_ld, const float* dy, int dy_ld, float* dw, float* db,
                                   int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dw || N <= 0 || C <= 0 || K <= 0 || stride <= 0 || dil <= 0 || x_ld < C || dy_ld < K) return SEMSEG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int OH = out_dim(H, R, stride, pad, dil), OW = out_dim(W, S, stride, pad, dil);
    if (OH <= 0 || OW <= 0) return SEMSEG_EINVAL;
    WgradParams p = {};
    p.x = x; p.dy = dy; p.dw = dw;
    p.x_ld = x_ld; p.dy_ld = dy_ld;
    p.H = H; p.W = W; p.C = C; p.OH = OH; p.OW = OW; p.K = K;
    p.M = N * OH * OW;
    p.S = S; p.T = R * S;
    p.stride = stride; p.pad = pad; p.dil = dil;
    // 32-bit element offsets + float-reciprocal pixel decomposition in the kernel
    if ((size_t)N * H * W * x_ld >= ((size_t)1 << 31) || (size_t)p.M * dy_ld >= ((size_t)1 << 31) || p.M >= (1 << 24))
        return SEMSEG_EINVAL;
    const WgradPlan pl = plan_wgrad(p.M, K, C, p.T);
    p.tiles_k = pl.tiles_k; p.tiles_c = pl.tiles_c;
    p.m_per_split = pl.m_per_split; p.splits = pl.splits;
    if (pl.splits > 1) {
        const size_t need = (size_t)pl.splits * K * p.T * C * sizeof(float);
        if (!workspace || workspace_bytes < need) return SEMSEG_EWORKSPACE;
        p.partial = (float*)workspace;
    }
    const bool vec = (C % 4 == 0) && (K % 4 == 0) && (x_ld % 4 == 0) && (dy_ld % 4 == 0) && aligned16(x) && aligned16(dy);
    int rc;
    if (pl.BM == 128) rc = vec ? launch_wgrad<128, 128, true>(p, st) : launch_wgrad<128, 128, false>(p, st);
    else rc = vec ? launch_wgrad<64, 64, true>(p, st) : launch_wgrad<64, 64, false>(p, st);
    if (rc) return rc;
    if (pl.splits > 1) {
        const size_t total = (size_t)K * p.T * C;
        const int blocks = (int)min((size_t)2048, ceil_div_sz(total, 256));
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.partial, dw, total, pl.splits);
        SEMSEG_LAUNCH_CHECK();
    }
    if (db) {
        hipLaunchKernelGGL(bias_grad_kernel, dim3(ceil_div(K, 16)), dim3(256), 0, st, dy, dy_ld, db, p.M, K);
        SEMSEG_LAUNCH_CHECK();
    }
    return 0;
}
