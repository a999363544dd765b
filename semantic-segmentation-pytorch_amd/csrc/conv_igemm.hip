// Implicit-GEMM (im2col-free) fp32 convolution for gfx950 on the f32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chain, 157 TFLOP/s peak).
//
// One kernel serves conv forward AND data-gradient: both are
//     out[m, n] = sum_{tap t, channel c}  in[pix(m, t), c] * B[n][t][c]
// with m an output pixel, n an output channel, and `pix` the tap geometry
//     num_h = oh*a + off + r*step ;  valid iff num_h >= 0, num_h % div == 0, num_h/div < Hin
//   forward: a=stride, off=-pad, step=+dil, div=1     (B = KRSC weights)
//   dgrad  : a=1,      off=+pad, step=-dil, div=stride (B = CRSK transposed weights, in = dy)
// Replaces nn.Conv2d at reference resnet.py:18-21,61-66,130; models.py:163,406,448,456-463,519-540;
// hrnet.py:26-29,188-205,316-338 and its autograd dgrad.
//
// Tiling: 256 threads = 4 waves (2x2), block tile BM x BN x BK, wave tile (BM/2)x(BN/2) built from
// 32x32 MFMA fragments.  A (gathered activations) and B (weights) are both K-contiguous, so a lane's
// fragment values for 4 consecutive k-steps are one ds_read_b128 from a [rows][BK+4] LDS image (row
// stride 36/20 floats: conflict-free for the b128 lane groups).  Global->register->LDS double
// buffering, one barrier per K tile.  Optional split-K (grid.y) writes fp32 partial slabs that
// splitk_reduce_kernel sums in a fixed order (deterministic).
#include "common.h"
#include <stdlib.h>

struct IGemmParams {
    const float* in;
    const float* wgt;
    const float* bias;
    float* out;        // direct output (splits == 1)
    float* partial;    // split-K slabs [splits][M][Cout]
    int in_ld, out_ld;
    int Hin, Win, Cin;
    int Hout, Wout, Cout;
    int M;
    int S, T;
    int a, off, step, div;
    int chunks;        // ceil(Cin / BK)
    int ktiles;        // T * chunks
    int kt_per_split;
    int tiles_m, tiles_n;
    int splits;
};

template <int BM, int BN, int BK, bool VEC>
__global__ __launch_bounds__(256) void igemm_conv_kernel(const IGemmParams p) {
    constexpr int LDK = BK + 4;
    constexpr int QK = BK / 4;            // float4 per tile row
    constexpr int RPP = 256 / QK;         // rows per load pass
    constexpr int APASS = BM / RPP;
    constexpr int BPASS = BN / RPP;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 32, FN = WN / 32;
    static_assert(APASS >= 1 && BPASS >= 1 && FM >= 1 && FN >= 1, "tile too small");

    extern __shared__ __align__(16) float smem[];
    float* As = smem;                      // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;       // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // the remap runs over (tile, split) together (block_order.h): a remap of x alone is an XCD map only when gridDim.x % 8 == 0
    const int ntiles = p.tiles_m * p.tiles_n;
    const WgradBlock tb = wgrad_block(blockIdx.x + gridDim.x * blockIdx.y, ntiles, gridDim.y);
    const int tile = tb.tile;
    const int tn = tile % p.tiles_n;
    const int tm = tile / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = tb.z;
    const int kt_begin = z * p.kt_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.kt_per_split);

    const int q = tid % QK;
    const int lrow = tid / QK;

    // per-thread gather rows (fixed for the whole K loop)
    int a_ih0[APASS], a_iw0[APASS], a_base[APASS];
    const int HWout = p.Hout * p.Wout;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int m = m0 + lrow + i * RPP;
        if (m < p.M) {
            const int n = m / HWout;
            const int rem = m - n * HWout;
            const int oh = rem / p.Wout;
            const int ow = rem - oh * p.Wout;
            a_ih0[i] = oh * p.a + p.off;
            a_iw0[i] = ow * p.a + p.off;
            a_base[i] = n * p.Hin * p.Win;
        } else {
            a_ih0[i] = -(1 << 28);
            a_iw0[i] = -(1 << 28);
            a_base[i] = 0;
        }
    }

    float4 ra[APASS], rb[BPASS];
    const uint32_t w_row = (uint32_t)p.T * p.Cin;

    // Branch-free tile loader.  hipcc turns a per-element `cond ? load : 0` into one basic block per load with its
    // own s_waitcnt (serialised L2 round trips, ~350 scalar/vector instructions ahead of the MFMA block), so every
    // load is issued UNCONDITIONALLY from a clamped in-bounds offset and the zero fill is a v_cndmask afterwards.
    // Offsets are 32-bit element offsets from the tensor base (host checks the tensors are < 2^31 elements); the
    // gather offsets only change when the tap changes, inside a tap the k-loop adds the channel offset.
    uint32_t a_off[APASS], b_off[BPASS];
    uint32_t a_ok = 0, b_ok = 0;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
        const int n = n0 + lrow + i * RPP;
        const bool ok = n < p.Cout;
        b_off[i] = ok ? (uint32_t)n * w_row + 4u * q : 0u;
        b_ok |= (ok ? 1u : 0u) << i;
    }
    int cur_t = -1;
    auto set_tap = [&](int t) {
        const int r = t / p.S;
        const int s = t - r * p.S;
        a_ok = 0;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            int nh = a_ih0[i] + r * p.step;
            int nw = a_iw0[i] + s * p.step;
            bool ok = (nh >= 0) & (nw >= 0);
            if (p.div > 1) {
                ok = ok & ((nh % p.div) == 0) & ((nw % p.div) == 0);
                nh /= p.div;
                nw /= p.div;
            }
            ok = ok & (nh < p.Hin) & (nw < p.Win);
            a_off[i] = ok ? (uint32_t)(a_base[i] + nh * p.Win + nw) * (uint32_t)p.in_ld + 4u * q : 0u;
            a_ok |= (ok ? 1u : 0u) << i;
        }
    };

    auto load_tile = [&](int kt) {
        const int t = kt / p.chunks;
        const int c0 = (kt - t * p.chunks) * BK;
        if (t != cur_t) {          // wave-uniform
            set_tap(t);
            cur_t = t;
        }
        const uint32_t koff = (uint32_t)t * p.Cin + c0;
        if (VEC) {
            const bool cok = (c0 + 4 * q) < p.Cin;          // only false inside a partial last chunk
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                const bool ok = ((a_ok >> i) & 1u) && cok;
                const float4 v = *reinterpret_cast<const float4*>(p.in + (ok ? a_off[i] + c0 : 0u));
                ra[i] = ok ? v : f4zero();
            }
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                const bool ok = ((b_ok >> i) & 1u) && cok;
                const float4 v = *reinterpret_cast<const float4*>(p.wgt + (ok ? b_off[i] + koff : 0u));
                rb[i] = ok ? v : f4zero();
            }
        } else {
            // scalar path (3-channel stem, 150-class dgrad): tiny layers, conditional loads are fine here
            const int nvalid = max(0, min(4, p.Cin - (c0 + 4 * q)));
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                const bool ok = (a_ok >> i) & 1u;
                ra[i] = load4<false>(p.in + (ok ? a_off[i] + c0 : 0u), ok ? nvalid : 0);
            }
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                const bool ok = (b_ok >> i) & 1u;
                rb[i] = load4<false>(p.wgt + (ok ? b_off[i] + koff : 0u), ok ? nvalid : 0);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < APASS; ++i)
            *reinterpret_cast<float4*>(&As[(buf * BM + lrow + i * RPP) * LDK + 4 * q]) = ra[i];
#pragma unroll
        for (int i = 0; i < BPASS; ++i)
            *reinterpret_cast<float4*>(&Bs[(buf * BN + lrow + i * RPP) * LDK + 4 * q]) = rb[i];
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 31;
    const int fk = (lane >> 5) * 4;

    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0);
    }
    __syncthreads();

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const bool more = (kt + 1) < kt_end;
        if (more) load_tile(kt + 1);

        const float* Ab = As + (cur * BM + wm * WM + frow) * LDK + fk;
        const float* Bb = Bs + (cur * BN + wn * WN + frow) * LDK + fk;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float4 av[FM], bv[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) av[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDK + g * 8);
#pragma unroll
            for (int j = 0; j < FN; ++j) bv[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDK + g * 8);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    float* dst;
    int dst_ld;
    const bool direct = p.splits == 1;
    if (direct) {
        dst = p.out;
        dst_ld = p.out_ld;
    } else {
        dst = p.partial + (size_t)z * p.M * p.Cout;
        dst_ld = p.Cout;
    }
    const int col_l = lane & 31;
    const int row_l = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int col = n0 + wn * WN + j * 32 + col_l;
        if (col >= p.Cout) continue;
        float bv = (direct && p.bias) ? p.bias[col] : 0.f;
        asm volatile("" : "+v"(bv));      // one wait for the conditional load here, not one per stored row (conv_split.hip gemm_epilogue)
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + row_l;
                if (row < p.M) dst[(size_t)row * dst_ld + col] = acc[i][j][e] + bv;
            }
        }
    }
}

// out[m*out_ld + n] = bias[n] + sum_z partial[z][m*Cout + n]   (fixed order => deterministic)
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                     float* __restrict__ out, int out_ld, int M, int Cout, int splits) {
    const size_t total = (size_t)M * Cout;
    const size_t slab = total;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / Cout);
        const int n = (int)(i - (size_t)m * Cout);
        float s = partial[i];
        for (int zz = 1; zz < splits; ++zz) s += partial[(size_t)zz * slab + i];
        if (bias) s += bias[n];
        out[(size_t)m * out_ld + n] = s;
    }
}

// [K][T][C] -> [C][T][K], one 32x32 LDS-transposed tile per block per tap
__global__ void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int K, int T, int C) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int c0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i, c = c0 + tx;
        tile[i][tx] = (k < K && c < C) ? w[((size_t)k * T + t) * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, k = k0 + tx;
        if (c < C && k < K) wt[((size_t)c * T + t) * K + k] = tile[tx][i];
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct IGemmPlan {
    int BM, BN, BK;
    int tiles_m, tiles_n, chunks, ktiles, splits, kt_per_split;
    bool vec;
};

// tuning overrides for tools/conv_bench.py: SEMSEG_IGEMM_TILE=0|1|2 (128x128 / 128x64 / 64x64), SEMSEG_IGEMM_SPLITK=n
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

static inline double partial_wave_cost(int blocks_on_cu, int cu_slots) {
    if (blocks_on_cu >= cu_slots) return 1.0;
    if (cu_slots == 2) return 0.567;
    const double g4[4] = {0.0, 0.40, 0.65, 0.85};
    return g4[blocks_on_cu];
}

// Launch plan.  Measured on MI355X (tools/conv_bench.py --sweep, profiles/conv_bench_sweep_r1.txt): the 128x128
// tile is the most efficient per MFMA (least L2 traffic, most MFMAs per barrier), but only if the grid fills the
// 512 resident-workgroup slots (2 per CU) in whole waves -- a 1152-block grid runs 3 waves for 2.25 waves of work.
// So: enumerate tile x split-K candidates and take the one with the smallest modelled time
//     time = ceil(blocks / slots) * (ktiles_per_block * tile_cost + fixed) + split-K reduction traffic.
static IGemmPlan plan_igemm(int M, int Cout, int Cin, int T, bool vec) {
    IGemmPlan pl;
    pl.vec = vec;
    pl.BK = (Cin % 32 == 0) ? 32 : 16;
    const int force_bk = env_int("SEMSEG_IGEMM_BK", 0);
    if (force_bk == 16 || (force_bk == 32 && Cin % 32 == 0)) pl.BK = force_bk;
    pl.chunks = ceil_div(Cin, pl.BK);
    pl.ktiles = T * pl.chunks;
    const int cand[3][2] = {{128, 128}, {128, 64}, {64, 64}};
    // CU time of one k-tile for a full set of resident blocks (2 / 2 / 4 per CU), relative to 128x128x32 at ~127 TF:
    // fitted to the sweep (conv_last fwd: 127 / 112 / 107 TF for the three tiles at equal wave counts)
    const double tile_cost[3] = {1.0, 0.56, 0.59};
    const int slots[3] = {512, 512, 1024};
    const double kscale = pl.BK / 32.0;
    double best = 1e30;
    int best_t = 2, best_s = 1;
    for (int t = 0; t < 3; ++t) {
        if (Cout <= 64 && cand[t][1] > 64) continue;          // never waste a 128-wide N tile on <=64 channels
        const long tiles = (long)ceil_div(M, cand[t][0]) * ceil_div(Cout, cand[t][1]);
        for (int sp = 1; sp <= 16; ++sp) {
            if (sp > 1 && pl.ktiles / sp < 8) break;
            const int kps = ceil_div(pl.ktiles, sp);
            const int nsp = ceil_div(pl.ktiles, kps);
            if (nsp != sp) continue;
            const long blocks = tiles * sp;
            // whole waves cost 1.  A partial last wave costs as much as its most loaded CU: b = ceil(rem / 256) blocks
            // on a CU that could hold slots/256; measured: one 128x128 block alone on a CU runs 1.76x faster than two.
            const long rem = blocks % slots[t];
            const double waves = (double)(blocks / slots[t]) + (rem ? partial_wave_cost((int)((rem + 255) / 256), slots[t] / 256) : 0.0);
            double time = waves * (kps * tile_cost[t] * kscale + 4.0 * tile_cost[t]);
            if (sp > 1) time += (double)(sp + 1) * M * Cout * 4.0 / 5.0e12 / 4.2e-6;   // slab write + reduce, in tile units (4.2 us)
            if (time < best) { best = time; best_t = t; best_s = sp; }
        }
    }
    const int force_tile = env_int("SEMSEG_IGEMM_TILE", -1);
    if (force_tile >= 0 && force_tile <= 2) { best_t = force_tile; best_s = 1; }
    const int force_split = env_int("SEMSEG_IGEMM_SPLITK", 0);
    if (force_split > 0) best_s = min(force_split, pl.ktiles);
    pl.BM = cand[best_t][0];
    pl.BN = cand[best_t][1];
    pl.tiles_m = ceil_div(M, pl.BM);
    pl.tiles_n = ceil_div(Cout, pl.BN);
    pl.kt_per_split = ceil_div(pl.ktiles, best_s);
    pl.splits = ceil_div(pl.ktiles, pl.kt_per_split);
    return pl;
}

template <int BM, int BN, int BK, bool VEC>
static int launch_igemm(const IGemmParams& p, hipStream_t st) {
    constexpr size_t smem = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float);
    static SmemAttrCache attr_cache;
    if (int e = ensure_smem_attr(attr_cache, (const void*)igemm_conv_kernel<BM, BN, BK, VEC>, smem)) return e;
    dim3 grid(p.tiles_m * p.tiles_n, p.splits);
    hipLaunchKernelGGL((igemm_conv_kernel<BM, BN, BK, VEC>), grid, dim3(256), smem, st, p);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

template <bool VEC>
static int dispatch_igemm(const IGemmPlan& pl, const IGemmParams& p, hipStream_t st) {
    if (pl.BK == 32) {
        if (pl.BM == 128 && pl.BN == 128) return launch_igemm<128, 128, 32, VEC>(p, st);
        if (pl.BM == 128 && pl.BN == 64) return launch_igemm<128, 64, 32, VEC>(p, st);
        return launch_igemm<64, 64, 32, VEC>(p, st);
    }
    if (pl.BM == 128 && pl.BN == 128) return launch_igemm<128, 128, 16, VEC>(p, st);
    if (pl.BM == 128 && pl.BN == 64) return launch_igemm<128, 64, 16, VEC>(p, st);
    return launch_igemm<64, 64, 16, VEC>(p, st);
}

static int run_igemm(IGemmParams p, bool vec, void* workspace, size_t workspace_bytes, hipStream_t st) {
    // the kernel addresses both operands with 32-bit element offsets
    const size_t in_elems = (size_t)(p.M / (p.Hout * p.Wout)) * p.Hin * p.Win * p.in_ld;
    if (in_elems >= ((size_t)1 << 31) || (size_t)p.Cout * p.T * p.Cin >= ((size_t)1 << 31)) return SEMSEG_EINVAL;
    const IGemmPlan pl = plan_igemm(p.M, p.Cout, p.Cin, p.T, vec);
    p.chunks = pl.chunks;
    p.ktiles = pl.ktiles;
    p.kt_per_split = pl.kt_per_split;
    p.tiles_m = pl.tiles_m;
    p.tiles_n = pl.tiles_n;
    p.splits = pl.splits;
    p.partial = nullptr;
    if (pl.splits > 1) {
        const size_t need = (size_t)pl.splits * p.M * p.Cout * sizeof(float);
        if (!workspace || workspace_bytes < need) return SEMSEG_EWORKSPACE;
        p.partial = (float*)workspace;
    }
    int rc = vec ? dispatch_igemm<true>(pl, p, st) : dispatch_igemm<false>(pl, p, st);
    if (rc) return rc;
    if (pl.splits > 1) {
        const size_t total = (size_t)p.M * p.Cout;
        const int blocks = (int)min((size_t)2048, ceil_div_sz(total, 256));
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.partial, p.bias, p.out, p.out_ld,
                           p.M, p.Cout, pl.splits);
        SEMSEG_LAUNCH_CHECK();
    }
    return 0;
}

size_t igemm_workspace_bytes(int M, int Cout, int Cin, int T) {
    const IGemmPlan pl = plan_igemm(M, Cout, Cin, T, true);
    return pl.splits > 1 ? (size_t)pl.splits * M * Cout * sizeof(float) : 0;
}

static inline int out_dim(int in, int k, int stride, int pad, int dil) {
    return (in + 2 * pad - dil * (k - 1) - 1) / stride + 1;
}

extern "C" int semseg_conv2d_fwd(const float* x, int x_ld, const float* w, const float* bias, float* y, int y_ld,
                                 int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !w || !y || N <= 0 || C <= 0 || K <= 0 || stride <= 0 || dil <= 0 || x_ld < C || y_ld < K) return SEMSEG_EINVAL;
    const int OH = out_dim(H, R, stride, pad, dil), OW = out_dim(W, S, stride, pad, dil);
    if (OH <= 0 || OW <= 0) return SEMSEG_EINVAL;
    IGemmParams p = {};
    p.in = x; p.wgt = w; p.bias = bias; p.out = y;
    p.in_ld = x_ld; p.out_ld = y_ld;
    p.Hin = H; p.Win = W; p.Cin = C;
    p.Hout = OH; p.Wout = OW; p.Cout = K;
    p.M = N * OH * OW;
    p.S = S; p.T = R * S;
    p.a = stride; p.off = -pad; p.step = dil; p.div = 1;
    const bool vec = (C % 4 == 0) && (x_ld % 4 == 0) && aligned16(x) && aligned16(w);
    return run_igemm(p, vec, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int semseg_conv2d_dgrad(const float* dy, int dy_ld, const float* wt, float* dx, int dx_ld,
                                   int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy || !wt || !dx || N <= 0 || C <= 0 || K <= 0 || stride <= 0 || dil <= 0 || dy_ld < K || dx_ld < C) return SEMSEG_EINVAL;
    const int OH = out_dim(H, R, stride, pad, dil), OW = out_dim(W, S, stride, pad, dil);
    if (OH <= 0 || OW <= 0) return SEMSEG_EINVAL;
    IGemmParams p = {};
    p.in = dy; p.wgt = wt; p.bias = nullptr; p.out = dx;
    p.in_ld = dy_ld; p.out_ld = dx_ld;
    p.Hin = OH; p.Win = OW; p.Cin = K;
    p.Hout = H; p.Wout = W; p.Cout = C;
    p.M = N * H * W;
    p.S = S; p.T = R * S;
    p.a = 1; p.off = pad; p.step = -dil; p.div = stride;
    const bool vec = (K % 4 == 0) && (dy_ld % 4 == 0) && aligned16(dy) && aligned16(wt);
    return run_igemm(p, vec, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int semseg_weight_krsc_to_crsk(const float* w, float* wt, int K, int T, int C, void* stream) {
    if (!w || !wt || K <= 0 || T <= 0 || C <= 0) return SEMSEG_EINVAL;
    dim3 grid(ceil_div(C, 32), ceil_div(K, 32), T);
    hipLaunchKernelGGL(weight_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wt, K, T, C);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}
