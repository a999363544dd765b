// fp32-accurate convolution on the 16-bit MFMA of gfx950 by operand splitting.
//
// Why: v_mfma_f32_32x32x2_f32 (exact fp32, conv_igemm.hip) runs at the fp32 VECTOR rate, 157 TFLOP/s --
// 1/16 of v_mfma_f32_32x32x16_{bf16,f16} (2.5 PFLOP/s).  Two split schemes share every kernel in this file:
//
//  s3  (3 x bf16, 6 products; scale free)
//      a = a0 + a1 + a2,  a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)          (8 + 8 + 8 mantissa bits)
//      a*b ~ a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0)     dropped terms <= 2^-24 |ab|
//      ceiling 2.5 PF / 6 = 417 TFLOP/s of fp32-equivalent convolution.
//
//  h2  (2 x fp16, 3 products; per-tensor power-of-two scale)
//      A = 2^e a with e chosen per TENSOR so that max|A| lies in [2^14, 2^15)   (absmax pass, exact scaling)
//      A = A0 + A1 + rho,  A0 = fp16(A), A1 = fp16(A - A0)                                  (11 + 11 mantissa bits)
//      |rho| <= 2^-22 |A| for every element >= 2^-18 max|A| (A1 normal); below that A1 is an fp16 subnormal and
//      |rho| <= 2^-25 = 2^-40 max|A| ABSOLUTE -- error relative to the tensor's large elements, which is what a
//      dot product sees.
//      A*B ~ A0 B0 + A0 B1 + A1 B0     dropped A1 B1 <= 2^-22 |AB|; the fp32 accumulator is descaled by
//      2^-(ea+eb) in the epilogue (exact).  Per-product error <= ~3 * 2^-22, rms ~2e-7: below the ~sqrt(K) 2^-24
//      rounding noise every fp32 accumulation of K >= 27 products carries (measured: tools/s3_check.py).
//      ceiling 2.5 PF / 3 = 833 TFLOP/s of fp32-equivalent convolution.
//
// Operands are split ONCE per tensor into dense 16-bit planes [NP][rows][pitch] (channels rounded up to 32, zero
// filled) -- the conv kernels then stream 16-bit data with no conversion work in their main loops.
// Activations: rows = pixels (NHWC); weights: rows = (k, tap) (KRSC) or (c, tap) (CRSK, for the data gradient).
//
// Kernels
//   split3_kernel / absmax_partial_kernel + split_h2_kernel     fp32 rows -> planes, HBM-bound streaming passes
//   igemm_rs_kernel     forward AND data-gradient implicit GEMM (tap algebra of conv_igemm.hip), 4 waves,
//                       register-staged global loads, one LDS buffer; small / odd layers
//   igemm_dma_kernel    same GEMM for the large layers: 8 waves, LDS filled by buffer_load_dwordx4 ... lds (no
//                       staging VGPRs, no ds_write), 2-slot (two barriers per k-tile) or 3-slot ring (one barrier
//                       per k-tile, two tiles in flight)
//   wgrad_kernel        weight gradient: reduction over pixels = the slow axis of both NHWC operands; tiles are
//                       staged pixel-major and the MFMA fragments (8 consecutive pixels per lane) are produced by
//                       the LDS transpose read ds_read_b64_tr_b16
// Replaces nn.Conv2d at reference resnet.py:18-21,61-66,130; models.py:163,406,448,456-463,519-540;
// hrnet.py:26-29,188-205,316-338 and its autograd backward.
#include "common.h"
#include "split_layout.h"
#include "batch.h"
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <array>
#include <map>
#include <mutex>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------
// split schemes
// ------------------------------------------------------------------------------------------------
struct SchS3 {
    static constexpr int ID = 0;
    static constexpr int NP = 3;
    static constexpr bool SCALED = false;
    typedef bf16x8 frag;
    // smallest terms first
    static __device__ __forceinline__ void mac(const frag (&a)[3], const frag (&b)[3], f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
};

struct SchH2 {
    static constexpr int ID = 1;
    static constexpr int NP = 2;
    static constexpr bool SCALED = true;
    typedef f16x8 frag;
    static __device__ __forceinline__ void mac(const frag (&a)[2], const frag (&b)[2], f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
    }
};

// ------------------------------------------------------------------------------------------------
// launch-plan overrides (set by the host-side tuner, mit_semseg/tuner.py): scheme + conv geometry + pass -> (tile, split)
// ------------------------------------------------------------------------------------------------
typedef std::array<int, 12> SKey;     // scheme, pass, N, H, W, C, K, R, S, stride, pad, dil
static std::map<SKey, std::pair<int, int>> g_plans;
static std::mutex g_plans_mu;

// Plans of launches that are being RECORDED for a side-by-side launch (csrc/batch.h; HRNetV2's branches): the per-geometry plans
// were timed with the chip to one problem -- every geometry its own tile form, small ones split along K to reach 256 CUs -- but
// records pair up only under ONE kernel instantiation, and in a launch of four problems the others fill the chip.  So inside a
// scope the forward / data-gradient GEMMs of small problems CAN run on ONE tile form without a split: SEMSEG_BATCH_TILE (15 / 17 / 19:
// the 64 x 64 tiles on the LDS-DMA rings, which take every channel count; 23 = the 64-deep form where the reduction is whole
// 64-channel chunks, else 17), up to SEMSEG_BATCH_MAX_TILES 64 x 64 tiles per problem (default 1024; larger problems keep their own
// plan).  Measured on HRNetV2 at 2 x 512 x 512 (gpurun r8e, ms per step): no scope 25.27; per-geometry plans inside the scopes 22.32;
// tile 15: 22.58, 19: 22.63, 17: 23.81, 23: 24.51, 2: 24.95 -- a GEMM of four problems on one tile form is bound by the chip's block
// slots (80 KiB of LDS per 64-deep block: two per CU), not by its launches, and the per-geometry tiles are each the fastest for
// their problem.  Default -1: the per-geometry plans (the GEMMs of a position pair up where their plans name the same kernel).
static int g_batch_tile = [] { const char* e = getenv("SEMSEG_BATCH_TILE"); return e ? atoi(e) : -2; }();
static int g_batch_max_tiles = [] { const char* e = getenv("SEMSEG_BATCH_MAX_TILES"); return e ? atoi(e) : 1024; }();
// the same two knobs at run time (tests compare the side-by-side launches bit for bit with the sequential ones on the per-geometry
// plans: tile -1); returns the previous tile
extern "C" int semseg_batch_plan(int tile, int max_tiles) {
    const int prev = g_batch_tile;
    g_batch_tile = tile;
    if (max_tiles > 0) g_batch_max_tiles = max_tiles;
    return prev;
}
static bool batch_plan(int pass, int M, int Cout, int Cred, int* tile, int* split) {
    if (pass > 1 || !semseg_batch::recording()) return false;
    const int want = g_batch_tile, max_tiles = g_batch_max_tiles;
    if (want < 0) return false;
    if ((long)((M + 63) / 64) * ((Cout + 63) / 64) > max_tiles) return false;
    *tile = want;
    *split = 1;
    return true;
}

// The plan of a SMALL weight gradient as a member of a batch (round 6).  The per-geometry plans were timed with the chip to one launch:
// a 96 x 96 x 3 x 3 weight gradient wins on one 128 x 128 LDS-DMA tile split 24 ways (20 us) -- but inside TrainStep's deferral
// (ops.defer_wgrad_reduces) the gradients whose plan is the register-staged 64 x 64 tile wait for backward to return and run 24
// problems per launch (semseg_conv2d_wgrad_multi_h2), where the launch is shared.  While the host has the member mode on
// (semseg_conv2d_wgrad_member_plan, around the deferral's own calls only -- never while the tuner times a plan), a plan on any
// other tile becomes the 64 x 64 tile with the same number of blocks, for weight tensors of at most `max_tiles` 64 x 64 tiles
// (default 4: K, C <= 128).  HRNetV2 (its 32 weight gradients of the 96-channel branch): 19.29 -> 18.36 ms per step (gpurun r9g).
static int g_wgrad_member = 0, g_wgrad_member_tiles = [] { const char* e = getenv("SEMSEG_WGRAD_MEMBER_TILES"); return e ? atoi(e) : 4; }();
static const int kWTiles[15][2] = {{128, 128}, {64, 64}, {128, 128}, {256, 128}, {256, 256}, {128, 128}, {256, 256}, {256, 256},
                                    {128, 128}, {128, 128}, {64, 64}, {256, 256}, {256, 256}, {128, 128}, {128, 128}};
static void wgrad_member_plan(int K, int C, int T, int* tile, int* split) {
    if (!g_wgrad_member || *tile < 0 || *tile == 1 || *tile == 10 || *tile > 14) return;      // 10: the all-taps kernel wins its layers by fetching a third of the bytes
    const int nt = ((K + 63) / 64) * ((C + 63) / 64);
    if (nt > g_wgrad_member_tiles) return;
    const long old_blocks = (long)((K + kWTiles[*tile][0] - 1) / kWTiles[*tile][0]) * ((C + kWTiles[*tile][1] - 1) / kWTiles[*tile][1]) * T *
                            (*split > 0 ? *split : 1);
    long sp = old_blocks / ((long)nt * T);
    if (sp < 1) sp = 1;
    if (sp > 64) sp = 64;
    *tile = 1;
    *split = (int)sp;
}
extern "C" int semseg_conv2d_wgrad_member_plan(int enable) {
    const int prev = g_wgrad_member;
    g_wgrad_member = enable ? 1 : 0;
    return prev;
}

static bool lookup_plan(int sch, int pass, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                        int* tile, int* split) {
    if (sch == 1 /* SchH2::ID */ && pass <= 1) {
        const int OH = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1, OW = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
        if (pass == 0 ? batch_plan(0, N * OH * OW, K, C, tile, split) : batch_plan(1, N * H * W, C, K, tile, split)) return true;
    }
    std::lock_guard<std::mutex> lk(g_plans_mu);
    if (g_plans.empty()) return false;
    auto it = g_plans.find(SKey{sch, pass, N, H, W, C, K, R, S, stride, pad, dil});
    if (it == g_plans.end()) return false;
    *tile = it->second.first;
    *split = it->second.second;
    if (sch == 1 && pass == 2) wgrad_member_plan(K, C, R * S, tile, split);
    // SEMSEG_BATCH_TILE=-2 (default): inside a side-by-side scope every 64 x 64 form of the per-geometry plans (register staged,
    // 2- / 3- / 5-slot rings) becomes THE 64-deep form, split kept -- the branches' plans then name one kernel where they differed
    // only in the ring (the 96-channel branch could not take the 64-deep kernel when the plans were timed) and their GEMMs pair up
    if (sch == 1 && pass <= 1 && g_batch_tile == -2 && semseg_batch::recording() &&
        (*tile == 2 || *tile == 15 || *tile == 17 || *tile == 19))
        *tile = 23;
    // measured and removed again (gpurun r8q / r8r, HRNetV2 19.24 ms per step with the mapping above): no split-K for the recorded 64-deep
    // GEMMs 19.66 ms; k loops equalised to at most 9 / 14 / 18 / 27 k-tiles per block by more split-K 20.83 / 20.08 / 19.53 / 19.31 ms (a split
    // costs the statistics sweep and a reduce); plans re-timed with the 64-deep tile open to the 96-channel layers 19.79 against 19.59 ms;
    // the 128-wide forms mapped to the 64-deep 64 x 64 tile as well: 18.13 against 18.14 ms (gpurun r9j)
    return true;
}

// fwd/dgrad tiles: 0 = 128x128, 1 = 128x64, 2 = 64x64 (register staged); 3 = 256x128 LDS-DMA 2-slot;
// h2 only: 4 = 256x128 LDS-DMA 3-slot ring, 5 = 256x256 LDS-DMA 2-slot, 6 = 128x128 LDS-DMA 2-slot (8 waves, 64 KiB of LDS:
// two blocks per CU overlap each other's prologue / epilogue on the short-K layers); 7 / 8 / 9 = the 2-slot tiles 3 / 5 / 6 with
// software-pipelined fragment reads (one barrier per k-tile), 10 = the 3-slot ring 4 with the same pipeline.  wgrad tiles: 0 = 128x128, 1 = 64x64 (register
// staged); h2 only: 2 = 128x128 LDS-DMA 2-slot, 3 = 256x128 LDS-DMA 3-slot ring, 4 = 256x256 LDS-DMA 2-slot, 5 / 6 = 2 / 4 with
// software-pipelined fragment reads.
static int max_tile(int sch, int pass) { return pass == 2 ? (sch == SchH2::ID ? 14 : 1) : (sch == SchH2::ID ? 26 : 3); }

static inline bool tile_is_k64(int tile);
// wgrad tile 10 (wgrad_taps_kernel) takes 3x3 stride-1 pad == dil convolutions whose output rows are whole 32-pixel chunks
static bool wtaps_geometry_ok(int N, int H, int W, int R, int S, int stride, int pad, int dil) {
    if (!(R == 3 && S == 3 && stride == 1 && pad == dil && dil >= 1 && dil <= 5)) return false;
    return W % 32 == 0 && ((long)N * H * W) % 32 == 0;          // OH = H, OW = W for these convolutions
}

static int set_plan(int sch, int pass, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil, int tile,
                    int split) {
    // pass 3: the batched GEMM of the Winograd forward (semseg_winograd_gemm_h2), keyed (tiles, 1, 1, C, K, 3, 3, 1, 1, 1); no split
    // split <= 64, except the all-taps weight-gradient tile (pass 2, tile 10): its tiles are 9x fewer, its splits fill the chip
    if (pass < 0 || pass > 3 || tile > max_tile(sch, pass == 3 ? 0 : pass) || split > ((pass == 2 && tile == 10) ? 512 : 64)) return SEMSEG_EINVAL;
    // a plan is refused where its kernel does not take the geometry (the tuner then skips the candidate; a plan inherited from
    // another image size of the same layer -- mit_semseg/tuner.py buckets -- is re-timed instead of failing at launch)
    if (pass == 2 && tile == 10 && !wtaps_geometry_ok(N, H, W, R, S, stride, pad, dil)) return SEMSEG_EINVAL;
    // tiles 22 ... 24 (64-deep k-tiles) in the batched Winograd GEMM: the reduction C padded to 32 channels must be whole 64-channel chunks
    // (round 6: the forward / data-gradient kernel takes an odd count too -- the missing half chunk is fetched as zeros)
    if (pass == 3 && tile_is_k64(tile) && ((C + 31) / 32) % 2) return SEMSEG_EINVAL;
    if (pass == 3 && tile >= 0 && (sch != SchH2::ID || split != 1 || !(tile == 0 || tile == 6 || tile == 7 || tile == 8 || tile == 9 ||
                                                                       tile == 10 || tile == 14 || tile == 22 || tile == 24 || tile == 25 || tile == 26)))
        return SEMSEG_EINVAL;
    std::lock_guard<std::mutex> lk(g_plans_mu);
    const SKey key{sch, pass, N, H, W, C, K, R, S, stride, pad, dil};
    if (tile < 0 || split < 1) g_plans.erase(key);
    else g_plans[key] = std::make_pair(tile, split);
    return 0;
}

extern "C" int semseg_conv2d_s3_set_plan(int pass, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                         int tile, int split) {
    return set_plan(SchS3::ID, pass, N, H, W, C, K, R, S, stride, pad, dil, tile, split);
}
extern "C" int semseg_conv2d_h2_set_plan(int pass, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                         int tile, int split) {
    return set_plan(SchH2::ID, pass, N, H, W, C, K, R, S, stride, pad, dil, tile, split);
}

// Wait for the MFMA pipe to retire the last accumulator writes before the epilogue reads them (belt and braces on top of
// the compiler's own hazard nops: 2 x 16 wait states > the 8-pass latency of v_mfma_f32_32x32x16_*).
#define S_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")

// plane layout, pitch, zero tail and the h2 header: split_layout.h
template <class SCH>
static size_t split_bytes(int rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    return (size_t)SCH::NP * rows * split_pitch(C) * sizeof(uint16_t) + SPLIT_ZERO_TAIL_BYTES + (SCH::SCALED ? H2_HDR_BYTES : 0);
}

// ------------------------------------------------------------------------------------------------
// split: s3
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split3_of(float v, __bf16& h0, __bf16& h1, __bf16& h2) {
    h0 = (__bf16)v;
    const float r1 = v - (float)h0;      // exact
    h1 = (__bf16)r1;
    const float r2 = r1 - (float)h1;     // exact
    h2 = (__bf16)r2;
}

// one thread = one group of 8 channels of one row: reads 32 B, writes 3 x 16 B
template <bool VEC>
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, int x_ld, uint16_t* __restrict__ out,
                                                     int rows, int C, int Cp, int pitch, size_t plane) {
    const int G = Cp >> 3;
    const size_t total = (size_t)rows * G;
    if (blockIdx.x == 0 && threadIdx.x < SPLIT_ZERO_TAIL_BYTES / 16)
        reinterpret_cast<uint4*>(out + 3 * plane)[threadIdx.x] = make_uint4(0, 0, 0, 0);
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int row = (int)(idx / G);
        const int g = (int)(idx - (size_t)row * G);
        const int c = g << 3;
        float v[8];
        const float* src = x + (size_t)row * x_ld + c;
        if (VEC && c + 8 <= C) {
            const float4 a = *reinterpret_cast<const float4*>(src);
            const float4 b = *reinterpret_cast<const float4*>(src + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (c + e < C) ? src[e] : 0.f;
        }
        bf16x8 p0, p1, p2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 h0, h1, h2;
            split3_of(v[e], h0, h1, h2);
            p0[e] = h0; p1[e] = h1; p2[e] = h2;
        }
        const size_t o = (size_t)row * pitch + c;
        *reinterpret_cast<bf16x8*>(out + o) = p0;
        *reinterpret_cast<bf16x8*>(out + plane + o) = p1;
        *reinterpret_cast<bf16x8*>(out + 2 * plane + o) = p2;
    }
}

extern "C" size_t semseg_split3_bytes(int rows, int C) { return split_bytes<SchS3>(rows, C); }

extern "C" int semseg_split3(const float* x, int x_ld, void* xs, int rows, int C, void* stream) {
    if (!x || !xs || rows <= 0 || C <= 0 || x_ld < C) return SEMSEG_EINVAL;
    const int Cp = round_up32(C), pitch = split_pitch(C);
    const size_t plane = (size_t)rows * pitch;
    const size_t total = (size_t)rows * (Cp >> 3);
    const int blocks = (int)min((size_t)16384, ceil_div_sz(total, 256));
    const bool vec = (x_ld % 4 == 0) && aligned16(x);
    if (vec)
        hipLaunchKernelGGL(split3_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, x_ld, (uint16_t*)xs,
                           rows, C, Cp, pitch, plane);
    else
        hipLaunchKernelGGL(split3_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, x_ld, (uint16_t*)xs,
                           rows, C, Cp, pitch, plane);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// split: h2 (absmax pass + scaled 2 x fp16 split)
// ------------------------------------------------------------------------------------------------
// partial[b] = max |x| over the rows x C window, as fp32 bit patterns (monotone for non-negative floats; a NaN sorts
// above +inf and is handled by h2_exponent)
template <bool VEC>
__global__ __launch_bounds__(256) void absmax_partial_kernel(const float* __restrict__ x, int x_ld, int rows, int C,
                                                             uint32_t* __restrict__ partial) {
    const int G = (C + 7) >> 3;
    const size_t total = (size_t)rows * G;
    uint32_t m = 0;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int row = (int)(idx / G);
        const int c = (int)(idx - (size_t)row * G) << 3;
        const float* src = x + (size_t)row * x_ld + c;
        if (VEC && c + 8 <= C) {
            const float4 a = *reinterpret_cast<const float4*>(src);
            const float4 b = *reinterpret_cast<const float4*>(src + 4);
            m = max(m, max(max(absbits(a.x), absbits(a.y)), max(absbits(a.z), absbits(a.w))));
            m = max(m, max(max(absbits(b.x), absbits(b.y)), max(absbits(b.z), absbits(b.w))));
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c + e < C) m = max(m, absbits(src[e]));
        }
    }
    m = block_max_u32(m);
    if (threadIdx.x == 0) partial[blockIdx.x] = m;
}

template <bool VEC>
__global__ __launch_bounds__(256) void split_h2_kernel(const float* __restrict__ x, int x_ld, uint16_t* __restrict__ out,
                                                       int rows, int C, int Cp, int pitch, size_t plane,
                                                       const uint32_t* __restrict__ partial, int npartial,
                                                       int* __restrict__ hdr) {
    uint32_t m = 0;
    for (int i = threadIdx.x; i < npartial; i += 256) m = max(m, partial[i]);
    m = block_max_u32(m);
    const int ex = h2_exponent(m);
    const float sc = pow2i(ex);
    if (blockIdx.x == 0) {
        if (threadIdx.x < SPLIT_ZERO_TAIL_BYTES / 16) reinterpret_cast<uint4*>(out + 2 * plane)[threadIdx.x] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) hdr[0] = ex;
    }
    const int G = Cp >> 3;
    const size_t total = (size_t)rows * G;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int row = (int)(idx / G);
        const int c = (int)(idx - (size_t)row * G) << 3;
        float v[8];
        const float* src = x + (size_t)row * x_ld + c;
        if (VEC && c + 8 <= C) {
            const float4 a = *reinterpret_cast<const float4*>(src);
            const float4 b = *reinterpret_cast<const float4*>(src + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (c + e < C) ? src[e] : 0.f;
        }
        f16x8 p0, p1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float s = v[e] * sc;                  // exact (power of two)
            const _Float16 h0 = (_Float16)s;
            const float r = s - (float)h0;              // exact
            p0[e] = h0;
            p1[e] = (_Float16)r;
        }
        const size_t o = (size_t)row * pitch + c;
        *reinterpret_cast<f16x8*>(out + o) = p0;
        *reinterpret_cast<f16x8*>(out + plane + o) = p1;
    }
}

extern "C" size_t semseg_split_h2_bytes(int rows, int C) { return split_bytes<SchH2>(rows, C); }

extern "C" int semseg_split_h2(const float* x, int x_ld, void* xs, int rows, int C, void* stream) {
    if (!x || !xs || rows <= 0 || C <= 0 || x_ld < C) return SEMSEG_EINVAL;
    const int Cp = round_up32(C), pitch = split_pitch(C);
    const size_t plane = (size_t)rows * pitch;
    int* hdr = const_cast<int*>(h2_exp_ptr(xs, (size_t)rows, C));
    uint32_t* partial = reinterpret_cast<uint32_t*>(hdr) + 64;        // 256 B after the exponent word
    const bool vec = (x_ld % 4 == 0) && aligned16(x);
    const size_t total_c = (size_t)rows * ((C + 7) >> 3);
    const int nb = (int)min((size_t)H2_MAX_PARTIALS, ceil_div_sz(total_c, 1024));     // >= 4 groups per thread
    if (vec)
        hipLaunchKernelGGL(absmax_partial_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, x_ld, rows, C, partial);
    else
        hipLaunchKernelGGL(absmax_partial_kernel<false>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, x_ld, rows, C, partial);
    SEMSEG_LAUNCH_CHECK();
    const size_t total = (size_t)rows * (Cp >> 3);
    const int blocks = (int)min((size_t)16384, ceil_div_sz(total, 256));
    if (vec)
        hipLaunchKernelGGL(split_h2_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, x_ld, (uint16_t*)xs, rows,
                           C, Cp, pitch, plane, partial, nb, hdr);
    else
        hipLaunchKernelGGL(split_h2_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, x_ld, (uint16_t*)xs, rows,
                           C, Cp, pitch, plane, partial, nb, hdr);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// The same split when an UPPER BOUND of max|x| is already known as device scalars (BN outputs of the fused chain carry one;
// max over the inputs of a concat / pooling window): the exponent comes from the bound(s) -- any upper bound will do, h2_exponent
// -- and the absmax pass over x is not needed: one launch instead of two.
struct SplitBounds {
    const float* p[8];
    int n;
};
template <bool VEC>
struct split_h2_bounds_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ x, int x_ld, uint16_t* __restrict__ out,
                                                              int rows, int C, int Cp, int pitch, size_t plane, SplitBounds b,
                                                              int* __restrict__ hdr) {
    float bound = 0.f;
    bool bad = false;
    for (int i = 0; i < b.n; ++i) {
        const float v = b.p[i][0];
        bad = bad || !(v == v);
        bound = fmaxf(bound, v);
    }
    const int ex = h2_exponent(bad ? 0x7fc00000u : __float_as_uint(bound));
    const float sc = pow2i(ex);
    if (blockIdx.x == 0) {
        if (threadIdx.x < SPLIT_ZERO_TAIL_BYTES / 16) reinterpret_cast<uint4*>(out + 2 * plane)[threadIdx.x] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) hdr[0] = ex;
    }
    const int G = Cp >> 3;
    const size_t total = (size_t)rows * G;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int row = (int)(idx / G);
        const int c = (int)(idx - (size_t)row * G) << 3;
        float v[8];
        const float* src = x + (size_t)row * x_ld + c;
        if (VEC && c + 8 <= C) {
            const float4 a = *reinterpret_cast<const float4*>(src);
            const float4 bb = *reinterpret_cast<const float4*>(src + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bb.x; v[5] = bb.y; v[6] = bb.z; v[7] = bb.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (c + e < C) ? src[e] : 0.f;
        }
        f16x8 p0, p1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 h0, h1;
            h2_split_of(v[e] * sc, h0, h1);
            p0[e] = h0;
            p1[e] = h1;
        }
        const size_t o = (size_t)row * pitch + c;
        *reinterpret_cast<f16x8*>(out + o) = p0;
        *reinterpret_cast<f16x8*>(out + plane + o) = p1;
    }
    }
};

extern "C" int semseg_split_h2_bounds(const float* x, int x_ld, void* xs, int rows, int C, const float* const* bounds_host,
                                      int nbounds, void* stream) {
    if (!x || !xs || rows <= 0 || C <= 0 || x_ld < C || !bounds_host || nbounds <= 0 || nbounds > 8) return SEMSEG_EINVAL;
    SplitBounds b;
    b.n = nbounds;
    for (int i = 0; i < 8; ++i) b.p[i] = i < nbounds ? bounds_host[i] : nullptr;
    for (int i = 0; i < nbounds; ++i)
        if (!b.p[i]) return SEMSEG_EINVAL;
    const int Cp = round_up32(C), pitch = split_pitch(C);
    const size_t plane = (size_t)rows * pitch;
    int* hdr = const_cast<int*>(h2_exp_ptr(xs, (size_t)rows, C));
    const bool vec = (x_ld % 4 == 0) && aligned16(x);
    const size_t total = (size_t)rows * (Cp >> 3);
    const int blocks = (int)min((size_t)16384, ceil_div_sz(total, 256));
    if (vec)
        SEMSEG_LAUNCH_BODY((split_h2_bounds_kernel_body<true>), dim3(blocks), 0, (hipStream_t)stream, x, x_ld, (uint16_t*)xs, rows,
                           C, Cp, pitch, plane, b, hdr);
    else
        SEMSEG_LAUNCH_BODY((split_h2_bounds_kernel_body<false>), dim3(blocks), 0, (hipStream_t)stream, x, x_ld, (uint16_t*)xs,
                           rows, C, Cp, pitch, plane, b, hdr);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// out[0] = sum of `n` (<= 8) non-negative device scalars: the bound of |a + b + ...| from the bounds of the terms (the exchange
// sums of hrnet.py:231-248, whose result feeds the residual branch of the next module's blocks: without a bound of it the BN
// kernels of those blocks cannot bound their own outputs and every conv of the branch falls back to absmax + split passes)
struct bound_sum_kernel_body {
    static constexpr int THREADS = 1;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               SplitBounds b, float* __restrict__ out) {
    float s = 0.f;
    bool bad = false;
    for (int i = 0; i < b.n; ++i) {
        const float v = b.p[i][0];
        bad = bad || !(v == v);
        s += v;
    }
    out[0] = bad ? __uint_as_float(0x7fc00000u) : s * 1.0000005f;      // rounded up: stays an upper bound of the exact sum
    }
};

extern "C" int semseg_bound_sum(const float* const* bounds_host, int nbounds, float* out, void* stream) {
    if (!bounds_host || !out || nbounds <= 0 || nbounds > 8) return SEMSEG_EINVAL;
    SplitBounds b;
    b.n = nbounds;
    for (int i = 0; i < 8; ++i) b.p[i] = i < nbounds ? bounds_host[i] : nullptr;
    for (int i = 0; i < nbounds; ++i)
        if (!b.p[i]) return SEMSEG_EINVAL;
    SEMSEG_LAUNCH_BODY((bound_sum_kernel_body), dim3(1), 0, (hipStream_t)stream, b, out);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// max |x| of an fp32 [rows][x_ld] window as ONE device scalar (the bound the Winograd input transform scales by when no producer
// kernel left one: the evaluation-mode forward, where BN applies running statistics and carries no min / max)
__global__ __launch_bounds__(256) void absmax_finish_kernel(const uint32_t* __restrict__ partial, int npartial,
                                                            float* __restrict__ out) {
    uint32_t m = 0;
    for (int i = threadIdx.x; i < npartial; i += 256) m = max(m, partial[i]);
    m = block_max_u32(m);
    if (threadIdx.x == 0) out[0] = __uint_as_float(m);
}

extern "C" int semseg_absmax(const float* x, int x_ld, int rows, int C, float* out, void* workspace, size_t workspace_bytes,
                             void* stream) {
    if (!x || !out || rows <= 0 || C <= 0 || x_ld < C) return SEMSEG_EINVAL;
    const size_t total_c = (size_t)rows * ((C + 7) >> 3);
    const int nb = (int)min((size_t)H2_MAX_PARTIALS, ceil_div_sz(total_c, 1024));
    if (!workspace || workspace_bytes < (size_t)nb * sizeof(uint32_t)) return SEMSEG_EWORKSPACE;
    uint32_t* partial = (uint32_t*)workspace;
    if ((x_ld % 4 == 0) && aligned16(x))
        hipLaunchKernelGGL(absmax_partial_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, x_ld, rows, C, partial);
    else
        hipLaunchKernelGGL(absmax_partial_kernel<false>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, x_ld, rows, C, partial);
    SEMSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(absmax_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)partial, nb, out);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// forward / data-gradient implicit GEMM on split operands
// ------------------------------------------------------------------------------------------------
struct SParams {
    const uint16_t* in;    // [NP][in_rows][pitch]
    const uint16_t* wgt;   // [NP][Cout*T][pitch]
    const int* in_exp;     // h2: exponent words of the two operands
    const int* w_exp;
    uint32_t in_plane, w_plane;   // elements per plane (host checks the byte size < 2^31)
    const float* bias;
    float* out;
    float* partial;
    int Cp, pitch, out_ld;   // Cp: channels padded to 32; pitch: row stride of the split planes (elements)
    int Hin, Win;
    int Hout, Wout, Cout;
    int M;
    int S, T;
    int a, off, step, div;
    int chunks, ktiles, kt_per_split;
    int tiles_m, tiles_n, splits;
    int batches;                                       // 0 / 1: plain; > 1: gridDim.z batches (splits must be 1)
    int tn_fast;                                        // block order inside an XCD: column tiles fastest (run_gemm)
    int k64_tail;                                       // 64-deep k-tiles on a reduction of an ODD number of 32-channel chunks: the upper half of the
                                                        // last 64-channel chunk of every tap does not exist and is fetched as zeros (igemm_dma64_kernel)
    int batch_in_rows, batch_w_rows, batch_out_rows;   // batched GEMM (one batch per grid z): row offsets per batch of the
                                                       // activation planes, of the weight planes (in units of T rows) and of `out`
    // BN statistics of the result gathered in the epilogue (forward convs that a BatchNorm follows, splits == 1 only): every
    // BLOCK writes the column sums / sums of squares (fp64) and column minima / maxima (fp32) of its BM x BN tile as partial row
    // tm of  st_sum[tiles_m][2 Cout] / st_mm[tiles_m][2 Cout] -- the layout the BN finish kernels reduce (csrc/bn.hip), so the
    // separate statistics sweep over the result (one launch, one read of it) is not needed
    double* st_sum;
    float* st_mm;
    float* st_zero;          // the |y| bound word the finish kernel max-reduces into with atomics: zeroed here
};

// (chunk, tap) walk of the k loop, shared by both GEMM kernels.  Default order: channel chunk major, taps inner -- the T
// taps of one 32-channel chunk re-read (shifted) the same pixels in consecutive k-tiles, so only the first tap of a chunk
// misses L2, and a split-K slice is a channel range (profiles/r1i: HBM-side fetch of conv_last fwd 1.75 GB -> 0.34 GB).
struct KWalk {
    int cc, t, r, s;     // channel chunk, tap, tap row / column of the NEXT k-tile
    bool dirty;          // the tap changed since the last set_tap
    __device__ __forceinline__ void init(const SParams& p, int kt) {
        if (kt == 0) {                     // wave-uniform; every launch without split-K: no divisions in the prologue
            cc = 0; t = 0; r = 0; s = 0;
        } else {
            cc = kt / p.T;
            t = kt - cc * p.T;
            r = t / p.S;
            s = t - r * p.S;
        }
        dirty = true;
    }
    __device__ __forceinline__ void next_tap(const SParams& p) {
        dirty = true;
        if (++s == p.S) { s = 0; ++r; }
        if (++t == p.T) { t = 0; r = 0; s = 0; }
    }
    __device__ __forceinline__ void advance(const SParams& p) {
        if (p.T > 1) {
            next_tap(p);
            if (t == 0) ++cc;
        } else {
            ++cc;
        }
    }
};

// Output pixel m -> (image n, row oh, column ow) in the prologue of the GEMM kernels.  An integer division by a run-time divisor
// is a ~30-instruction dependent chain on this part and the prologue had up to six of them in front of the first DMA piece; for
// m < 2^24 (every map of the path: 2 x 512 x 512 = 2^19) a float reciprocal + ONE correction step is exact (|q_est - m / d| < 1: the
// product of an exactly represented m and a correctly rounded 1 / d is within 2^-23 of m / d relative, and q < 2^23 for d >= 2; d = 1
// is exact) -- as wgrad_block_body has done since round 2.  Larger M takes the integer divisions.
struct PixDiv {
    int hw, w;
    float inv_hw, inv_w;
    bool small;
};
__device__ __forceinline__ PixDiv pix_div(int Hout, int Wout, int M) {
    PixDiv d;
    d.hw = Hout * Wout;
    d.w = Wout;
    d.inv_hw = 1.0f / (float)d.hw;
    d.inv_w = 1.0f / (float)Wout;
    d.small = M <= (1 << 24);
    return d;
}
__device__ __forceinline__ void pix_of(const PixDiv& d, int m, int& n, int& oh, int& ow) {
    if (d.small) {
        n = (int)((float)m * d.inv_hw);
        int rem = m - n * d.hw;
        if (rem < 0) { --n; rem += d.hw; } else if (rem >= d.hw) { ++n; rem -= d.hw; }
        oh = (int)((float)rem * d.inv_w);
        ow = rem - oh * d.w;
        if (ow < 0) { --oh; ow += d.w; } else if (ow >= d.w) { ++oh; ow -= d.w; }
    } else {
        n = m / d.hw;
        const int rem = m - n * d.hw;
        oh = rem / d.w;
        ow = rem - oh * d.w;
    }
}

// 2^-(ea+eb) as two factors (each within the normal range; see the header comment of the h2 scheme)
template <class SCH>
__device__ __forceinline__ void descale_factors_of(int s, float& f1, float& f2) {
    if constexpr (SCH::SCALED) {
        const int h = s / 2;               // |s| <= 200
        f1 = pow2i(-h);
        f2 = pow2i(-(s - h));
    } else {
        f1 = 1.f;
        f2 = 1.f;
    }
}
template <class SCH>
__device__ __forceinline__ void descale_factors(const int* ea, const int* eb, float& f1, float& f2) {
    int s = 0;
    if constexpr (SCH::SCALED) s = *ea + *eb;
    descale_factors_of<SCH>(s, f1, f2);
}
// The two exponent words of a GEMM's operands are requested at the TOP of the kernel (exp_request) and settled in a scalar
// register before the first LDS-DMA piece is issued (exp_settle): read in the epilogue they were one more memory latency at the end
// of every block.  The settle point matters: the LDS-DMA rings below count their vmcnt by hand, so no other vector load may be in
// flight among their pieces -- the value is consumed (readfirstlane) and a compiler-level memory barrier follows, so whatever load
// the compiler chose (scalar or vector) was issued and waited for before the first piece.
template <class SCH>
__device__ __forceinline__ int exp_request(const int* ea, const int* eb) {
    if constexpr (SCH::SCALED) return *ea + *eb;           // |sum| <= 200
    else return 0;
}
// The LDS-DMA kernels fetch the sum with scalar loads of their own BEHIND the prologue's DMA pieces (exp_fetch): SMEM counts in
// lgkmcnt, not in the vmcnt these kernels count by hand, and its latency hides behind the first tiles' flight.  One asm statement
// incl. the wait: the outputs are valid when it ends (an `s_load` left in flight across compiler-generated code could be copied or
// spilled before it lands).
template <class SCH>
__device__ __forceinline__ int exp_fetch(const int* ea, const int* eb) {
    if constexpr (SCH::SCALED) {
        int va, vb;
        asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(va), "=&s"(vb) : "s"(ea), "s"(eb) : "memory");
        return va + vb;
    } else {
        return 0;
    }
}
template <class SCH>
__device__ __forceinline__ int exp_settle(int e) {
    if constexpr (SCH::SCALED) {
        e = __builtin_amdgcn_readfirstlane(e);
        asm volatile("" : "+s"(e) : : "memory");
    }
    return e;
}

// LDS image of one operand tile: [part][row][4 x 16 B], the 16-byte slot of channel group q of row r stored at
// q ^ ((r >> 2) & 3): a ds_read_b128 lane group (16 lanes = 16 rows that are distinct mod 16, one q) then covers
// all 64 banks exactly once, and a ds_write_b128 lane group (8 lanes = 2 whole rows) covers 128 contiguous bytes.
__device__ __forceinline__ int s_slot(int row, int q) { return row * 4 + (q ^ ((row >> 2) & 3)); }

// epilogue shared by both GEMM kernels.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
// `bz` = batch of a batched GEMM, `tm` = row-tile index of the block, `wm` / `wcol` = the wave's row position in the block (0 .. WGM-1) and first column inside
// the block tile, `smem` = the block's LDS (the operand tiles are dead by now): when the launch gathers BN statistics
// (p.st_sum), the WGM waves that share a column range combine their column sums / minima / maxima through LDS in wave order,
// so that ONE partial row per block row tile reaches memory (tiles_m rows for the BN finish kernel instead of tiles_m x WGM).
template <class SCH, int FM, int FN, int WGM, int BN>
__device__ __forceinline__ void gemm_epilogue(const SParams& p, f32x16 (&acc)[FM][FN], int row0, int col0, int z, int bz, int lane,
                                              int tm, int wm, int wcol, void* smem, const bool first_block, const int esum) {
    // esum: sum of the operands' exponent words (exp_request / exp_settle at the top of the kernel)
    // first_block: this is block (0, 0) of ITS launch's grid -- a parameter, not the built-in index: inside a side-by-side launch
    // (csrc/batch.h) a problem's first block sits anywhere in the combined grid
    float* dst;
    int dst_ld;
    const bool direct = p.splits == 1;
    if (direct) {
        dst = p.out + (size_t)bz * p.batch_out_rows * p.out_ld;
        dst_ld = p.out_ld;
    } else {
        dst = p.partial + (size_t)z * p.M * p.Cout;
        dst_ld = p.Cout;
    }
    float f1, f2;
    descale_factors_of<SCH>(esum, f1, f2);
    const int col_l = lane & 31;
    const int row_l = 4 * (lane >> 5);
    const bool stats = direct && p.st_sum != nullptr;            // uniform over the block
    if (stats && p.st_zero && first_block && threadIdx.x == 0) p.st_zero[0] = 0.f;
    double su[FN], sq[FN];
    float mn[FN], mx[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        su[j] = 0.0; sq[j] = 0.0; mn[j] = INFINITY; mx[j] = -INFINITY;
        const int col = col0 + j * 32 + col_l;
        if (col >= p.Cout) continue;
        float bvl = (direct && p.bias) ? p.bias[col] : 0.f;
        // An UNCONDITIONAL use of the conditionally loaded bias: the one wait for its load happens here.  Without it every row's
        // store sat behind an `s_waitcnt vmcnt(0)` of its own -- the rows are conditional blocks (row < M), the compiler cannot
        // carry "already waited" across them, and on gfx9 vmcnt(0) also waits for the previous row's STORE to be acknowledged: 16 x
        // FM x FN stores per thread went out one round trip at a time (the "8 us epilogue" of the 64 x 64 blocks).
        asm volatile("" : "+v"(bvl));
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + i * 32 + (e & 3) + 8 * (e >> 2) + row_l;
                float v = acc[i][j][e];
                if constexpr (SCH::SCALED) v = (v * f1) * f2;
                if (row < p.M) {
                    v += bvl;
                    dst[(size_t)row * dst_ld + col] = v;
                    if (stats) {               // of the value as stored: what the BN that follows normalises
                        const double d = (double)v;
                        su[j] += d;
                        sq[j] = fma(d, d, sq[j]);
                        mn[j] = fminf(mn[j], v);
                        mx[j] = fmaxf(mx[j], v);
                    }
                }
            }
        }
    }
    if (!stats) return;
    // the other 16 rows of every 32-row fragment live in lane ^ 32; fixed order (low half + high half): deterministic
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        su[j] += __shfl_xor(su[j], 32);
        sq[j] += __shfl_xor(sq[j], 32);
        mn[j] = fminf(mn[j], __shfl_xor(mn[j], 32));
        mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], 32));
    }
    if constexpr (WGM > 1) {
        double* sd = reinterpret_cast<double*>(smem);                       // [WGM][BN][2]
        float* sf = reinterpret_cast<float*>(sd + (size_t)WGM * BN * 2);    // [WGM][BN][2]
        __syncthreads();                       // every wave has left the k loop: the operand tiles in LDS are dead
        if (lane < 32) {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int c = wcol + j * 32 + col_l;
                sd[((size_t)wm * BN + c) * 2] = su[j];
                sd[((size_t)wm * BN + c) * 2 + 1] = sq[j];
                sf[((size_t)wm * BN + c) * 2] = mn[j];
                sf[((size_t)wm * BN + c) * 2 + 1] = mx[j];
            }
        }
        __syncthreads();
        if (wm != 0 || lane >= 32) return;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int c = wcol + j * 32 + col_l;
#pragma unroll
            for (int w = 1; w < WGM; ++w) {    // wave order: deterministic
                su[j] += sd[((size_t)w * BN + c) * 2];
                sq[j] += sd[((size_t)w * BN + c) * 2 + 1];
                mn[j] = fminf(mn[j], sf[((size_t)w * BN + c) * 2]);
                mx[j] = fmaxf(mx[j], sf[((size_t)w * BN + c) * 2 + 1]);
            }
        }
    } else if (lane >= 32) {
        return;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int col = col0 + j * 32 + col_l;
        if (col >= p.Cout) continue;
        const size_t at = (size_t)tm * 2 * p.Cout + col;
        p.st_sum[at] = su[j];
        p.st_sum[at + p.Cout] = sq[j];
        p.st_mm[at] = mn[j];
        p.st_mm[at + p.Cout] = mx[j];
    }
}

template <class SCH, int BM, int BN>
struct igemm_rs_kernel_body {
    static constexpr int THREADS = 256;
    // the many-problem form only where small problems meet (HRNetV2 branches): h2 scheme, tiles up to 128 x 128
    static constexpr bool MULTI = std::is_same<SCH, SchH2>::value && BM <= 128 && BN <= 128;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const SParams p) {
    constexpr int NP = SCH::NP;
    typedef typename SCH::frag frag;
    constexpr int RPP = 64;                 // rows per load pass (256 threads x 16 B = 64 rows x 64 B)
    constexpr int APASS = BM / RPP, BPASS = BN / RPP;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 32, FN = WN / 32;
    static_assert(APASS >= 1 && BPASS >= 1 && FM >= 1 && FN >= 1, "tile too small");

    extern __shared__ __align__(16) uint4 smem4[];
    uint4* As = smem4;                       // [NP][BM][4]
    uint4* Bs = smem4 + NP * BM * 4;         // [NP][BN][4]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int esum = exp_request<SCH>(p.in_exp, p.w_exp);

    // block -> (tm, tn, z, batch): the blocks of one XCD (consecutive ids after xcd_remap) walk tm fastest, so they share ONE
    // weight slice (tn, z) -- streamed once into that XCD's L2 -- and read adjacent pixel tiles (shared halo rows); the
    // batches of a batched GEMM (Winograd positions) are dealt to the XCDs WHOLE, so both operands of a batch stay in one L2
    const GemmBlock gb = gemm_block(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), p.tiles_m, p.tiles_n, p.splits,
                                    gridDim.z, p.tn_fast);      // block_order.h
    const int tm = gb.tm, tn = gb.tn, z = gb.z, bz = gb.batch;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kt_begin = z * p.kt_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.kt_per_split);

    const int q = tid & 3;
    const int lrow = tid >> 2;

    int a_ih0[APASS], a_iw0[APASS], a_base[APASS];
    const PixDiv pd = pix_div(p.Hout, p.Wout, p.M);
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int m = m0 + lrow + i * RPP;
        if (m < p.M) {
            int n, oh, ow;
            pix_of(pd, m, n, oh, ow);
            a_ih0[i] = oh * p.a + p.off;
            a_iw0[i] = ow * p.a + p.off;
            a_base[i] = n * p.Hin * p.Win + bz * p.batch_in_rows;
        } else {
            a_ih0[i] = -(1 << 28);
            a_iw0[i] = -(1 << 28);
            a_base[i] = 0;
        }
    }

    uint4 ra[APASS][NP], rb[BPASS][NP];
    const uint32_t w_row = (uint32_t)p.T * p.pitch;
    uint32_t a_off[APASS], b_off[BPASS];
    uint32_t a_ok = 0, b_ok = 0;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
        const int n = n0 + lrow + i * RPP;
        const bool ok = n < p.Cout;
        b_off[i] = ok ? (uint32_t)(n + bz * p.batch_w_rows) * w_row + 8u * q : 0u;
        b_ok |= (ok ? 1u : 0u) << i;
    }
    KWalk kw;
    kw.init(p, kt_begin);
    // stride-1 geometry (div == 1; every forward conv and the dgrad of stride-1 convs): the offset of tap (r, s) relative
    // to tap (0, 0) is the same for every pixel, so a tap switch is one scalar delta + a per-row validity bit
    const bool fast_tap = (p.div == 1) && (p.T <= 32);          // block-uniform
    uint32_t a_row[APASS], a_vm[APASS];
    if (fast_tap) {
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            a_row[i] = ((uint32_t)a_base[i] + (uint32_t)a_ih0[i] * (uint32_t)p.Win + (uint32_t)a_iw0[i]) * (uint32_t)p.pitch +
                       8u * q;                                   // arithmetic mod 2^32 (halo rows are "negative")
            uint32_t vm = 0;
            int r = 0, s2 = 0;
            for (int t = 0; t < p.T; ++t) {
                const int nh = a_ih0[i] + r * p.step, nw = a_iw0[i] + s2 * p.step;
                vm |= ((nh >= 0) & (nw >= 0) & (nh < p.Hin) & (nw < p.Win)) ? (1u << t) : 0u;
                if (++s2 == p.S) { s2 = 0; ++r; }
            }
            a_vm[i] = vm;
        }
    }
    auto set_tap = [&](int t, int r, int s) {
        a_ok = 0;
        if (fast_tap) {
            const uint32_t delta = (uint32_t)((r * p.Win + s) * p.step) * (uint32_t)p.pitch;      // step < 0 for dgrad: mod 2^32
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                const bool ok = (a_vm[i] >> t) & 1u;
                a_off[i] = ok ? a_row[i] + delta : 0u;
                a_ok |= (ok ? 1u : 0u) << i;
            }
            return;
        }
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
            a_off[i] = ok ? (uint32_t)(a_base[i] + nh * p.Win + nw) * (uint32_t)p.pitch + 8u * q : 0u;
            a_ok |= (ok ? 1u : 0u) << i;
        }
    };

    // unconditional loads from clamped in-bounds offsets + select (see conv_igemm.hip on why not `cond ? load : 0`)
    auto load_tile = [&](int) {            // k-tiles are loaded in order: the (chunk, tap) walk advances here
        const int c0 = kw.cc * 32;
        if (kw.dirty) {                    // block-uniform
            set_tap(kw.t, kw.r, kw.s);
            kw.dirty = false;
        }
        const uint32_t koff = (uint32_t)kw.t * p.pitch + c0;
        kw.advance(p);
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const bool ok = (a_ok >> i) & 1u;
            const uint32_t o = ok ? a_off[i] + c0 : 0u;
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                const uint4 v = *reinterpret_cast<const uint4*>(p.in + (size_t)s * p.in_plane + o);
                ra[i][s] = ok ? v : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            const bool ok = (b_ok >> i) & 1u;
            const uint32_t o = ok ? b_off[i] + koff : 0u;
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                const uint4 v = *reinterpret_cast<const uint4*>(p.wgt + (size_t)s * p.w_plane + o);
                rb[i][s] = ok ? v : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < APASS; ++i)
#pragma unroll
            for (int s = 0; s < NP; ++s) As[s * BM * 4 + s_slot(lrow + i * RPP, q)] = ra[i][s];
#pragma unroll
        for (int i = 0; i < BPASS; ++i)
#pragma unroll
            for (int s = 0; s < NP; ++s) Bs[s * BN * 4 + s_slot(lrow + i * RPP, q)] = rb[i][s];
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 31;
    const int kb = lane >> 5;

    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile();
    }
    __syncthreads();

    auto compute_tile = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = 2 * ks + kb;
            frag av[FM][NP], bv[FN][NP];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int r = wm * WM + i * 32 + frow;
#pragma unroll
                for (int s = 0; s < NP; ++s) av[i][s] = *reinterpret_cast<const frag*>(&As[s * BM * 4 + s_slot(r, ch)]);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int r = wn * WN + j * 32 + frow;
#pragma unroll
                for (int s = 0; s < NP; ++s) bv[j][s] = *reinterpret_cast<const frag*>(&Bs[s * BN * 4 + s_slot(r, ch)]);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) SCH::mac(av[i], bv[j], acc[i][j]);
        }
    };

    esum = exp_settle<SCH>(esum);
    // steady state: tile kt is in LDS, tile kt+1 travels global -> registers behind the MFMAs of tile kt;
    // the last tile is peeled so that the loop body is branch free (keeps the accumulators in AGPRs)
    for (int kt = kt_begin; kt + 1 < kt_end; ++kt) {
        load_tile(kt + 1);
        compute_tile();
        __syncthreads();                        // every wave is done reading tile kt
        store_tile();
        __syncthreads();
    }
    if (kt_begin < kt_end) compute_tile();
    S_MFMA_DRAIN();
    gemm_epilogue<SCH, FM, FN, 2, BN>(p, acc, m0 + wm * WM, n0 + wn * WN, z, bz, lane, tm, wm, wn * WN, smem4, blockIdx.x == 0 && blockIdx.y == 0, esum);
    }
};

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant for the large layers: BM x BN x 32 block tile, 8 waves (WGM x WGN), NSLOT LDS buffers filled by
// buffer_load_dwordx4 ... lds -- no staging registers, no ds_write pass; the loads of the next tile(s) stay in flight
// across the barriers while the current tile is multiplied (counted s_waitcnt vmcnt, raw s_barrier).
// The DMA writes M0 + lane*16, i.e. a lane-linear image: one instruction = 16 rows x 64 B of one part, and the XOR
// swizzle of s_slot is applied on the SOURCE side (lane l of a 16-row group fetches channel group
// (l&3) ^ ((l>>4)&3)).  Padded / out-of-range rows fetch the zero tail of the split buffer.
//   NSLOT == 2: wait(tile it) | barrier | multiply slot it&1 | barrier | issue tile it+2 into the slot just read
//   NSLOT == 3: wait(tile it) | barrier | issue tile it+2 into the slot read in iteration it-1 | multiply slot it%3
//               (one barrier per k-tile: it proves both "tile it has landed for every wave" and "every wave has
//                finished reading the slot that is about to be overwritten")
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    // lgkmcnt(0): this wave's LDS reads of the previous tile have returned (free: the MFMAs consumed them) -- the WAR
    // condition for re-staging that slot right after the barrier (3-slot ring)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// -DSEMSEG_STAMPS (tools/probes/gemm_phase_stamps.py builds such a library beside the product one): wave 0 of every block of
// igemm_dma_kernel leaves four wall-clock stamps -- entry, first tile landed, k loop done, epilogue done -- in a device table
#ifdef SEMSEG_STAMPS
__device__ unsigned long long semseg_dbg_stamps[4 * 16384];
#define SEMSEG_STAMP(i)                                                                                                          \
    do {                                                                                                                          \
        if (threadIdx.x == 0)                                                                                                     \
            semseg_dbg_stamps[((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 16383) * 4 + (i)] = wall_clock64(); \
    } while (0)
extern "C" int semseg_debug_stamps(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(semseg_dbg_stamps), n * sizeof(unsigned long long));
}
#else
#define SEMSEG_STAMP(i)
#endif

template <class SCH, int BM, int BN, int WGM, int WGN, int NSLOT>
struct igemm_dma_kernel_body {
    static constexpr int THREADS = WGM * WGN * 64;
    // the many-problem form only where small problems meet (HRNetV2 branches): h2 scheme, tiles up to 128 x 128
    static constexpr bool MULTI = std::is_same<SCH, SchH2>::value && BM <= 128 && BN <= 128;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const SParams p) {
    constexpr int NP = SCH::NP;
    SEMSEG_STAMP(0);
    typedef typename SCH::frag frag;
    // waves: 8 (4x2 / 2x4) for the round-1 tiles; 4 (2x2: 128x128 / 128x64 per wave -- a third less LDS read traffic per MFMA, one
    // wave per SIMD with up to 512 registers) and 16 (4x4: 64x64 per wave, four waves per SIMD to hide the waits) since round 2
    constexpr int NW = WGM * WGN;
    static_assert(NW == 4 || NW == 8 || NW == 16, "wave grid");
    constexpr int WM = BM / WGM, WN = BN / WGN;        // wave tile
    constexpr int FM = WM / 32, FN = WN / 32;
    constexpr int AG = BM / 16 / NW;                   // 16-row groups of A per wave
    constexpr int BG = BN / 16 / NW;
    constexpr int LPT = NP * (AG + BG);                // DMA instructions per wave per tile
    constexpr int A_BYTES = NP * BM * 64, B_BYTES = NP * BN * 64, BUF_BYTES = A_BYTES + B_BYTES;
    static_assert(AG >= 1 && BG >= 1 && FM >= 1 && FN >= 1, "tile");
    // 12 ... 15: 2 ... 5 slots, software-pipelined fragment reads; 25 (round 5): ring of FIVE HALF tiles -- an entry is the A rows or
    // the B rows of a k-tile, as in igemm_dma64_kernel -- with the DMA pieces spread behind the MFMA groups
    // 33 ... 35 (round 5): the 3 ... 5-slot rings with the pieces of a k-tile issued in two halves behind different MFMA groups
    static_assert(NSLOT == 2 || NSLOT == 3 || (NSLOT >= 12 && NSLOT <= 15) || NSLOT == 25 || (NSLOT >= 33 && NSLOT <= 35), "slots");
    constexpr int RING = NSLOT == 25 ? 3 : (NSLOT >= 33 ? NSLOT - 30 : (NSLOT >= 12 ? NSLOT - 10 : NSLOT));
    static_assert((RING > 3 ? RING - 1 : 3) * LPT < 64, "vmcnt range");
    static_assert(NSLOT != 25 || (BM == BN && NP == 2), "the half-tile ring takes square tiles of the h2 scheme");

    extern __shared__ __align__(16) uint4 smem4[];
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    int esum = 0;              // sum of the operands' exponent words: exp_fetch behind the prologue's DMA pieces

    // block -> (tm, tn, z, batch): the blocks of one XCD (consecutive ids after xcd_remap) walk tm fastest, so they share ONE
    // weight slice (tn, z) -- streamed once into that XCD's L2 -- and read adjacent pixel tiles (shared halo rows); the
    // batches of a batched GEMM (Winograd positions) are dealt to the XCDs WHOLE, so both operands of a batch stay in one L2
    const GemmBlock gb = gemm_block(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), p.tiles_m, p.tiles_n, p.splits,
                                    gridDim.z, p.tn_fast);      // block_order.h
    const int tm = gb.tm, tn = gb.tn, z = gb.z, bz = gb.batch;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kt_begin = z * p.kt_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.kt_per_split);
    const int nk = kt_end - kt_begin;

    const uint32_t a_zero = 2u * NP * p.in_plane, b_zero = 2u * NP * p.w_plane;      // byte offsets of the zero tails
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)(a_zero + SPLIT_ZERO_TAIL_BYTES), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, (int)(b_zero + SPLIT_ZERO_TAIL_BYTES), 0x00020000);
    const uint32_t a_plane_b = 2u * p.in_plane, b_plane_b = 2u * p.w_plane;

    // this lane's row inside a 16-row group and its source channel group (pre-swizzled)
    const int lrow = lane >> 2;
    const int q = (lane & 3) ^ ((lane >> 4) & 3);

    int a_ih0[AG], a_iw0[AG], a_base[AG];
    const PixDiv pd = pix_div(p.Hout, p.Wout, p.M);
#pragma unroll
    for (int i = 0; i < AG; ++i) {
        const int m = m0 + (wave + NW * i) * 16 + lrow;
        if (m < p.M) {
            int n, oh, ow;
            pix_of(pd, m, n, oh, ow);
            a_ih0[i] = oh * p.a + p.off;
            a_iw0[i] = ow * p.a + p.off;
            a_base[i] = n * p.Hin * p.Win + bz * p.batch_in_rows;
        } else {
            a_ih0[i] = -(1 << 28);
            a_iw0[i] = -(1 << 28);
            a_base[i] = 0;
        }
    }
    // Per-lane DMA source state.  Every load is issued by all 64 lanes with a plain VGPR offset -- no per-lane branch,
    // so each wave issues exactly LPT VMEM instructions per tile (the vmcnt accounting below depends on it):
    //   src = byte offset of (row, tap, channel group q) in plane 0, or the zero tail for padded / out-of-range rows
    //   msk = all ones for live rows, 0 for dead rows (turns the per-tile / per-plane increments off)
    uint32_t b_src[BG], b_msk[BG];
    const uint32_t w_row_b = 2u * (uint32_t)p.T * p.pitch;
#pragma unroll
    for (int i = 0; i < BG; ++i) {
        const int n = n0 + (wave + NW * i) * 16 + lrow;
        const bool ok = n < p.Cout;
        b_src[i] = ok ? (uint32_t)(n + bz * p.batch_w_rows) * w_row_b + 16u * q : b_zero;
        b_msk[i] = ok ? 0xffffffffu : 0u;
    }
    uint32_t a_src[AG], a_msk[AG];
    KWalk kw;
    kw.init(p, kt_begin);
    const bool fast_tap = (p.div == 1) && (p.T <= 32);          // block-uniform, see igemm_rs_kernel
    uint32_t a_row[AG], a_vm[AG];
    if (fast_tap) {
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            a_row[i] = 2u * (((uint32_t)a_base[i] + (uint32_t)a_ih0[i] * (uint32_t)p.Win + (uint32_t)a_iw0[i]) *
                             (uint32_t)p.pitch) + 16u * q;       // arithmetic mod 2^32 (halo rows are "negative")
            uint32_t vm = 0;
            int r = 0, s2 = 0;
            for (int t = 0; t < p.T; ++t) {
                const int nh = a_ih0[i] + r * p.step, nw = a_iw0[i] + s2 * p.step;
                vm |= ((nh >= 0) & (nw >= 0) & (nh < p.Hin) & (nw < p.Win)) ? (1u << t) : 0u;
                if (++s2 == p.S) { s2 = 0; ++r; }
            }
            a_vm[i] = vm;
        }
    }
    auto set_tap = [&](int t, int r, int s) {
        if (fast_tap) {
            const uint32_t delta = 2u * ((uint32_t)((r * p.Win + s) * p.step) * (uint32_t)p.pitch);
#pragma unroll
            for (int i = 0; i < AG; ++i) {
                const bool ok = (a_vm[i] >> t) & 1u;
                a_src[i] = ok ? a_row[i] + delta : a_zero;
                a_msk[i] = ok ? 0xffffffffu : 0u;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            int nh = a_ih0[i] + r * p.step;
            int nw = a_iw0[i] + s * p.step;
            bool ok = (nh >= 0) & (nw >= 0);
            if (p.div > 1) {
                ok = ok & ((nh % p.div) == 0) & ((nw % p.div) == 0);
                nh /= p.div;
                nw /= p.div;
            }
            ok = ok & (nh < p.Hin) & (nw < p.Win);
            a_src[i] = ok ? 2u * ((uint32_t)(a_base[i] + nh * p.Win + nw) * (uint32_t)p.pitch) + 16u * q : a_zero;
            a_msk[i] = ok ? 0xffffffffu : 0u;
        }
    };

    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem + (uint32_t)wave * 1024u;   // this wave's first 16-row group
    // issue the DMA of k-tile `kt` into LDS slot `slot` (every lane fetches the zero tail when kt is out of range)
    auto issue = [&](int kt, int slot) {
#ifdef SEMSEG_PROBE_NODMA
        return;                                  // timing probe (tools/probes/gemm_phase_stamps.py): the loop without its DMA
#endif
        const uint32_t abuf = lds0 + (uint32_t)slot * BUF_BYTES;
        const uint32_t bbuf = abuf + A_BYTES;
        if (kt < kt_end) {                  // wave-uniform; tiles are issued in k order: the counters advance here
            const int c0 = kw.cc * 32;
            if (kw.dirty) {                 // wave-uniform
                set_tap(kw.t, kw.r, kw.s);
                kw.dirty = false;
            }
            const uint32_t ka_b = 2u * (uint32_t)c0;
            const uint32_t kb_b = 2u * ((uint32_t)kw.t * p.pitch + c0);
            kw.advance(p);
#pragma unroll
            for (int i = 0; i < AG; ++i)
#pragma unroll
                for (int s = 0; s < NP; ++s) {
                    const uint32_t vo = a_src[i] + ((ka_b + s * a_plane_b) & a_msk[i]);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(uintptr_t)(abuf + (s * BM + NW * i * 16) * 64), 16, vo,
                                                             0, 0, 0);
                }
#pragma unroll
            for (int i = 0; i < BG; ++i)
#pragma unroll
                for (int s = 0; s < NP; ++s) {
                    const uint32_t vo = b_src[i] + ((kb_b + s * b_plane_b) & b_msk[i]);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(uintptr_t)(bbuf + (s * BN + NW * i * 16) * 64), 16, vo,
                                                             0, 0, 0);
                }
        } else {
#pragma unroll
            for (int i = 0; i < AG; ++i)
#pragma unroll
                for (int s = 0; s < NP; ++s)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(uintptr_t)(abuf + (s * BM + NW * i * 16) * 64), 16,
                                                             a_zero, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < BG; ++i)
#pragma unroll
                for (int s = 0; s < NP; ++s)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(uintptr_t)(bbuf + (s * BN + NW * i * 16) * 64), 16,
                                                             b_zero, 0, 0, 0);
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 31;
    const int kb = lane >> 5;

    // fragments of k-step `ks` (16 of the 32 channels of the k-tile) of the tile in `slot`
    auto read_frags = [&](int slot, int ks, frag (&av)[FM][NP], frag (&bv)[FN][NP]) {
#ifdef SEMSEG_PROBE_NOREAD
        return;                                  // timing probe: no fragment reads
#endif
        const uint4* As = reinterpret_cast<const uint4*>(smem + slot * BUF_BYTES);
        const uint4* Bs = reinterpret_cast<const uint4*>(smem + slot * BUF_BYTES + A_BYTES);
        const int ch = 2 * ks + kb;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int r = wn * WN + j * 32 + frow;
#pragma unroll
            for (int s = 0; s < NP; ++s) bv[j][s] = *reinterpret_cast<const frag*>(&Bs[s * BN * 4 + s_slot(r, ch)]);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int r = wm * WM + i * 32 + frow;
#pragma unroll
            for (int s = 0; s < NP; ++s) av[i][s] = *reinterpret_cast<const frag*>(&As[s * BM * 4 + s_slot(r, ch)]);
        }
    };
    auto mma = [&](const frag (&av)[FM][NP], const frag (&bv)[FN][NP]) {
#ifdef SEMSEG_PROBE_NOMMA                                 // timing probe: the fragments are read and dropped
#ifndef SEMSEG_PROBE_NOREAD
#pragma unroll
        for (int s = 0; s < NP; ++s) {
#pragma unroll
            for (int i = 0; i < FM; ++i) asm volatile("" ::"v"(av[i][s]));
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" ::"v"(bv[j][s]));
        }
#endif
        return;
#endif
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) SCH::mac(av[i], bv[j], acc[i][j]);
    };
    auto compute_tile = [&](int slot) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag av[FM][NP], bv[FN][NP];
            read_frags(slot, ks, av, bv);
            mma(av, bv);
        }
    };

    if constexpr (NSLOT == 25) {
        // ---- ring of five half tiles (round 5) --------------------------------------------------------------------------------------
        // The 256 x 256 block keeps two planes per operand: a 32-deep k-tile is 64 KiB, two whole slots are all the LDS holds, so
        // tile it+2 could only be issued -- as one burst of LPT pieces per wave -- after the barrier that ends tile it, ONE tile
        // before it is needed.  Entries of HALF a tile (the A rows, then the B rows: A_BYTES each) make the ring five deep in the
        // same 160 KiB: tile it is multiplied from two slots, three entries (1.5 tiles) are in flight, and the two entries that
        // refill the slots of tile it are issued behind DIFFERENT MFMA groups (the B rows of tile it+2 behind the second group of
        // tile it, the A rows of tile it+3 behind the first group of tile it+1), not together right after the barrier
        // (profiles/r6_*: the same change on the 64-deep kernels).  Same fragments, same MFMA order: bit-identical results.
        constexpr int PPE = NP * AG;                       // pieces per wave per entry (AG == BG)
        constexpr int NENT = 5;
        static_assert(3 * PPE < 64, "vmcnt range");
        int i_ent = 0, i_slot = 0;
        auto issue_entry = [&]() {
            const bool live = kt_begin + (i_ent >> 1) < kt_end;              // wave-uniform
            const bool is_b = i_ent & 1;
            const uint32_t base = lds0 + (uint32_t)i_slot * A_BYTES;
            if (!is_b) {
                uint32_t ka_b = 0;
                if (live) {
                    if (kw.dirty) {
                        set_tap(kw.t, kw.r, kw.s);
                        kw.dirty = false;
                    }
                    ka_b = 2u * 32u * (uint32_t)kw.cc;
                }
#pragma unroll
                for (int i = 0; i < AG; ++i)
#pragma unroll
                    for (int s2 = 0; s2 < NP; ++s2) {
                        const uint32_t vo = live ? a_src[i] + ((ka_b + s2 * a_plane_b) & a_msk[i]) : a_zero;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(uintptr_t)(base + (s2 * BM + NW * i * 16) * 64), 16, vo,
                                                                 0, 0, 0);
                    }
            } else {
                uint32_t kb_b = 0;
                if (live) {
                    kb_b = 2u * ((uint32_t)kw.t * p.pitch + 32u * (uint32_t)kw.cc);
                    kw.advance(p);                          // the tile's B entry closes it
                }
#pragma unroll
                for (int i = 0; i < BG; ++i)
#pragma unroll
                    for (int s2 = 0; s2 < NP; ++s2) {
                        const uint32_t vo = live ? b_src[i] + ((kb_b + s2 * b_plane_b) & b_msk[i]) : b_zero;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(uintptr_t)(base + (s2 * BN + NW * i * 16) * 64), 16, vo,
                                                                 0, 0, 0);
                    }
            }
            ++i_ent;
            i_slot = (i_slot == NENT - 1) ? 0 : i_slot + 1;
        };
        auto read_frags2 = [&](int sa, int sb, int ks, frag (&av)[FM][NP], frag (&bv)[FN][NP]) {
            const uint4* As = reinterpret_cast<const uint4*>(smem + sa * A_BYTES);
            const uint4* Bs = reinterpret_cast<const uint4*>(smem + sb * A_BYTES);
            const int ch = 2 * ks + kb;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int r = wn * WN + j * 32 + frow;
#pragma unroll
                for (int s2 = 0; s2 < NP; ++s2) bv[j][s2] = *reinterpret_cast<const frag*>(&Bs[s2 * BN * 4 + s_slot(r, ch)]);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int r = wm * WM + i * 32 + frow;
#pragma unroll
                for (int s2 = 0; s2 < NP; ++s2) av[i][s2] = *reinterpret_cast<const frag*>(&As[s2 * BM * 4 + s_slot(r, ch)]);
            }
        };
#pragma unroll
        for (int j = 0; j < NENT; ++j) issue_entry();        // A(0) B(0) A(1) B(1) A(2)
        frag a0[FM][NP], b0[FN][NP], a1[FM][NP], b1[FN][NP];
        esum = exp_fetch<SCH>(p.in_exp, p.w_exp);
        wait_vm_barrier<3 * PPE>();                           // entries 0 and 1 have landed for every wave
        int sa = 0, sb = 1;
        read_frags2(sa, sb, 0, a0, b0);
        for (int it = 0; it < nk; ++it) {
            read_frags2(sa, sb, 1, a1, b1);
            mma(a0, b0);
            if (it) issue_entry();                            // A rows of tile it + 2 into the second slot tile it - 1 has freed
            wait_vm_barrier<1 * PPE>();                       // my reads of tile it are done; tile it + 1 has landed (the A entry just issued may fly)
            const int na = (sb == NENT - 1) ? 0 : sb + 1;
            const int nb = (na == NENT - 1) ? 0 : na + 1;
            read_frags2(na, nb, 0, a0, b0);                   // past the last tile: zero tails, never multiplied
            mma(a1, b1);
            issue_entry();                                    // B rows of tile it + 2 into slot sa (i_slot == sa here)
            sa = na;
            sb = nb;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        S_MFMA_DRAIN();
        gemm_epilogue<SCH, FM, FN, WGM, BN>(p, acc, m0 + wm * WM, n0 + wn * WN, z, bz, lane, tm, wm, wn * WN, smem, blockIdx.x == 0 && blockIdx.y == 0, esum);
        return;
    }
    // Every iteration issues exactly LPT DMA instructions per wave (dummy zero-tail fetches past the end), so "all but
    // the newest LPT have landed" == "tile `it` has landed" at the top of every iteration.
    if constexpr (NSLOT >= 33) {
        // the ring of whole k-tiles (NSLOT 13 ... 15) with SPREAD issue: the A pieces of tile it + RING go out behind the second MFMA
        // group of tile it (right after the barrier that frees the slot), its B pieces behind the first group of tile it + 1 --
        // the waves of a block leave the barrier together, and LPT back-to-back DMA issues per wave were a window in which none of
        // them feeds the matrix pipe (measured on the 64-deep kernels: + 5 ... 10 %, profiles/r6_conv_ab_spread_dma_issue.txt)
        uint32_t pend_kb = 0, pend_base = 0;
        bool pend_live = false;
        auto issue_a = [&](int kt, int slot) {
            const uint32_t abuf = lds0 + (uint32_t)slot * BUF_BYTES;
            pend_base = abuf + A_BYTES;
            pend_live = kt < kt_end;                        // wave-uniform
            uint32_t ka_b = 0;
            if (pend_live) {
                const int c0 = kw.cc * 32;
                if (kw.dirty) {
                    set_tap(kw.t, kw.r, kw.s);
                    kw.dirty = false;
                }
                ka_b = 2u * (uint32_t)c0;
                pend_kb = 2u * ((uint32_t)kw.t * p.pitch + c0);
                kw.advance(p);
            }
#pragma unroll
            for (int i = 0; i < AG; ++i)
#pragma unroll
                for (int s2 = 0; s2 < NP; ++s2) {
                    const uint32_t vo = pend_live ? a_src[i] + ((ka_b + s2 * a_plane_b) & a_msk[i]) : a_zero;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(uintptr_t)(abuf + (s2 * BM + NW * i * 16) * 64), 16, vo,
                                                             0, 0, 0);
                }
        };
        auto issue_b = [&]() {
#pragma unroll
            for (int i = 0; i < BG; ++i)
#pragma unroll
                for (int s2 = 0; s2 < NP; ++s2) {
                    const uint32_t vo = pend_live ? b_src[i] + ((pend_kb + s2 * b_plane_b) & b_msk[i]) : b_zero;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(uintptr_t)(pend_base + (s2 * BN + NW * i * 16) * 64), 16, vo,
                                                             0, 0, 0);
                }
        };
#pragma unroll
        for (int t = 0; t < RING; ++t) {
            issue_a(kt_begin + t, t);
            issue_b();
        }
        frag a0[FM][NP], b0[FN][NP], a1[FM][NP], b1[FN][NP];
        esum = exp_fetch<SCH>(p.in_exp, p.w_exp);
        wait_vm_barrier<(RING - 1) * LPT>();                  // tile 0 has landed for every wave
        read_frags(0, 0, a0, b0);
        int slot = 0;
        bool half_open = false;
        for (int it = 0; it < nk; ++it) {
            const int next = (slot == RING - 1) ? 0 : slot + 1;
            read_frags(slot, 1, a1, b1);
            mma(a0, b0);
            if (half_open) issue_b();                         // closes tile it - 1 + RING
            wait_vm_barrier<(RING - 2) * LPT>();              // my reads of `slot` are done, tile it + 1 has landed
            read_frags(next, 0, a0, b0);
            mma(a1, b1);
            issue_a(kt_begin + it + RING, slot);
            half_open = true;
            slot = next;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        S_MFMA_DRAIN();
        gemm_epilogue<SCH, FM, FN, WGM, BN>(p, acc, m0 + wm * WM, n0 + wn * WN, z, bz, lane, tm, wm, wn * WN, smem, blockIdx.x == 0 && blockIdx.y == 0, esum);
        return;
    }
    issue(kt_begin, 0);
    issue(kt_begin + 1, 1);
    esum = exp_fetch<SCH>(p.in_exp, p.w_exp);                // every form below: behind the first two tiles' DMA pieces
    if constexpr (NSLOT >= 13) {
        // RING-slot ring (3 ... 5) with the same software pipeline: tile it+RING is issued into the slot of tile it right after the
        // barrier, RING - 1 k-tiles before it is needed (the 2-slot form leaves one).  Round 4 (slots 4 / 5): the short-reduction
        // launches are bound by the latency of one DMA round trip per k-tile, not by its bandwidth (the DMA stream of the fused
        // Winograd kernel alone: profiles/r5_winograd_dgrad_forms.txt), and the small tiles have the LDS to keep more in flight
#pragma unroll
        for (int t = 2; t < RING; ++t) issue(kt_begin + t, t);
        frag a0[FM][NP], b0[FN][NP], a1[FM][NP], b1[FN][NP];
        wait_vm_barrier<(RING - 1) * LPT>();                  // tile 0 has landed for every wave
        read_frags(0, 0, a0, b0);
        int slot = 0;
        for (int it = 0; it < nk; ++it) {
            const int next = (slot == RING - 1) ? 0 : slot + 1;
            read_frags(slot, 1, a1, b1);
            mma(a0, b0);
            wait_vm_barrier<(RING - 2) * LPT>();              // my reads of `slot` are done, tile it+1 has landed
            issue(kt_begin + it + RING, slot);
            read_frags(next, 0, a0, b0);
            mma(a1, b1);
            slot = next;
        }
    } else if constexpr (NSLOT == 12) {
        // Software-pipelined 2-slot loop, ONE barrier per k-tile: the first half of the next tile's fragments is read
        // from LDS behind the MFMAs of the second half of the current tile, so no wave ever sits in front of the MFMA pipe
        // waiting for its first ds_read after a barrier (with the plain loop all 8 waves hit the LDS at once there).
        //   it: read ks=1 of tile it | MMA ks=0 | wait: my reads done, tile it+1 landed | barrier |
        //       DMA tile it+2 -> the slot just read | read ks=0 of tile it+1 | MMA ks=1
        frag a0[FM][NP], b0[FN][NP], a1[FM][NP], b1[FN][NP];
        wait_vm_barrier<LPT>();                               // tile 0 has landed for every wave
        SEMSEG_STAMP(1);
        read_frags(0, 0, a0, b0);
        for (int it = 0; it < nk; ++it) {
            const int slot = it & 1;
            read_frags(slot, 1, a1, b1);
            mma(a0, b0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            issue(kt_begin + it + 2, slot);
            read_frags(slot ^ 1, 0, a0, b0);                  // past the last tile: zero-tail / stale data, never multiplied
            mma(a1, b1);
        }
    } else if constexpr (NSLOT == 2) {
        for (int it = 0; it < nk; ++it) {
            const int slot = it & 1;
            wait_vm_barrier<LPT>();
            compute_tile(slot);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done reading `slot`
            issue(kt_begin + it + 2, slot);
        }
    } else {
        int slot = 0, fill = 2;
        for (int it = 0; it < nk; ++it) {
            wait_vm_barrier<LPT>();
            issue(kt_begin + it + 2, fill);
            compute_tile(slot);
            slot = (slot == 2) ? 0 : slot + 1;
            fill = (fill == 2) ? 0 : fill + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    S_MFMA_DRAIN();
    SEMSEG_STAMP(2);
    gemm_epilogue<SCH, FM, FN, WGM, BN>(p, acc, m0 + wm * WM, n0 + wn * WN, z, bz, lane, tm, wm, wn * WN, smem, blockIdx.x == 0 && blockIdx.y == 0, esum);
    SEMSEG_STAMP(3);
    }
};

// ------------------------------------------------------------------------------------------------
// igemm_dma_kernel on 64-deep k-tiles (round 4; tiles 22 / 23).  What the 32-deep loop leaves on the table was measured on the fused
// Winograd kernel (profiles/r5_winograd_dgrad_forms.txt: the same block on 64-deep tiles runs 10 - 15 % faster): every DMA piece of
// the 32-deep form gathers 16 rows x 64 B -- half a cache line per row -- and every 32 channels cost a barrier and a restart of the
// fragment reads.  Here a k-tile is one tap x 64 channels: a piece is 8 rows x 128 B (whole lines), the LDS row is 128 B with chunk
// c of row r stored at c ^ ((r >> 1) & 7) (applied on the DMA source address; a ds_read_b128 lane group covers the 64 banks once),
// four k16 steps per barrier.  Two planes per operand make a whole 64-deep tile of a 128 x 128 block 64 KiB, so the ring holds HALF
// tiles -- entry j = the A rows (j even) or the B rows (j odd) of tile j / 2 -- five entries (160 KiB at 128 x 128: one block per
// CU; 80 KiB at 64 x 64: two), 1.5 tiles in flight behind the one being multiplied.  Needs BM == BN and a reduction that is whole
// 64-channel chunks (channels padded to 32 must be a multiple of 64); everything else -- tap algebra, split-K, batches, block order,
// epilogue incl. the BN statistics -- is igemm_dma_kernel's.
// ------------------------------------------------------------------------------------------------
// SPREAD (round 5, as wino_fused64_kernel's SCHED bit 0): the DMA pieces that refill a tile's two slots are issued in four groups
// behind the four MFMA groups that follow the barrier, not as one burst right after it.
template <class SCH, int BM, int BN, int WGM, int WGN, bool SPREAD = false>
struct igemm_dma64_kernel_body {
    static constexpr int THREADS = WGM * WGN * 64;
    // the many-problem form only where small problems meet (HRNetV2 branches): h2 scheme, tiles up to 128 x 128
    static constexpr bool MULTI = std::is_same<SCH, SchH2>::value && BM <= 128 && BN <= 128;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const SParams p) {
    constexpr int NP = SCH::NP, NENT = 5;
    typedef typename SCH::frag frag;
    constexpr int NW = WGM * WGN;
    static_assert(BM == BN && (NW == 4 || NW == 8 || NW == 16), "tile");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int FM = WM / 32, FN = WN / 32;
    constexpr int G8 = BM / 8 / NW;                    // 8-row groups per wave and plane, A and B alike
    constexpr int PPE = NP * G8;                       // DMA pieces per wave per ring entry
    constexpr int ENT_BYTES = NP * BM * 128;
    static_assert(G8 >= 1 && FM >= 1 && FN >= 1 && 3 * PPE < 64, "tile");

    extern __shared__ __align__(16) uint4 smem4[];
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem4);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    int esum = 0;              // sum of the operands' exponent words: exp_fetch behind the prologue's DMA pieces

    const GemmBlock gb = gemm_block(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), p.tiles_m, p.tiles_n, p.splits,
                                    gridDim.z, p.tn_fast);
    const int tm = gb.tm, tn = gb.tn, z = gb.z, bz = gb.batch;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kt_begin = z * p.kt_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.kt_per_split);
    const int nk = kt_end - kt_begin;

    const uint32_t a_zero = 2u * NP * p.in_plane, b_zero = 2u * NP * p.w_plane;
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)(a_zero + SPLIT_ZERO_TAIL_BYTES), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, (int)(b_zero + SPLIT_ZERO_TAIL_BYTES), 0x00020000);
    const uint32_t a_plane_b = 2u * p.in_plane, b_plane_b = 2u * p.w_plane;

    // DMA lane constants: row inside an 8-row piece; the physical chunk l & 7 of row r holds the logical chunk (l & 7) ^ ((r >> 1) & 7),
    // r = 8 g + (l >> 3), g = wave + NW i (NW even: the parity of g is the wave's)
    const int lrow = lane >> 3;
    const int q = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);

    int a_ih0[G8], a_iw0[G8], a_base[G8];
    const PixDiv pd = pix_div(p.Hout, p.Wout, p.M);
#pragma unroll
    for (int i = 0; i < G8; ++i) {
        const int m = m0 + (wave + NW * i) * 8 + lrow;
        if (m < p.M) {
            int n, oh, ow;
            pix_of(pd, m, n, oh, ow);
            a_ih0[i] = oh * p.a + p.off;
            a_iw0[i] = ow * p.a + p.off;
            a_base[i] = n * p.Hin * p.Win + bz * p.batch_in_rows;
        } else {
            a_ih0[i] = -(1 << 28);
            a_iw0[i] = -(1 << 28);
            a_base[i] = 0;
        }
    }
    uint32_t b_src[G8], b_msk[G8];
    const uint32_t w_row_b = 2u * (uint32_t)p.T * p.pitch;
#pragma unroll
    for (int i = 0; i < G8; ++i) {
        const int n = n0 + (wave + NW * i) * 8 + lrow;
        const bool ok = n < p.Cout;
        b_src[i] = ok ? (uint32_t)(n + bz * p.batch_w_rows) * w_row_b + 16u * q : b_zero;
        b_msk[i] = ok ? 0xffffffffu : 0u;
    }
    uint32_t a_src[G8], a_msk[G8];
    KWalk kw;
    kw.init(p, kt_begin);
    const bool fast_tap = (p.div == 1) && (p.T <= 32);
    uint32_t a_row[G8], a_vm[G8];
    if (fast_tap) {
#pragma unroll
        for (int i = 0; i < G8; ++i) {
            a_row[i] = 2u * (((uint32_t)a_base[i] + (uint32_t)a_ih0[i] * (uint32_t)p.Win + (uint32_t)a_iw0[i]) *
                             (uint32_t)p.pitch) + 16u * q;
            uint32_t vm = 0;
            int r = 0, s2 = 0;
            for (int t = 0; t < p.T; ++t) {
                const int nh = a_ih0[i] + r * p.step, nw = a_iw0[i] + s2 * p.step;
                vm |= ((nh >= 0) & (nw >= 0) & (nh < p.Hin) & (nw < p.Win)) ? (1u << t) : 0u;
                if (++s2 == p.S) { s2 = 0; ++r; }
            }
            a_vm[i] = vm;
        }
    }
    auto set_tap = [&](int t, int r, int s) {
        if (fast_tap) {
            const uint32_t delta = 2u * ((uint32_t)((r * p.Win + s) * p.step) * (uint32_t)p.pitch);
#pragma unroll
            for (int i = 0; i < G8; ++i) {
                const bool ok = (a_vm[i] >> t) & 1u;
                a_src[i] = ok ? a_row[i] + delta : a_zero;
                a_msk[i] = ok ? 0xffffffffu : 0u;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < G8; ++i) {
            int nh = a_ih0[i] + r * p.step;
            int nw = a_iw0[i] + s * p.step;
            bool ok = (nh >= 0) & (nw >= 0);
            if (p.div > 1) {
                ok = ok & ((nh % p.div) == 0) & ((nw % p.div) == 0);
                nh /= p.div;
                nw /= p.div;
            }
            ok = ok & (nh < p.Hin) & (nw < p.Win);
            a_src[i] = ok ? 2u * ((uint32_t)(a_base[i] + nh * p.Win + nw) * (uint32_t)p.pitch) + 16u * q : a_zero;
            a_msk[i] = ok ? 0xffffffffu : 0u;
        }
    };

    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
    int i_ent = 0, i_slot = 0;                         // ring entry to issue next (even: A rows of tile i_ent / 2, odd: its B rows)
    // one ring entry in two halves: half h = the row groups [h G8 / 2, (h + 1) G8 / 2) (all of them in half 0 when G8 == 1); the second
    // half closes the entry
    constexpr int HG = G8 >= 2 ? G8 / 2 : 1;
    uint32_t ent_k = 0;                                // byte offset of the open entry's k position (set by its first half)
    bool ent_live = false;
    // a reduction of an odd number of 32-channel chunks (96 channels: HRNetV2's second branch): the last 64-channel chunk of a tap
    // has only its lower half; the lanes that would fetch the upper half (logical 16-byte chunks 4 ... 7 of the 128-byte row) read
    // the zero tail instead -- for BOTH operands, so the products are 0 x 0 whatever lies behind the row
    const bool upper_half_lane = q >= 4;
    bool ent_tail = false;                             // wave-uniform: the open entry is such a last chunk
    auto issue_half = [&](int h) {
        const bool is_b = i_ent & 1;
        const uint32_t base = lds0 + (uint32_t)i_slot * ENT_BYTES;
        if (h == 0) {
            ent_live = kt_begin + (i_ent >> 1) < kt_end;                     // wave-uniform
            ent_k = 0;
            ent_tail = false;
            if (!is_b) {
                if (ent_live) {
                    if (kw.dirty) {
                        set_tap(kw.t, kw.r, kw.s);
                        kw.dirty = false;
                    }
                    ent_k = 2u * 64u * (uint32_t)kw.cc;
                    ent_tail = p.k64_tail && kw.cc == p.chunks - 1;
                }
            } else if (ent_live) {
                ent_k = 2u * ((uint32_t)kw.t * p.pitch + 64u * (uint32_t)kw.cc);
                ent_tail = p.k64_tail && kw.cc == p.chunks - 1;
                kw.advance(p);                          // the tile's B entry closes it
            }
        }
        const bool fetch = ent_live && !(ent_tail && upper_half_lane);
        if (!(G8 < 2 && h == 1)) {
#pragma unroll
            for (int ii = 0; ii < HG; ++ii) {
                const int i = (G8 >= 2 ? h * HG : 0) + ii;
#pragma unroll
                for (int s = 0; s < NP; ++s) {
                    if (!is_b) {
                        const uint32_t vo = fetch ? a_src[i] + ((ent_k + s * a_plane_b) & a_msk[i]) : a_zero;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(uintptr_t)(base + (s * BM + (wave + NW * i) * 8) * 128), 16,
                                                                 vo, 0, 0, 0);
                    } else {
                        const uint32_t vo = fetch ? b_src[i] + ((ent_k + s * b_plane_b) & b_msk[i]) : b_zero;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(uintptr_t)(base + (s * BN + (wave + NW * i) * 8) * 128), 16,
                                                                 vo, 0, 0, 0);
                    }
                }
            }
        }
        if (h == 1) {
            ++i_ent;
            i_slot = (i_slot == NENT - 1) ? 0 : i_slot + 1;
        }
    };
    auto issue_entry = [&]() {
        issue_half(0);
        issue_half(1);
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 31;
    const int kb2 = lane >> 5;
    auto read_frags = [&](int sa, int sb, int ks, frag (&av)[FM][NP], frag (&bv)[FN][NP]) {
        const unsigned char* As = smem + sa * ENT_BYTES;
        const unsigned char* Bs = smem + sb * ENT_BYTES;
        const int ch = 2 * ks + kb2;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int r = wn * WN + j * 32 + frow;
            const int off = r * 128 + ((ch ^ ((r >> 1) & 7)) << 4);
#pragma unroll
            for (int s = 0; s < NP; ++s) bv[j][s] = *reinterpret_cast<const frag*>(Bs + s * BN * 128 + off);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int r = wm * WM + i * 32 + frow;
            const int off = r * 128 + ((ch ^ ((r >> 1) & 7)) << 4);
#pragma unroll
            for (int s = 0; s < NP; ++s) av[i][s] = *reinterpret_cast<const frag*>(As + s * BM * 128 + off);
        }
    };
    auto mma = [&](const frag (&av)[FM][NP], const frag (&bv)[FN][NP]) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) SCH::mac(av[i], bv[j], acc[i][j]);
    };

#pragma unroll
    for (int j = 0; j < NENT; ++j) issue_entry();
    frag a0[FM][NP], b0[FN][NP], a1[FM][NP], b1[FN][NP];
    esum = exp_fetch<SCH>(p.in_exp, p.w_exp);
    wait_vm_barrier<3 * PPE>();                               // entries 0 and 1 have landed for every wave
    int sa = 0, sb = 1;
    read_frags(sa, sb, 0, a0, b0);
    for (int it = 0; it < nk; ++it) {
        read_frags(sa, sb, 1, a1, b1);
        mma(a0, b0);
        if (SPREAD && it) issue_half(1);                      // closes the B entry begun behind the last MFMA group of tile it - 1
        read_frags(sa, sb, 2, a0, b0);
        mma(a1, b1);
        if (SPREAD && it) issue_half(0);                      // the A entry into the other slot tile it - 1 has freed
        read_frags(sa, sb, 3, a1, b1);
        mma(a0, b0);
        if (SPREAD && it) issue_half(1);
        wait_vm_barrier<1 * PPE>();                           // my reads of this tile are done, the next tile's two entries have landed
        if (!SPREAD) {
            issue_entry();
            issue_entry();
        }
        const int na = (sb == NENT - 1) ? 0 : sb + 1;
        const int nb = (na == NENT - 1) ? 0 : na + 1;
        read_frags(na, nb, 0, a0, b0);                        // past the last tile: zero tails, never multiplied
        mma(a1, b1);
        if (SPREAD) issue_half(0);                            // first half of the B entry that refills slot sa
        sa = na;
        sb = nb;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    S_MFMA_DRAIN();
    gemm_epilogue<SCH, FM, FN, WGM, BN>(p, acc, m0 + wm * WM, n0 + wn * WN, z, bz, lane, tm, wm, wn * WN, smem, blockIdx.x == 0 && blockIdx.y == 0, esum);
    }
};

static int dma64_spread() {          // SEMSEG_DMA64_SPREAD=0: the burst form (A/B switch of the round-5 measurement)
    static const int v = [] { const char* e = getenv("SEMSEG_DMA64_SPREAD"); return e ? atoi(e) : 1; }();
    return v;
}

template <class SCH, int BM, int BN, int WGM, int WGN>
static int launch_dma64(const SParams& p, hipStream_t st) {
    constexpr size_t smem = (size_t)5 * SCH::NP * BM * 128;
    static_assert(smem <= 160 * 1024, "LDS");
    dim3 grid(p.tiles_m * p.tiles_n, p.splits, p.batches > 0 ? p.batches : 1);
    if (dma64_spread()) {
            SEMSEG_LAUNCH_BODY((igemm_dma64_kernel_body<SCH, BM, BN, WGM, WGN, true>), grid, smem, st, p);
    } else {
            SEMSEG_LAUNCH_BODY((igemm_dma64_kernel_body<SCH, BM, BN, WGM, WGN, false>), grid, smem, st, p);
    }
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

template <bool VEC>
struct split_gemm_reduce_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ partial, const float* __restrict__ bias,
                                         float* __restrict__ out, int out_ld,
                                         int M, int Cout, int splits) {
    const size_t total = (size_t)M * Cout;
    if (VEC) {          // Cout, out_ld multiples of 4 and 16-byte aligned bases (checked by the launcher)
        const int qpr = Cout >> 2;
        const size_t quads = total >> 2;
        const float4* p4 = reinterpret_cast<const float4*>(partial);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) {
            const int m = (int)(i / qpr);
            const int n = (int)(i - (size_t)m * qpr) << 2;
            // all loads of a group of 4 slabs are issued before the first add (the plain loop has ONE load in flight per
            // thread and is latency-bound); the additions keep the order z = 0, 1, 2, ...
            float4 s = p4[i];
            const float4 b = bias ? *reinterpret_cast<const float4*>(bias + n) : f4zero();
            for (int zz = 1; zz < splits; zz += 4) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (zz + u < splits) v[u] = p4[(size_t)(zz + u) * quads + i];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (zz + u < splits) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            }
            if (bias) { s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w; }
            *reinterpret_cast<float4*>(out + (size_t)m * out_ld + n) = s;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / Cout);
        const int n = (int)(i - (size_t)m * Cout);
        float s = partial[i];
        for (int zz = 1; zz < splits; ++zz) s += partial[(size_t)zz * total + i];
        if (bias) s += bias[n];
        out[(size_t)m * out_ld + n] = s;
    }
    }
};

struct SPlan {
    int tile, BM, BN;
    int tiles_m, tiles_n, chunks, ktiles, splits, kt_per_split;
};

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

static const int kTiles[27][2] = {{128, 128}, {128, 64}, {64, 64}, {256, 128}, {256, 128}, {256, 256}, {128, 128},
                                   {256, 128}, {256, 256}, {128, 128}, {256, 128}, {256, 256}, {256, 128}, {256, 128},
                                   {256, 256}, {64, 64}, {128, 64}, {64, 64}, {128, 128},
                                   {64, 64}, {128, 128}, {128, 128},      // 19 ... 21: deep rings (round 4)
                                   {128, 128}, {64, 64}, {128, 128},      // 22 ... 24: 64-deep k-tiles (igemm_dma64_kernel)
                                   {256, 256}, {256, 256}};               // 25 / 26: 256 x 256 on the ring of five half tiles (16 / 8 waves)
static inline bool tile_is_k64(int tile) { return tile >= 22 && tile <= 24; }

constexpr int kMaxEpilogueParts = 512;      // partial rows (= block row tiles) the BN finish kernel is asked to reduce; beyond
                                            // that (the 256 x 256 maps of the stem on small tiles) the separate sweep is cheaper

struct EpilogueStats {      // optional request of run_gemm's caller
    void* buf;              // room for parts x 2 Cout x (double + float)
    size_t bytes;
    float* zero_word;
    int parts;              // out: partial rows written; 0 = not gathered (split-K plan, too many parts, buffer too small)
};

// Launch plan: same wave-quantisation model as plan_igemm (conv_igemm.hip) -- tile x split-K candidates, cost =
// waves * (k-tiles per block * tile cost + fixed) + split-K slab traffic.  The LDS-DMA tiles (3..5) are chosen by the
// tuner / overrides only.  Tuning overrides for tools/conv_bench.py: SEMSEG_S3_TILE=0..5, SEMSEG_S3_SPLITK=n
static SPlan plan_gemm(int sch, int M, int Cout, int Cp, int T, int ov_tile = -1, int ov_split = 0) {
    SPlan pl;
    pl.chunks = Cp / 32;
    pl.ktiles = T * pl.chunks;
    const double tile_cost[3] = {1.0, 0.52, 0.36};  // fitted to the MI355X sweep (profiles/r1c_conv_bench_s3_sweep.txt)
    const int slots[3] = {512, 768, 1280};          // resident blocks: 2 / 3 / 5 per CU
    double best = 1e30;
    int best_t = 2, best_s = 1;
    for (int t = 0; t < 3; ++t) {
        if (Cout <= 64 && kTiles[t][1] > 64) continue;
        const long tiles = (long)ceil_div(M, kTiles[t][0]) * ceil_div(Cout, kTiles[t][1]);
        for (int sp = 1; sp <= 16; ++sp) {
            if (sp > 1 && pl.ktiles / sp < 8) break;
            const int kps = ceil_div(pl.ktiles, sp);
            if (ceil_div(pl.ktiles, kps) != sp) continue;
            const long blocks = tiles * sp;
            const long rem = blocks % slots[t];
            const int per_cu = slots[t] / 256;
            const double frac = rem ? (double)min((long)per_cu, (rem + 255) / 256) / per_cu : 0.0;
            const double waves = (double)(blocks / slots[t]) + (rem ? 0.35 + 0.65 * frac : 0.0);
            double time = waves * (kps * tile_cost[t] + 6.0 * tile_cost[t]);
            // slab write + reduce, in units of one 128x128x32 k-tile on a full chip (~1.6 us)
            if (sp > 1) time += (double)(sp + 1) * M * Cout * 4.0 / 4.0e12 / 1.6e-6;
            if (time < best) { best = time; best_t = t; best_s = sp; }
        }
    }
    const int force_tile = ov_tile >= 0 ? ov_tile : env_int("SEMSEG_S3_TILE", -1);
    if (force_tile >= 0 && force_tile <= max_tile(sch, 0)) { best_t = force_tile; best_s = 1; }
    const int force_split = ov_split > 0 ? ov_split : env_int("SEMSEG_S3_SPLITK", 0);
    if (force_split > 0) best_s = min(force_split, pl.ktiles);
    pl.tile = best_t;
    pl.BM = kTiles[best_t][0];
    pl.BN = kTiles[best_t][1];
    pl.tiles_m = ceil_div(M, pl.BM);
    pl.tiles_n = ceil_div(Cout, pl.BN);
    pl.kt_per_split = ceil_div(pl.ktiles, best_s);
    pl.splits = ceil_div(pl.ktiles, pl.kt_per_split);
    return pl;
}

template <class SCH, int BM, int BN>
static int launch_rs(const SParams& p, hipStream_t st) {
    constexpr size_t smem = (size_t)SCH::NP * (BM + BN) * 64;
    dim3 grid(p.tiles_m * p.tiles_n, p.splits, p.batches > 0 ? p.batches : 1);
    SEMSEG_LAUNCH_BODY((igemm_rs_kernel_body<SCH, BM, BN>), grid, smem, st, p);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

template <class SCH, int BM, int BN, int WGM, int WGN, int NSLOT>
static int launch_dma(const SParams& p, hipStream_t st) {
    constexpr size_t smem = NSLOT == 25 ? (size_t)5 * SCH::NP * BM * 64      // five half tiles
                                        : (size_t)(NSLOT >= 33 ? NSLOT - 30 : (NSLOT >= 12 ? NSLOT - 10 : NSLOT)) * SCH::NP * (BM + BN) * 64;
    static_assert(smem <= 160 * 1024, "LDS");
    dim3 grid(p.tiles_m * p.tiles_n, p.splits, p.batches > 0 ? p.batches : 1);
    SEMSEG_LAUNCH_BODY((igemm_dma_kernel_body<SCH, BM, BN, WGM, WGN, NSLOT>), grid, smem, st, p);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// the ring forms 13 ... 15 with spread DMA issue (NSLOT + 20) unless SEMSEG_DMA64_SPREAD=0
template <class SCH, int BM, int BN, int WGM, int WGN, int NSLOT>
static int launch_dma_ring(const SParams& p, hipStream_t st) {
    static_assert(NSLOT >= 13 && NSLOT <= 15, "ring form");
    return dma64_spread() ? launch_dma<SCH, BM, BN, WGM, WGN, NSLOT + 20>(p, st) : launch_dma<SCH, BM, BN, WGM, WGN, NSLOT>(p, st);
}


template <class SCH>
static int run_gemm(SParams p, size_t in_rows, int ov_tile, int ov_split, void* workspace, size_t workspace_bytes,
                    hipStream_t st, EpilogueStats* es = nullptr) {
    const size_t in_plane = in_rows * p.pitch, w_plane = (size_t)p.Cout * p.T * p.pitch;
    if (2 * SCH::NP * in_plane + SPLIT_ZERO_TAIL_BYTES >= ((size_t)1 << 31) ||
        2 * SCH::NP * w_plane + SPLIT_ZERO_TAIL_BYTES >= ((size_t)1 << 31))
        return SEMSEG_EINVAL;      // 32-bit byte offsets (buffer descriptors of the LDS-DMA kernel)
    p.in_plane = (uint32_t)in_plane;
    p.w_plane = (uint32_t)w_plane;
    SPlan pl = plan_gemm(SCH::ID, p.M, p.Cout, p.Cp, p.T, ov_tile, ov_split);
    if (tile_is_k64(pl.tile)) {
        // 64-deep k-tiles: the reduction is counted in 64-channel chunks; the split of the plan is kept (never more slabs than planned)
        if (p.batches > 1 && (p.Cp % 64)) return SEMSEG_EINVAL;     // the batched Winograd GEMM keeps whole chunks
        p.k64_tail = (p.Cp % 64) ? 1 : 0;
        pl.chunks = (p.Cp + 63) / 64;
        pl.ktiles = p.T * pl.chunks;
        pl.kt_per_split = ceil_div(pl.ktiles, pl.splits);
        pl.splits = ceil_div(pl.ktiles, pl.kt_per_split);
    }
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
    // Block order inside an XCD.  Row tiles fastest (default): the blocks of an XCD share one weight slice and the halo rows of
    // neighbouring pixel tiles, and every pixel row is fetched by as many XCDs as there are column tiles.  For a 1x1 convolution
    // the weights are the small operand (C x K against M x C) and there is no halo: column tiles fastest fetches the pixel rows
    // ONCE and streams the whole weight matrix through every L2 instead (profiles/r4_pmc_step_traffic_cfg1.txt: layer3's
    // 1024 -> 256 conv fetched its input 3.9 times, the 256 -> 1024 one 7.2 times).
    static const int order_env = [] { const char* e = getenv("SEMSEG_IGEMM_ORDER"); return e ? atoi(e) : -1; }();   // tools only
    p.tn_fast = order_env >= 0 ? (order_env != 0) : (p.T == 1 && p.batches <= 1 && pl.tiles_n > 1);
    p.st_sum = nullptr; p.st_mm = nullptr; p.st_zero = nullptr;
    if (es) {
        es->parts = 0;
        const int parts = pl.tiles_m;          // one partial row per block row tile (the waves combine through LDS)
        const size_t need = (size_t)parts * 2 * p.Cout * (sizeof(double) + sizeof(float));
        if (pl.splits == 1 && p.batches <= 1 && parts <= kMaxEpilogueParts && es->buf && es->bytes >= need && aligned16(es->buf)) {
            p.st_sum = (double*)es->buf;
            p.st_mm = reinterpret_cast<float*>(p.st_sum + (size_t)parts * 2 * p.Cout);
            p.st_zero = es->zero_word;
            es->parts = parts;
        }
    }
    int rc = SEMSEG_EINVAL;
    semseg_batch::hint_cost(p.kt_per_split);       // a recorded launch: blocks with the longest k loop go first in a side-by-side launch
    switch (pl.tile) {
        case 0: rc = launch_rs<SCH, 128, 128>(p, st); break;
        case 1: rc = launch_rs<SCH, 128, 64>(p, st); break;
        case 2: rc = launch_rs<SCH, 64, 64>(p, st); break;
        case 3: rc = launch_dma<SCH, 256, 128, 4, 2, 2>(p, st); break;
        case 4:
            if constexpr (SCH::NP == 2) rc = launch_dma<SCH, 256, 128, 4, 2, 3>(p, st);
            break;
        case 5:
            if constexpr (SCH::NP == 2) rc = launch_dma<SCH, 256, 256, 2, 4, 2>(p, st);
            break;
        case 6:
            if constexpr (SCH::NP == 2) rc = launch_dma<SCH, 128, 128, 4, 2, 2>(p, st);
            break;
        case 7:
            if constexpr (SCH::NP == 2) rc = launch_dma<SCH, 256, 128, 4, 2, 12>(p, st);
            break;
        case 8:
            if constexpr (SCH::NP == 2) rc = launch_dma<SCH, 256, 256, 2, 4, 12>(p, st);
            break;
        case 9:
            if constexpr (SCH::NP == 2) rc = launch_dma<SCH, 128, 128, 4, 2, 12>(p, st);
            break;
        case 10:
            if constexpr (SCH::NP == 2) rc = launch_dma_ring<SCH, 256, 128, 4, 2, 13>(p, st);
            break;
        case 11:
            if constexpr (SCH::NP == 2) rc = launch_dma<SCH, 256, 256, 2, 2, 12>(p, st);      // 4 waves, 128x128 per wave
            break;
        case 12:
            if constexpr (SCH::NP == 2) rc = launch_dma<SCH, 256, 128, 2, 2, 12>(p, st);      // 4 waves, 128x64 per wave
            break;
        case 13:
            if constexpr (SCH::NP == 2) rc = launch_dma_ring<SCH, 256, 128, 2, 2, 13>(p, st);
            break;
        case 14:
            if constexpr (SCH::NP == 2) rc = launch_dma<SCH, 256, 256, 4, 4, 12>(p, st);      // 16 waves, 64x64 per wave
            break;
        // small tiles on the LDS-DMA ring (4 waves): the register-staged igemm_rs_kernel keeps ONE k-tile in flight and pays two
        // barriers per tile -- on the short-M layers (HRNet's 192 / 384-channel branches, layer1 / layer2) a 64x64 block multiplies
        // for 0.1 us per k-tile and waits ~1 us for the next; the ring keeps two tiles in flight behind one barrier
        case 15:
            if constexpr (SCH::NP == 2) rc = launch_dma_ring<SCH, 64, 64, 2, 2, 13>(p, st);
            break;
        case 16:
            if constexpr (SCH::NP == 2) rc = launch_dma_ring<SCH, 128, 64, 2, 2, 13>(p, st);
            break;
        case 17:
            if constexpr (SCH::NP == 2) rc = launch_dma<SCH, 64, 64, 2, 2, 12>(p, st);
            break;
        case 18:
            if constexpr (SCH::NP == 2) rc = launch_dma_ring<SCH, 128, 128, 2, 2, 13>(p, st);      // 4 waves, 64x64 per wave, 3-slot ring
            break;
        // deep rings on the small tiles (round 4): 4 / 5 k-tiles in flight behind one barrier per tile.  Sweep of the net's layers
        // (profiles/r5_conv_sweep_deep_rings.txt): +3 ... 5 % on layer3's dilated 3x3 convs and two of layer4's 1x1 convs, neutral on
        // HRNet's branches, worse wherever the shallower ring lets two blocks share a CU; the 128x64 tile on five slots and the
        // weight-gradient tiles on 4 / 5 slots won no layer and were dropped again.
        case 19:
            if constexpr (SCH::NP == 2) rc = launch_dma_ring<SCH, 64, 64, 2, 2, 15>(p, st);        // 5 x 16 KiB: two blocks per CU
            break;
        case 20:
            if constexpr (SCH::NP == 2) rc = launch_dma_ring<SCH, 128, 128, 4, 2, 14>(p, st);      // 8 waves, 4 x 32 KiB
            break;
        case 21:
            if constexpr (SCH::NP == 2) rc = launch_dma_ring<SCH, 128, 128, 4, 2, 15>(p, st);      // 8 waves, 5 x 32 KiB
            break;
        // 64-deep k-tiles, whole 128-byte lines per DMA piece, ring of five half tiles (igemm_dma64_kernel)
        case 22:
            if constexpr (SCH::NP == 2) rc = launch_dma64<SCH, 128, 128, 4, 2>(p, st);        // 8 waves, 160 KiB: one block per CU
            break;
        case 23:
            if constexpr (SCH::NP == 2) rc = launch_dma64<SCH, 64, 64, 2, 2>(p, st);          // 4 waves, 80 KiB: two blocks per CU
            break;
        case 24:
            if constexpr (SCH::NP == 2) rc = launch_dma64<SCH, 128, 128, 4, 4>(p, st);        // 16 waves (32 x 32 each), one block per CU
            break;
        // 256 x 256 on the ring of five half tiles with spread DMA issue (round 5): tiles 14 / 5 with 1.5 tiles in flight
        case 25:
            if constexpr (SCH::NP == 2) rc = launch_dma<SCH, 256, 256, 4, 4, 25>(p, st);      // 16 waves, 64 x 64 per wave
            break;
        case 26:
            if constexpr (SCH::NP == 2) rc = launch_dma<SCH, 256, 256, 2, 4, 25>(p, st);      // 8 waves, 128 x 64 per wave
            break;
    }
    if (rc) return rc;
    if (pl.splits > 1) {
        const size_t total = (size_t)p.M * p.Cout;
        const bool vec = (p.Cout % 4 == 0) && (p.out_ld % 4 == 0) && aligned16(p.out) && aligned16(p.partial) &&
                         (!p.bias || aligned16(p.bias));
        const int blocks = (int)min((size_t)2048, ceil_div_sz(vec ? total / 4 : total, 256));
        if (vec)
            SEMSEG_LAUNCH_BODY((split_gemm_reduce_kernel_body<true>), dim3(blocks), 0, st, p.partial, p.bias, p.out, p.out_ld,
                               p.M, p.Cout, pl.splits);
        else
            SEMSEG_LAUNCH_BODY((split_gemm_reduce_kernel_body<false>), dim3(blocks), 0, st, p.partial, p.bias, p.out, p.out_ld,
                               p.M, p.Cout, pl.splits);
        SEMSEG_LAUNCH_CHECK();
    }
    return 0;
}

static size_t gemm_workspace_bytes(int sch, int M, int Cout, int Cin, int T, int ov_tile, int ov_split) {
    const SPlan pl = plan_gemm(sch, M, Cout, round_up32(Cin), T, ov_tile, ov_split);
    return pl.splits > 1 ? (size_t)pl.splits * M * Cout * sizeof(float) : 0;
}

static inline int out_dim(int in, int k, int stride, int pad, int dil) {
    return (in + 2 * pad - dil * (k - 1) - 1) / stride + 1;
}

template <class SCH>
static int conv_fwd(const void* xs, const void* ws, const float* bias, float* y, int y_ld,
                    int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                    void* workspace, size_t workspace_bytes, void* stream, EpilogueStats* es = nullptr) {
    if (!xs || !ws || !y || N <= 0 || C <= 0 || K <= 0 || stride <= 0 || dil <= 0 || y_ld < K) return SEMSEG_EINVAL;
    if (!aligned16(xs) || !aligned16(ws)) return SEMSEG_EINVAL;
    const int OH = out_dim(H, R, stride, pad, dil), OW = out_dim(W, S, stride, pad, dil);
    if (OH <= 0 || OW <= 0) return SEMSEG_EINVAL;
    SParams p = {};
    p.in = (const uint16_t*)xs; p.wgt = (const uint16_t*)ws; p.bias = bias; p.out = y;
    p.Cp = round_up32(C); p.pitch = split_pitch(C); p.out_ld = y_ld;
    p.Hin = H; p.Win = W;
    p.Hout = OH; p.Wout = OW; p.Cout = K;
    p.M = N * OH * OW;
    p.S = S; p.T = R * S;
    p.a = stride; p.off = -pad; p.step = dil; p.div = 1;
    if (SCH::SCALED) {
        p.in_exp = h2_exp_ptr(xs, (size_t)N * H * W, C);
        p.w_exp = h2_exp_ptr(ws, (size_t)K * R * S, C);
    }
    int ov_tile = -1, ov_split = 0;
    lookup_plan(SCH::ID, 0, N, H, W, C, K, R, S, stride, pad, dil, &ov_tile, &ov_split);
    return run_gemm<SCH>(p, (size_t)N * H * W, ov_tile, ov_split, workspace, workspace_bytes, (hipStream_t)stream, es);
}

template <class SCH>
static int conv_dgrad(const void* dys, const void* wts, float* dx, int dx_ld,
                      int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!dys || !wts || !dx || N <= 0 || C <= 0 || K <= 0 || stride <= 0 || dil <= 0 || dx_ld < C) return SEMSEG_EINVAL;
    if (!aligned16(dys) || !aligned16(wts)) return SEMSEG_EINVAL;
    const int OH = out_dim(H, R, stride, pad, dil), OW = out_dim(W, S, stride, pad, dil);
    if (OH <= 0 || OW <= 0) return SEMSEG_EINVAL;
    SParams p = {};
    p.in = (const uint16_t*)dys; p.wgt = (const uint16_t*)wts; p.bias = nullptr; p.out = dx;
    p.Cp = round_up32(K); p.pitch = split_pitch(K); p.out_ld = dx_ld;
    p.Hin = OH; p.Win = OW;
    p.Hout = H; p.Wout = W; p.Cout = C;
    p.M = N * H * W;
    p.S = S; p.T = R * S;
    p.a = 1; p.off = pad; p.step = -dil; p.div = stride;
    if (SCH::SCALED) {
        p.in_exp = h2_exp_ptr(dys, (size_t)N * OH * OW, K);
        p.w_exp = h2_exp_ptr(wts, (size_t)C * R * S, K);
    }
    int ov_tile = -1, ov_split = 0;
    lookup_plan(SCH::ID, 1, N, H, W, C, K, R, S, stride, pad, dil, &ov_tile, &ov_split);
    return run_gemm<SCH>(p, (size_t)N * OH * OW, ov_tile, ov_split, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int semseg_conv2d_fwd_s3(const void* xs, const void* ws, const float* bias, float* y, int y_ld,
                                    int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    return conv_fwd<SchS3>(xs, ws, bias, y, y_ld, N, H, W, C, K, R, S, stride, pad, dil, workspace, workspace_bytes, stream);
}
extern "C" int semseg_conv2d_fwd_h2(const void* xs, const void* ws, const float* bias, float* y, int y_ld,
                                    int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    return conv_fwd<SchH2>(xs, ws, bias, y, y_ld, N, H, W, C, K, R, S, stride, pad, dil, workspace, workspace_bytes, stream);
}
// forward conv + the BN statistics of its result gathered in the GEMM epilogue (EpilogueStats): *parts_out = number of partial
// rows written to stats_ws (layout of semseg_bn_fwd_finish_fused), 0 = the launch plan of this geometry does not allow it
// (split-K, or more than kMaxEpilogueParts wave rows) and the caller runs the separate statistics sweep instead
extern "C" size_t semseg_conv2d_fwd_stats_bytes(int K) {
    return (size_t)kMaxEpilogueParts * 2 * (K > 0 ? K : 0) * (sizeof(double) + sizeof(float));
}
extern "C" int semseg_conv2d_fwd_stats_h2(const void* xs, const void* ws, float* y, int y_ld,
                                          int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                          void* workspace, size_t workspace_bytes, void* stats_ws, size_t stats_ws_bytes,
                                          float* bound_word, int* parts_out, void* stream) {
    if (!parts_out) return SEMSEG_EINVAL;
    EpilogueStats es = {stats_ws, stats_ws_bytes, bound_word, 0};
    const int rc = conv_fwd<SchH2>(xs, ws, nullptr, y, y_ld, N, H, W, C, K, R, S, stride, pad, dil, workspace, workspace_bytes,
                                   stream, &es);
    *parts_out = rc ? 0 : es.parts;
    return rc;
}
extern "C" int semseg_conv2d_dgrad_s3(const void* dys, const void* wts, float* dx, int dx_ld,
                                      int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    return conv_dgrad<SchS3>(dys, wts, dx, dx_ld, N, H, W, C, K, R, S, stride, pad, dil, workspace, workspace_bytes, stream);
}
extern "C" int semseg_conv2d_dgrad_h2(const void* dys, const void* wts, float* dx, int dx_ld,
                                      int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    return conv_dgrad<SchH2>(dys, wts, dx, dx_ld, N, H, W, C, K, R, S, stride, pad, dil, workspace, workspace_bytes, stream);
}

// the 16 GEMMs of a Winograd convolution (winograd.hip) as one batched launch of the LDS-DMA kernels: per frequency f
// M[f] ([tiles][K]) = V[f] ([tiles][C] planes) x U[f]^T ([K][C] planes); no split-K (16 batches fill the chip)
extern "C" int semseg_winograd_gemm_h2(const void* v_planes, const void* u_planes, float* Mout, int tiles, int C, int K,
                                       void* stream) {
    if (!v_planes || !u_planes || !Mout || tiles <= 0 || C <= 0 || K <= 0 || !aligned16(v_planes) || !aligned16(u_planes))
        return SEMSEG_EINVAL;
    SParams p = {};
    p.in = (const uint16_t*)v_planes; p.wgt = (const uint16_t*)u_planes; p.bias = nullptr; p.out = Mout;
    p.Cp = round_up32(C); p.pitch = split_pitch(C); p.out_ld = K;
    p.Hin = 1; p.Win = tiles; p.Hout = 1; p.Wout = tiles; p.Cout = K;
    p.M = tiles;
    p.S = 1; p.T = 1;
    p.a = 1; p.off = 0; p.step = 1; p.div = 1;
    p.in_exp = h2_exp_ptr(v_planes, (size_t)16 * tiles, C);
    p.w_exp = h2_exp_ptr(u_planes, (size_t)16 * K, C);
    const size_t in_plane = (size_t)16 * tiles * p.pitch, w_plane = (size_t)16 * K * p.pitch;
    if (2 * H2_NP * in_plane + SPLIT_ZERO_TAIL_BYTES >= ((size_t)1 << 31) || 2 * H2_NP * w_plane + SPLIT_ZERO_TAIL_BYTES >= ((size_t)1 << 31))
        return SEMSEG_EINVAL;
    p.in_plane = (uint32_t)in_plane;
    p.w_plane = (uint32_t)w_plane;
    p.chunks = p.Cp / 32;
    p.ktiles = p.chunks;
    p.kt_per_split = p.ktiles;
    p.splits = 1;
    p.partial = nullptr;
    p.batches = 16; p.batch_in_rows = tiles; p.batch_w_rows = K; p.batch_out_rows = tiles;
    int ov_tile = -1, ov_split = 0;
    lookup_plan(SchH2::ID, 3, tiles, 1, 1, C, K, 3, 3, 1, 1, 1, &ov_tile, &ov_split);
    const int tile = ov_tile >= 0 ? ov_tile : ((tiles >= 1024 && K >= 256) ? 8 : 9);
    const int BM = kTiles[tile][0], BN = kTiles[tile][1];
    p.tiles_m = ceil_div(tiles, BM);
    p.tiles_n = ceil_div(K, BN);
    hipStream_t st = (hipStream_t)stream;
    switch (tile) {
        case 0: return launch_rs<SchH2, 128, 128>(p, st);
        case 6: return launch_dma<SchH2, 128, 128, 4, 2, 2>(p, st);
        case 7: return launch_dma<SchH2, 256, 128, 4, 2, 12>(p, st);
        case 8: return launch_dma<SchH2, 256, 256, 2, 4, 12>(p, st);
        case 9: return launch_dma<SchH2, 128, 128, 4, 2, 12>(p, st);
        case 10: return launch_dma_ring<SchH2, 256, 128, 4, 2, 13>(p, st);
        case 14: return launch_dma<SchH2, 256, 256, 4, 4, 12>(p, st);
        case 25: return launch_dma<SchH2, 256, 256, 4, 4, 25>(p, st);
        case 26: return launch_dma<SchH2, 256, 256, 2, 4, 25>(p, st);
        case 22:
        case 24:
            if (p.Cp % 64) return SEMSEG_EINVAL;
            p.chunks = p.Cp / 64; p.ktiles = p.chunks; p.kt_per_split = p.ktiles;
            return tile == 22 ? launch_dma64<SchH2, 128, 128, 4, 2>(p, st) : launch_dma64<SchH2, 128, 128, 4, 4>(p, st);
        default: return SEMSEG_EINVAL;
    }
}

// ------------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3): batched GEMM AND output transform in one kernel.
//
// The unfused form (semseg_winograd_gemm_h2 + semseg_winograd_output) writes M[f] = V[f] U[f]^T for the 16 frequencies as fp32
// and reads it back -- 16 x tiles x K x 4 B each way; for the data gradient of conv_last (512 -> 4096 @ 64 x 64) that is 537 MB
// written + 537 MB read around a GEMM whose operands are 200 MB (profiles/r4_pmc_conv_last_dgrad_winograd_batch_xcd.txt:
// 1.7 GB moved for 226 MB of algorithmic bytes), and its 16 reductions are only C / 32 = 16 k-tiles long, so every block spends
// its time in prologue and epilogue.  Here ONE block owns a [BM tiles] x [BN output channels] tile for ALL 16 frequencies: an
// outer loop over f, the same LDS-DMA ring over the C / 32 k-tiles of V[f] / U[f] inside, one accumulator set for M[f] and four
// for the outputs y[i][j] = sum_{a,b} At[i][a] At[j][b] M[a][b]  (At = [[1, 1, 1, 0], [0, 1, -1, -1]], coefficients 0 / +-1):
// after the last k-tile of a frequency its M is added to / subtracted from the (up to four) outputs it feeds and cleared.  The
// loop is 16 x C / 32 k-tiles long with a single prologue, nothing but the result reaches memory, and the result is written
// once, as pixels.  Register budget: wave tile 32 x 64 -> 32 (M) + 128 (y) accumulator registers + 48 of fragments on 8 waves
// (two per SIMD, 256 registers each): 128 x 128 per block.  Worth it where K (output channels) is large against C (reduction):
// the data gradients of the wide 3x3 convs (models.py:455-465, 538-540); the forward of those layers has few output tiles and a
// small M and keeps the batched launch.
// ------------------------------------------------------------------------------------------------
struct WFParams {
    const uint16_t* v;       // [NP][16 * tiles][pitch]
    const uint16_t* u;       // [NP][16 * Cout][pitch]
    const int* v_exp;
    const int* u_exp;
    uint32_t v_plane, u_plane;      // elements per plane
    float* out;                     // [N][H][W] pixels x out_ld
    int out_ld;
    int tiles, Cout, chunks, pitch;
    int tiles_m, tiles_n;
    int N, H, W, dil, TH, TW;
};

// PROBE: 0 = the kernel; 1 = DMA stream only (no fragment reads, no MFMAs: what the LDS-DMA ring alone sustains on this access
// pattern); 2 = fragment reads + MFMAs on whatever the LDS holds, ONE tile fetched (what the multiply side alone sustains).
// The probes write garbage and exist for tools/probes/winograd_dgrad_pass.py --form 101.. only.
template <int BM, int BN, int WGM, int WGN, int NSLOT, int PROBE = 0>
__global__ __launch_bounds__(WGM * WGN * 64) void wino_fused_kernel(const WFParams p) {
    constexpr int NP = 2;
    typedef SchH2::frag frag;
    constexpr int NW = WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int FM = WM / 32, FN = WN / 32;
    constexpr int AG = BM / 16 / NW, BG = BN / 16 / NW;
    constexpr int LPT = NP * (AG + BG);
    constexpr int A_BYTES = NP * BM * 64, B_BYTES = NP * BN * 64, BUF_BYTES = A_BYTES + B_BYTES;
    static_assert(AG >= 1 && BG >= 1 && FM >= 1 && FN >= 1, "tile");
    static_assert(NSLOT >= 3 && NSLOT <= 5 && (NSLOT - 1) * LPT < 64, "ring");

    extern __shared__ __align__(16) uint4 smem4[];
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    const WinoFusedBlock gb = wino_fused_block(blockIdx.x, p.tiles_m, p.tiles_n);     // block_order.h
    const int m0 = gb.tm * BM, n0 = gb.tn * BN;

    const uint32_t a_zero = 2u * NP * p.v_plane, b_zero = 2u * NP * p.u_plane;
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.v, 0, (int)(a_zero + SPLIT_ZERO_TAIL_BYTES), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.u, 0, (int)(b_zero + SPLIT_ZERO_TAIL_BYTES), 0x00020000);
    const uint32_t a_plane_b = 2u * p.v_plane, b_plane_b = 2u * p.u_plane;
    const uint32_t row_b = 2u * (uint32_t)p.pitch;
    const uint32_t fa_b = (uint32_t)p.tiles * row_b, fb_b = (uint32_t)p.Cout * row_b;      // frequency strides (bytes)

    const int lrow = lane >> 2;
    const int q = (lane & 3) ^ ((lane >> 4) & 3);
    uint32_t a_src[AG], a_msk[AG], b_src[BG], b_msk[BG];
#pragma unroll
    for (int i = 0; i < AG; ++i) {
        const int m = m0 + (wave + NW * i) * 16 + lrow;
        const bool ok = m < p.tiles;
        a_src[i] = ok ? (uint32_t)m * row_b + 16u * q : a_zero;
        a_msk[i] = ok ? 0xffffffffu : 0u;
    }
#pragma unroll
    for (int i = 0; i < BG; ++i) {
        const int n = n0 + (wave + NW * i) * 16 + lrow;
        const bool ok = n < p.Cout;
        b_src[i] = ok ? (uint32_t)n * row_b + 16u * q : b_zero;
        b_msk[i] = ok ? 0xffffffffu : 0u;
    }
    const int nk = 16 * p.chunks;
    int i_f = 0, i_c = 0;                  // (frequency, chunk) of the next k-tile to ISSUE
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem + (uint32_t)wave * 1024u;
    // PROBE 3: the same stream of pieces fetched by PLAIN 16-byte buffer loads into registers (no LDS-DMA, no LDS at all): what the
    // L1 -> VGPR path sustains on this access pattern, against PROBE 1's L1 -> LDS path (tools/probes/winograd_dgrad_pass.py 103)
    constexpr int PD = 3;                   // k-tiles of plain loads in flight per wave (3 x LPT x 4 registers)
    uint4 dummy = make_uint4(0, 0, 0, 0), hold[PD * LPT];
#pragma unroll
    for (int i = 0; i < PD * LPT; ++i) hold[i] = make_uint4(0, 0, 0, 0);
    auto issue = [&](int kt, int slot) {
        const uint32_t abuf = lds0 + (uint32_t)slot * BUF_BYTES;
        const uint32_t bbuf = abuf + A_BYTES;
        const bool live = kt < nk && (PROBE != 2 || kt < NSLOT);                               // wave-uniform
        if (PROBE == 2 && kt >= NSLOT) return;
        const uint32_t ka = (uint32_t)i_f * fa_b + 64u * (uint32_t)i_c;
        const uint32_t kb = (uint32_t)i_f * fb_b + 64u * (uint32_t)i_c;
        if (live) {
            if (++i_c == p.chunks) { i_c = 0; ++i_f; }
        }
        if (PROBE == 4) {
            // hybrid: the A pieces by LDS-DMA, the B pieces by plain loads (consumed one call later): do the two paths add up?
#pragma unroll
            for (int i = 0; i < BG * NP; ++i) {
                dummy.x ^= hold[i].x; dummy.y ^= hold[i].y; dummy.z ^= hold[i].z; dummy.w ^= hold[i].w;
            }
#pragma unroll
            for (int i = 0; i < AG; ++i)
#pragma unroll
                for (int s = 0; s < NP; ++s) {
                    const uint32_t vo = live ? a_src[i] + ((ka + s * a_plane_b) & a_msk[i]) : a_zero;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(uintptr_t)(abuf + (s * BM + NW * i * 16) * 64), 16, vo, 0, 0, 0);
                }
#pragma unroll
            for (int i = 0; i < BG; ++i)
#pragma unroll
                for (int s = 0; s < NP; ++s) {
                    const uint32_t vo = live ? b_src[i] + ((kb + s * b_plane_b) & b_msk[i]) : b_zero;
                    hold[i * NP + s] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, vo, 0, 0));
                }
            return;
        }
        if (PROBE == 3) {
            // the OLDEST of the PD k-tiles in flight is consumed, the others shift down (register moves the compiler renames away
            // in the unrolled loop body), the new tile's loads take the last place
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                dummy.x ^= hold[i].x; dummy.y ^= hold[i].y; dummy.z ^= hold[i].z; dummy.w ^= hold[i].w;
            }
#pragma unroll
            for (int i = 0; i < (PD - 1) * LPT; ++i) hold[i] = hold[i + LPT];
            constexpr int H0 = (PD - 1) * LPT;
#pragma unroll
            for (int i = 0; i < AG; ++i)
#pragma unroll
                for (int s = 0; s < NP; ++s) {
                    const uint32_t vo = live ? a_src[i] + ((ka + s * a_plane_b) & a_msk[i]) : a_zero;
                    hold[H0 + i * NP + s] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, vo, 0, 0));
                }
#pragma unroll
            for (int i = 0; i < BG; ++i)
#pragma unroll
                for (int s = 0; s < NP; ++s) {
                    const uint32_t vo = live ? b_src[i] + ((kb + s * b_plane_b) & b_msk[i]) : b_zero;
                    hold[H0 + (AG + i) * NP + s] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, vo, 0, 0));
                }
            return;
        }
#pragma unroll
        for (int i = 0; i < AG; ++i)
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                const uint32_t vo = live ? a_src[i] + ((ka + s * a_plane_b) & a_msk[i]) : a_zero;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(uintptr_t)(abuf + (s * BM + NW * i * 16) * 64), 16, vo, 0, 0, 0);
            }
#pragma unroll
        for (int i = 0; i < BG; ++i)
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                const uint32_t vo = live ? b_src[i] + ((kb + s * b_plane_b) & b_msk[i]) : b_zero;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(uintptr_t)(bbuf + (s * BN + NW * i * 16) * 64), 16, vo, 0, 0, 0);
            }
    };

    f32x16 acc[FM][FN], y[4][FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                y[0][i][j][e] = 0.f; y[1][i][j][e] = 0.f; y[2][i][j][e] = 0.f; y[3][i][j][e] = 0.f;
            }

    const int frow = lane & 31;
    const int kb2 = lane >> 5;
    auto read_frags = [&](int slot, int ks, frag (&av)[FM][NP], frag (&bv)[FN][NP]) {
        const uint4* As = reinterpret_cast<const uint4*>(smem + slot * BUF_BYTES);
        const uint4* Bs = reinterpret_cast<const uint4*>(smem + slot * BUF_BYTES + A_BYTES);
        const int ch = 2 * ks + kb2;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int r = wn * WN + j * 32 + frow;
#pragma unroll
            for (int s = 0; s < NP; ++s) bv[j][s] = *reinterpret_cast<const frag*>(&Bs[s * BN * 4 + s_slot(r, ch)]);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int r = wm * WM + i * 32 + frow;
#pragma unroll
            for (int s = 0; s < NP; ++s) av[i][s] = *reinterpret_cast<const frag*>(&As[s * BM * 4 + s_slot(r, ch)]);
        }
    };
    auto mma = [&](const frag (&av)[FM][NP], const frag (&bv)[FN][NP]) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) SchH2::mac(av[i], bv[j], acc[i][j]);
    };
    // M[f] complete: y[i][j] += At[i][a] At[j][b] M  (f = 4 a + b), then clear M.  Coefficients are wave-uniform.
    auto fold = [&](int f) {
        const int a = f >> 2, b = f & 3;
        const float ra[2] = {a < 3 ? 1.f : 0.f, a == 0 ? 0.f : (a == 1 ? 1.f : -1.f)};
        const float cb[2] = {b < 3 ? 1.f : 0.f, b == 0 ? 0.f : (b == 1 ? 1.f : -1.f)};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float c = ra[o >> 1] * cb[o & 1];
            if (c != 0.f) {                                   // uniform branch
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) y[o][i][j][e] = fmaf(c, acc[i][j][e], y[o][i][j][e]);
            }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };

    // ring of NSLOT slots, NSLOT - 1 tiles in flight behind ONE barrier per k-tile, fragment reads software-pipelined as in
    // igemm_dma_kernel (NSLOT == 13 form)
#pragma unroll
    for (int t = 0; t < NSLOT; ++t) issue(t, t);
    frag a0[FM][NP], b0[FN][NP], a1[FM][NP], b1[FN][NP];
    if (PROBE < 3) wait_vm_barrier<(NSLOT - 1) * LPT>();      // tile 0 has landed for every wave
    read_frags(0, 0, a0, b0);
    int slot = 0, c_c = 0, c_f = 0;                            // (frequency, chunk) of the tile being MULTIPLIED
    for (int it = 0; it < nk; ++it) {
        const int next = (slot == NSLOT - 1) ? 0 : slot + 1;
        if (PROBE != 1 && PROBE < 3) {
            read_frags(slot, 1, a1, b1);
            mma(a0, b0);
        }
        if (PROBE == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (PROBE >= 3) __builtin_amdgcn_s_barrier();
        else wait_vm_barrier<(NSLOT - 2) * LPT>();            // my reads of `slot` are done, tile it+1 has landed
        issue(it + NSLOT, slot);
        if (PROBE != 1 && PROBE < 3) {
            read_frags(next, 0, a0, b0);                      // past the last tile: zero tail, never multiplied
            mma(a1, b1);
        }
        if (++c_c == p.chunks) {
            c_c = 0;
            fold(c_f);
            ++c_f;
        }
        slot = next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (PROBE >= 3) {
#pragma unroll
        for (int i = 0; i < PD * LPT; ++i) { dummy.x ^= hold[i].x; dummy.y ^= hold[i].y; dummy.z ^= hold[i].z; dummy.w ^= hold[i].w; }
        if ((dummy.x ^ dummy.y ^ dummy.z ^ dummy.w) == 0x9e3779b9u) p.out[0] = 1.f;      // keeps the loads alive
    }
    __syncthreads();                                           // the ring is dead: LDS becomes the row -> pixel table

    // row (tile) -> offsets of its 2 x 2 output pixels (floats from p.out), -1 where the pixel is outside the image
    int* tab = reinterpret_cast<int*>(smem);                   // [BM][4]
    for (int r = tid; r < BM; r += NW * 64) {
        int t = m0 + r;
        int o4[4] = {-1, -1, -1, -1};
        if (t < p.tiles) {
            const int tx = t % p.TW; t /= p.TW;
            const int ty = t % p.TH; t /= p.TH;
            const int pw = t % p.dil; t /= p.dil;
            const int ph = t % p.dil;
            const int n = t / p.dil;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int h = (2 * ty + i) * p.dil + ph, w = (2 * tx + j) * p.dil + pw;
                    if (h < p.H && w < p.W) o4[i * 2 + j] = ((n * p.H + h) * p.W + w);
                }
        }
        tab[r * 4 + 0] = o4[0]; tab[r * 4 + 1] = o4[1]; tab[r * 4 + 2] = o4[2]; tab[r * 4 + 3] = o4[3];
    }
    __syncthreads();
    float f1, f2;
    descale_factors<SchH2>(p.v_exp, p.u_exp, f1, f2);
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int col = n0 + wn * WN + j * 32 + col_l;
        if (col >= p.Cout) continue;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + row_l;
                const int4 px = *reinterpret_cast<const int4*>(&tab[r * 4]);
                if (px.x >= 0) p.out[(size_t)px.x * p.out_ld + col] = (y[0][i][j][e] * f1) * f2;
                if (px.y >= 0) p.out[(size_t)px.y * p.out_ld + col] = (y[1][i][j][e] * f1) * f2;
                if (px.z >= 0) p.out[(size_t)px.z * p.out_ld + col] = (y[2][i][j][e] * f1) * f2;
                if (px.w >= 0) p.out[(size_t)px.w * p.out_ld + col] = (y[3][i][j][e] * f1) * f2;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// The same kernel on 64-deep k-tiles with FULL 128-byte lines per DMA piece (round 4, form 5).
// The 32-deep loop above gathers 16 rows x 64 B per DMA piece -- half a cache line per row, the other half eight pieces later --
// and passes one barrier per 32 channels.  Here a k-tile is 64 channels: a piece is 8 rows x 128 B (a whole line per row), the LDS
// row is 128 B with the 16-byte chunk c of row r stored at c ^ ((r >> 1) & 7) (a ds_read_b128 lane group -- 16 rows, one chunk index
// -- then covers all 64 banks once; the swizzle is applied on the DMA source address), four k16 steps per barrier.  With two planes
// per operand a whole 64-deep tile of a 128 x 128 block is 64 KiB, so the ring holds HALF tiles: entry j = the A rows (j even) or the
// B rows (j odd) of tile j / 2, five entries of 32 KiB, 1.5 tiles in flight behind the tile being multiplied.
// ------------------------------------------------------------------------------------------------
// SCHED (round 5): bit 0 = the DMA pieces of a k-tile are issued in four groups BEHIND the four MFMA groups of the following tile
// (first group right after the barrier that frees the slots) instead of all eight right after the barrier -- both waves of a SIMD
// leave the barrier together, and eight back-to-back 60 - 185-cycle DMA issues per wave were a window in which neither feeds the
// matrix pipe; bit 1 = s_setprio 1 around every MFMA group.
template <int WGM, int WGN, int SCHED = 0>
__global__ __launch_bounds__(WGM * WGN * 64) void wino_fused64_kernel(const WFParams p) {
    constexpr bool SPREAD = SCHED & 1, PRIO = SCHED & 2;
    constexpr int NP = 2, BM = 128, BN = 128, NENT = 5;
    typedef SchH2::frag frag;
    constexpr int NW = WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int FM = WM / 32, FN = WN / 32;
    constexpr int GPW = BM / 8 / NW;                   // 8-row groups per wave and plane (A and B alike: BM == BN)
    constexpr int PPE = NP * GPW;                      // DMA pieces per wave per ring entry
    constexpr int ENT_BYTES = NP * BM * 128;           // 32 KiB
    static_assert(GPW >= 1 && FM >= 1 && FN >= 1 && 4 * PPE < 64, "tile");

    extern __shared__ __align__(16) uint4 smem4[];
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem4);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const WinoFusedBlock gb = wino_fused_block(blockIdx.x, p.tiles_m, p.tiles_n);
    const int m0 = gb.tm * BM, n0 = gb.tn * BN;

    const uint32_t a_zero = 2u * NP * p.v_plane, b_zero = 2u * NP * p.u_plane;
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.v, 0, (int)(a_zero + SPLIT_ZERO_TAIL_BYTES), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.u, 0, (int)(b_zero + SPLIT_ZERO_TAIL_BYTES), 0x00020000);
    const uint32_t a_plane_b = 2u * p.v_plane, b_plane_b = 2u * p.u_plane;
    const uint32_t row_b = 2u * (uint32_t)p.pitch;
    const uint32_t fa_b = (uint32_t)p.tiles * row_b, fb_b = (uint32_t)p.Cout * row_b;
    const int chunks64 = p.chunks >> 1;                // host: chunks even
    const int nk = 16 * chunks64;                      // k-tiles; ring entries 0 .. 2 nk - 1

    // DMA lane constants: row inside an 8-row piece, physical chunk l & 7 holds logical chunk (l & 7) ^ ((row >> 1) & 7);
    // row = 8 g + (l >> 3) with g = wave + NW i (NW is even: the parity of g is the wave's) -> (row >> 1) & 7 = (4 (wave & 1) + (l >> 4)) & 7
    const int lrow = lane >> 3;
    const int q = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    uint32_t a_src[GPW], a_msk[GPW], b_src[GPW], b_msk[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int m = m0 + (wave + NW * i) * 8 + lrow;
        const bool oka = m < p.tiles;
        a_src[i] = oka ? (uint32_t)m * row_b + 16u * q : a_zero;
        a_msk[i] = oka ? 0xffffffffu : 0u;
        const int n = n0 + (wave + NW * i) * 8 + lrow;
        const bool okb = n < p.Cout;
        b_src[i] = okb ? (uint32_t)n * row_b + 16u * q : b_zero;
        b_msk[i] = okb ? 0xffffffffu : 0u;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
    int i_f = 0, i_c = 0, i_ent = 0, i_slot = 0;       // (frequency, 64-chunk) of the tile the next entry belongs to; entry index; its slot
    // one ring entry in two halves (SPREAD issues them behind different MFMA groups): half h = the row groups [h GPW / 2, (h + 1) GPW / 2)
    // of the entry; the second half closes the entry (advances the (frequency, chunk) walk, the entry index and the slot)
    auto issue_half = [&](int h) {
        const bool live = i_ent < 2 * nk;               // wave-uniform
        const bool is_b = i_ent & 1;
        const uint32_t base = lds0 + (uint32_t)i_slot * ENT_BYTES;
        const uint32_t koff = (uint32_t)i_f * (is_b ? fb_b : fa_b) + 128u * (uint32_t)i_c;
        constexpr int HG = GPW >= 2 ? GPW / 2 : 1;
#pragma unroll
        for (int ii = 0; ii < HG; ++ii) {
            const int i = (GPW >= 2 ? h * HG : 0) + ii;
            if (GPW < 2 && h == 1) break;
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                const uint32_t dst = base + (uint32_t)((s * BM + (wave + NW * i) * 8) * 128);
                if (is_b) {
                    const uint32_t vo = live ? b_src[i] + ((koff + s * b_plane_b) & b_msk[i]) : b_zero;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(uintptr_t)dst, 16, vo, 0, 0, 0);
                } else {
                    const uint32_t vo = live ? a_src[i] + ((koff + s * a_plane_b) & a_msk[i]) : a_zero;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(uintptr_t)dst, 16, vo, 0, 0, 0);
                }
            }
        }
        if (h == 1) {
            if (is_b && live) {                         // the tile's B entry closes it: advance (frequency, chunk)
                if (++i_c == chunks64) { i_c = 0; ++i_f; }
            }
            ++i_ent;
            i_slot = (i_slot == NENT - 1) ? 0 : i_slot + 1;
        }
    };
    auto issue_entry = [&]() {
        issue_half(0);
        issue_half(1);
    };

    f32x16 acc[FM][FN], y[4][FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                y[0][i][j][e] = 0.f; y[1][i][j][e] = 0.f; y[2][i][j][e] = 0.f; y[3][i][j][e] = 0.f;
            }

    const int frow = lane & 31;
    const int kb2 = lane >> 5;
    auto read_frags = [&](int sa, int sb, int ks, frag (&av)[FM][NP], frag (&bv)[FN][NP]) {
        const unsigned char* As = smem + sa * ENT_BYTES;
        const unsigned char* Bs = smem + sb * ENT_BYTES;
        const int ch = 2 * ks + kb2;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int r = wn * WN + j * 32 + frow;
            const int off = r * 128 + ((ch ^ ((r >> 1) & 7)) << 4);
#pragma unroll
            for (int s = 0; s < NP; ++s) bv[j][s] = *reinterpret_cast<const frag*>(Bs + s * BN * 128 + off);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int r = wm * WM + i * 32 + frow;
            const int off = r * 128 + ((ch ^ ((r >> 1) & 7)) << 4);
#pragma unroll
            for (int s = 0; s < NP; ++s) av[i][s] = *reinterpret_cast<const frag*>(As + s * BM * 128 + off);
        }
    };
    auto mma = [&](const frag (&av)[FM][NP], const frag (&bv)[FN][NP]) {
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) SchH2::mac(av[i], bv[j], acc[i][j]);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };
    auto fold = [&](int f) {
        const int a = f >> 2, b = f & 3;
        const float ra[2] = {a < 3 ? 1.f : 0.f, a == 0 ? 0.f : (a == 1 ? 1.f : -1.f)};
        const float cb[2] = {b < 3 ? 1.f : 0.f, b == 0 ? 0.f : (b == 1 ? 1.f : -1.f)};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float c = ra[o >> 1] * cb[o & 1];
            if (c != 0.f) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) y[o][i][j][e] = fmaf(c, acc[i][j][e], y[o][i][j][e]);
            }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };

    // entries 0 .. 4 = A(0) B(0) A(1) B(1) A(2); tile t is multiplied from slots (2t) % 5 and (2t + 1) % 5 and frees them
#pragma unroll
    for (int j = 0; j < NENT; ++j) issue_entry();
    frag a0[FM][NP], b0[FN][NP], a1[FM][NP], b1[FN][NP];
    wait_vm_barrier<3 * PPE>();                               // entries 0 and 1 have landed for every wave
    int sa = 0, sb = 1;
    read_frags(sa, sb, 0, a0, b0);
    int c_c = 0, c_f = 0;
    for (int it = 0; it < nk; ++it) {
        read_frags(sa, sb, 1, a1, b1);
        mma(a0, b0);
        if (SPREAD && it) issue_half(1);                      // second half of the B entry begun behind the last MFMA group of tile it - 1
        read_frags(sa, sb, 2, a0, b0);
        mma(a1, b1);
        if (SPREAD && it) issue_half(0);                      // the A entry into the other slot tile it - 1 has freed
        read_frags(sa, sb, 3, a1, b1);
        mma(a0, b0);
        if (SPREAD && it) issue_half(1);
        // my reads of this tile's two slots are done; the next tile's two entries (the oldest two of the three in flight) have landed
        // (SPREAD: the newest PPE pieces are the A entry just issued, as in the burst form)
        wait_vm_barrier<1 * PPE>();
        if (!SPREAD) {
            issue_entry();                                    // into slot sa
            issue_entry();                                    // into slot sb
        }
        const int na = (sb == NENT - 1) ? 0 : sb + 1;
        const int nb = (na == NENT - 1) ? 0 : na + 1;
        read_frags(na, nb, 0, a0, b0);                        // past the last tile: zero tails, never multiplied
        mma(a1, b1);
        if (SPREAD) issue_half(0);                            // first half of the B entry that refills slot sa (i_slot == sa here)
        sa = na;
        sb = nb;
        if (++c_c == chunks64) {
            c_c = 0;
            fold(c_f);
            ++c_f;
        }
    }
    if (SPREAD) {                                             // close the entry left half issued (dead: past the last tile)
        issue_half(1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int* tab = reinterpret_cast<int*>(smem);                   // [BM][4]
    for (int r = tid; r < BM; r += NW * 64) {
        int t = m0 + r;
        int o4[4] = {-1, -1, -1, -1};
        if (t < p.tiles) {
            const int tx = t % p.TW; t /= p.TW;
            const int ty = t % p.TH; t /= p.TH;
            const int pw = t % p.dil; t /= p.dil;
            const int ph = t % p.dil;
            const int n = t / p.dil;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int h = (2 * ty + i) * p.dil + ph, w = (2 * tx + j) * p.dil + pw;
                    if (h < p.H && w < p.W) o4[i * 2 + j] = ((n * p.H + h) * p.W + w);
                }
        }
        tab[r * 4 + 0] = o4[0]; tab[r * 4 + 1] = o4[1]; tab[r * 4 + 2] = o4[2]; tab[r * 4 + 3] = o4[3];
    }
    __syncthreads();
    float f1, f2;
    descale_factors<SchH2>(p.v_exp, p.u_exp, f1, f2);
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int col = n0 + wn * WN + j * 32 + col_l;
        if (col >= p.Cout) continue;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + row_l;
                const int4 px = *reinterpret_cast<const int4*>(&tab[r * 4]);
                if (px.x >= 0) p.out[(size_t)px.x * p.out_ld + col] = (y[0][i][j][e] * f1) * f2;
                if (px.y >= 0) p.out[(size_t)px.y * p.out_ld + col] = (y[1][i][j][e] * f1) * f2;
                if (px.z >= 0) p.out[(size_t)px.z * p.out_ld + col] = (y[2][i][j][e] * f1) * f2;
                if (px.w >= 0) p.out[(size_t)px.w * p.out_ld + col] = (y[3][i][j][e] * f1) * f2;
            }
    }
}

template <int WGM, int WGN, int SCHED = 0>
static int launch_wino_fused64(const WFParams& p, hipStream_t st) {
    constexpr size_t smem = (size_t)5 * 2 * 128 * 128;
    if (p.chunks & 1) return SEMSEG_EINVAL;                  // the reduction must be whole 64-channel tiles
    static SmemAttrCache attr_cache;
    if (int e = ensure_smem_attr(attr_cache, (const void*)wino_fused64_kernel<WGM, WGN, SCHED>, smem)) return e;
    hipLaunchKernelGGL((wino_fused64_kernel<WGM, WGN, SCHED>), dim3(p.tiles_m * p.tiles_n), dim3(64 * WGM * WGN), smem, st, p);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

template <int BM, int BN, int WGM, int WGN, int NSLOT, int PROBE = 0>
static int launch_wino_fused(const WFParams& p, hipStream_t st) {
    constexpr size_t smem = (size_t)NSLOT * 2 * (BM + BN) * 64;
    static_assert(smem <= 160 * 1024 && smem >= (size_t)BM * 16, "LDS");
    static SmemAttrCache attr_cache;
    if (int e = ensure_smem_attr(attr_cache, (const void*)wino_fused_kernel<BM, BN, WGM, WGN, NSLOT, PROBE>, smem)) return e;
    hipLaunchKernelGGL((wino_fused_kernel<BM, BN, WGM, WGN, NSLOT, PROBE>), dim3(p.tiles_m * p.tiles_n), dim3(64 * WGM * WGN), smem, st, p);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// z[n, h, w, k] = A^T (sum_c V[f][tile, c] U[f][k, c]) A written as pixels: GEMM over the 16 frequencies + output transform in
// one launch.  v_planes: h2 planes of V, rows (f, tile) x C (semseg_winograd_input_h2 / _input_planes_h2); u_planes: rows (f, k) x C.
// form: 0 = 128 x 128 block on 8 waves, 3-slot ring; 1 / 2 = the same on a 4- / 5-slot ring (5 x 32 KiB = the whole LDS of a CU;
// the kernel runs one block per CU either way -- 205 registers x 8 waves -- so the ring depth is free); 3 / 4 = the block on 4
// waves of 64 x 64 (one per SIMD, 452 registers) on 4 / 5 slots
extern "C" int semseg_winograd_gemm_output_h2(const void* v_planes, const void* u_planes, float* z, int z_ld,
                                              int N, int H, int W, int C, int K, int dil, int form, void* stream) {
    if (!v_planes || !u_planes || !z || N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || dil <= 0 || z_ld < K ||
        !aligned16(v_planes) || !aligned16(u_planes))
        return SEMSEG_EINVAL;
    WFParams p = {};
    p.N = N; p.H = H; p.W = W; p.dil = dil;
    p.TH = (ceil_div(H, dil) + 1) / 2;
    p.TW = (ceil_div(W, dil) + 1) / 2;
    p.tiles = N * dil * dil * p.TH * p.TW;
    p.v = (const uint16_t*)v_planes; p.u = (const uint16_t*)u_planes; p.out = z; p.out_ld = z_ld;
    p.Cout = K; p.chunks = round_up32(C) / 32; p.pitch = split_pitch(C);
    p.v_exp = h2_exp_ptr(v_planes, (size_t)16 * p.tiles, C);
    p.u_exp = h2_exp_ptr(u_planes, (size_t)16 * K, C);
    const size_t v_plane = (size_t)16 * p.tiles * p.pitch, u_plane = (size_t)16 * K * p.pitch;
    if (2 * H2_NP * v_plane + SPLIT_ZERO_TAIL_BYTES >= ((size_t)1 << 31) || 2 * H2_NP * u_plane + SPLIT_ZERO_TAIL_BYTES >= ((size_t)1 << 31) ||
        (size_t)N * H * W * z_ld >= ((size_t)1 << 31))
        return SEMSEG_EINVAL;
    p.v_plane = (uint32_t)v_plane;
    p.u_plane = (uint32_t)u_plane;
    p.tiles_m = ceil_div(p.tiles, 128);
    p.tiles_n = ceil_div(K, 128);
    hipStream_t st = (hipStream_t)stream;
    switch (form) {
        case 0: return launch_wino_fused<128, 128, 4, 2, 3>(p, st);
        case 1: return launch_wino_fused<128, 128, 4, 2, 4>(p, st);
        case 2: return launch_wino_fused<128, 128, 4, 2, 5>(p, st);
        case 3: return launch_wino_fused<128, 128, 2, 2, 4>(p, st);      // 4 waves, 64 x 64 per wave (a third less LDS read traffic per MFMA)
        case 4: return launch_wino_fused<128, 128, 2, 2, 5>(p, st);
        case 5: return launch_wino_fused64<4, 2>(p, st);                       // 64-deep k-tiles, full-line DMA pieces, half-tile ring
        case 6: return launch_wino_fused64<2, 2>(p, st);
        case 7: return launch_wino_fused64<4, 2, 1>(p, st);                    // form 5 with the DMA pieces spread behind the MFMA groups
        case 8: return launch_wino_fused64<4, 2, 3>(p, st);                    // ... and s_setprio 1 around the MFMA groups
        case 9: return launch_wino_fused64<4, 2, 2>(p, st);                    // form 5 with s_setprio alone
        case 100: return launch_wino_fused<128, 128, 4, 2, 5, 1>(p, st);      // probes of form 2 (garbage results)
        case 101: return launch_wino_fused<128, 128, 4, 2, 5, 2>(p, st);
        case 102: return launch_wino_fused<128, 128, 4, 2, 5, 3>(p, st);      // the piece stream by plain loads into registers
        case 103: return launch_wino_fused<128, 128, 4, 2, 5, 4>(p, st);      // A pieces by LDS-DMA, B pieces by plain loads
        default: return SEMSEG_EINVAL;
    }
}

// ------------------------------------------------------------------------------------------------
// bias gradient: db[k] = sum_m dy[m][k]   (two deterministic passes, fp64 combine)
// ------------------------------------------------------------------------------------------------
// block = 64 channels x 4 row lanes; grid = (channel groups, row chunks); partial[chunk][k] (fp64)
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const float* __restrict__ dy, int dy_ld, int M, int K,
                                                                int rows_per_chunk, double* __restrict__ partial) {
    __shared__ double red[4][64];
    const int kx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kx;
    const int m_begin = blockIdx.y * rows_per_chunk;
    const int m_end = min(M, m_begin + rows_per_chunk);
    float s = 0.f;
    if (k < K)
        for (int m = m_begin + ry; m < m_end; m += 4) s += dy[(size_t)m * dy_ld + k];
    red[ry][kx] = (double)s;
    __syncthreads();
    if (ry == 0 && k < K) partial[(size_t)blockIdx.y * K + k] = red[0][kx] + red[1][kx] + red[2][kx] + red[3][kx];
}

// 16 channels x 16 partial lanes per block (a serial loop over 128 partials per thread was 30 us)
__global__ __launch_bounds__(256) void bias_grad_finish_kernel(const double* __restrict__ partial, int nchunks, int K,
                                                               float* __restrict__ db) {
    __shared__ double red[16][17];
    const int kl = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int k = blockIdx.x * 16 + kl;
    double s = 0.0;
    if (k < K)
        for (int i = lane; i < nchunks; i += 16) s += partial[(size_t)i * K + k];
    red[lane][kl] = s;
    __syncthreads();
    if (lane == 0 && k < K) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i][kl];
        db[k] = (float)t;
    }
}

extern "C" int semseg_bias_grad(const float* dy, int dy_ld, float* db, int M, int K, void* workspace, size_t workspace_bytes,
                                void* stream) {
    if (!dy || !db || M <= 0 || K <= 0 || dy_ld < K) return SEMSEG_EINVAL;
    const int nchunks = min(256, ceil_div(M, 64));
    const int rows_per_chunk = ceil_div(M, nchunks);
    if (!workspace || workspace_bytes < (size_t)nchunks * K * sizeof(double)) return SEMSEG_EWORKSPACE;
    hipLaunchKernelGGL(bias_grad_partial_kernel, dim3(ceil_div(K, 64), nchunks), dim3(256), 0, (hipStream_t)stream, dy, dy_ld,
                       M, K, rows_per_chunk, (double*)workspace);
    SEMSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(ceil_div(K, 16)), dim3(256), 0, (hipStream_t)stream,
                       (const double*)workspace, nchunks, K, db);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// weight gradient on split operands
//   dw[k][t][c] = sum_m dy[m][k] * x[pix(m,t)][c]
// One block = one (tap, 128/64 k, 128/64 c) tile over a range of pixels.  Both operand tiles are staged
// pixel-major ([32 pixels][channels], the HBM layout -> coalesced 16-byte loads); the MFMA wants 8 consecutive
// PIXELS per lane, which ds_read_b64_tr_b16 delivers: within a 16-lane group lane 4j+q addresses the 4 channels
// 4q..4q+3 of pixel j, and lane i receives channel i of pixels 0..3 (a 4x16 -> 16x4 transpose in the LDS crossbar).
// LDS row stride = channels*2 + 64 B: (row j, 16-channel half g, chunk q) -> bank 16j + 8g + 2q, all distinct for
// the 32 lanes of a half wave.
// ------------------------------------------------------------------------------------------------
struct WParams {
    const uint16_t* xs;    // [NP][N*H*W][xpitch]
    const uint16_t* dys;   // [NP][M][dypitch]
    const int* x_exp;      // h2: exponent words
    const int* dy_exp;
    uint32_t x_plane, dy_plane;
    float* dw;             // [K][T][C]
    float* partial;        // [splits][K*T*C]
    int Cp, Kp, xpitch, dypitch;
    int H, W, C;
    int OH, OW, K;
    int M;
    int S, T;
    int stride, pad, dil;
    int tiles_k, tiles_c;
    int m_per_split, splits;
    int batches, batch_rows;   // batched weight gradient (wgrad_dma_kernel only, blockIdx.z = batch): both operands advance by
                               // batch_rows rows per batch, the output by K*T*C; slabs are [split][batch][K*T*C]
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int NP, int BT>
struct WTile {
    static constexpr int CPR = BT / 8;            // 16-byte chunks per pixel row
    static constexpr int RPP = 256 / CPR;         // pixel rows per load pass
    static constexpr int PASS = 32 / RPP;
    static constexpr int STRIDE = BT * 2 + 64;    // bytes
    static constexpr int BYTES = NP * 32 * STRIDE;
};

// the block of wgrad_kernel: `lin` = the block's position in the launch order x + gridDim.x * y of a (tiles, splits) grid, `nsplits`
// = that grid's y extent -- parameters, so that wgrad_multi_kernel can run the blocks of MANY problems in one launch
template <class SCH, int BM, int BN>
__device__ __forceinline__ void wgrad_block_body(const WParams& p, int lin, int nsplits) {
    constexpr int NP = SCH::NP;
    typedef typename SCH::frag frag;
    using TA = WTile<NP, BM>;
    using TB = WTile<NP, BN>;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 32, FN = WN / 32;
    static_assert(TA::PASS >= 1 && TB::PASS >= 1 && FM >= 1 && FN >= 1, "tile");

    extern __shared__ __align__(16) unsigned char smem_w[];
    unsigned char* As = smem_w;                  // dy tile  [NP][32][STRIDE_A]
    unsigned char* Bs = smem_w + TA::BYTES;      // x tile   [NP][32][STRIDE_B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // block -> (tile, z): the hardware deals the blocks of a launch to the XCDs in the order x + gridDim.x * y, so the remap
    // runs over (tile, split) TOGETHER: an XCD receives whole row chunks z, and the tiles of a chunk -- the taps / channel
    // blocks that read the same rows of x and dy -- meet in ONE L2 (a remap of x alone assumes gridDim.x % 8 == 0 and still
    // deals every chunk to all eight XCDs: profiles/r4_pmc_step_traffic_cfg1.txt, 9 x 56 blocks fetched 9x their operands)
    const int ntiles = p.tiles_k * p.tiles_c * p.T;
    const WgradBlock wb = wgrad_block(lin, ntiles, nsplits);      // block_order.h
    const int tile = wb.tile;
    const int t = tile % p.T;
    const int tc = (tile / p.T) % p.tiles_c;
    const int tk = tile / (p.T * p.tiles_c);
    const int k0 = tk * BM, c0 = tc * BN;
    const int r = t / p.S, s = t - r * p.S;
    const int z = wb.z;
    const int m_begin = z * p.m_per_split;
    const int m_end = min(p.M, m_begin + p.m_per_split);

    const int qa = tid % TA::CPR, rowa = tid / TA::CPR;
    const int qb = tid % TB::CPR, rowb = tid / TB::CPR;
    const bool ka_ok = (k0 + 8 * qa) < p.Kp;
    const bool cb_ok = (c0 + 8 * qb) < p.Cp;
    const int HWo = p.OH * p.OW;
    const float inv_hwo = 1.0f / (float)HWo, inv_ow = 1.0f / (float)p.OW;

    uint4 ra[TA::PASS][NP], rb[TB::PASS][NP];
    auto load_tile = [&](int mt) {
#pragma unroll
        for (int i = 0; i < TA::PASS; ++i) {
            const int m = mt + rowa + i * TA::RPP;
            const bool ok = (m < m_end) && ka_ok;
            const uint32_t o = ok ? (uint32_t)m * (uint32_t)p.dypitch + k0 + 8 * qa : 0u;
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                const uint4 v = *reinterpret_cast<const uint4*>(p.dys + (size_t)h * p.dy_plane + o);
                ra[i][h] = ok ? v : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < TB::PASS; ++i) {
            const int mr = mt + rowb + i * TB::RPP;
            const int m = min(mr, p.M - 1);
            // m -> (n, oh, ow): float-reciprocal division + one correction step (exact for m < 2^24)
            int n = (int)((float)m * inv_hwo);
            int rem = m - n * HWo;
            if (rem < 0) { --n; rem += HWo; } else if (rem >= HWo) { ++n; rem -= HWo; }
            int oh = (int)((float)rem * inv_ow);
            int ow = rem - oh * p.OW;
            if (ow < 0) { --oh; ow += p.OW; } else if (ow >= p.OW) { ++oh; ow -= p.OW; }
            const int ih = oh * p.stride - p.pad + r * p.dil;
            const int iw = ow * p.stride - p.pad + s * p.dil;
            const bool ok = (mr < m_end) & (ih >= 0) & (iw >= 0) & (ih < p.H) & (iw < p.W) & cb_ok;
            const uint32_t o = ok ? (uint32_t)((n * p.H + ih) * p.W + iw) * (uint32_t)p.xpitch + c0 + 8 * qb : 0u;
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                const uint4 v = *reinterpret_cast<const uint4*>(p.xs + (size_t)h * p.x_plane + o);
                rb[i][h] = ok ? v : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < TA::PASS; ++i)
#pragma unroll
            for (int h = 0; h < NP; ++h)
                *reinterpret_cast<uint4*>(As + (h * 32 + rowa + i * TA::RPP) * TA::STRIDE + 16 * qa) = ra[i][h];
#pragma unroll
        for (int i = 0; i < TB::PASS; ++i)
#pragma unroll
            for (int h = 0; h < NP; ++h)
                *reinterpret_cast<uint4*>(Bs + (h * 32 + rowb + i * TB::RPP) * TB::STRIDE + 16 * qb) = rb[i][h];
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // transpose-read addressing: lane -> (pixel row within an 8-pixel block, channel offset)
    const int G = lane >> 4, i16 = lane & 15;
    const int tr_row = 8 * (G >> 1) + (i16 >> 2);            // + 16*ks + 4*half
    const int tr_col = 16 * (G & 1) + 4 * (i16 & 3);         // + fragment base channel
    const unsigned char* a_base = As + tr_row * TA::STRIDE + (wm * WM + tr_col) * 2;
    const unsigned char* b_base = Bs + tr_row * TB::STRIDE + (wn * WN + tr_col) * 2;

    if (m_begin < m_end) {
        load_tile(m_begin);
        store_tile();
    }
    __syncthreads();

    auto compute_tile = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag av[FM][NP], bv[FN][NP];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int h = 0; h < NP; ++h) {
                    const unsigned char* q0 = a_base + ((h * 32 + 16 * ks) * TA::STRIDE) + i * 64;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)q0);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * TA::STRIDE));
                    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    av[i][h] = __builtin_bit_cast(frag, v);
                }
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int h = 0; h < NP; ++h) {
                    const unsigned char* q0 = b_base + ((h * 32 + 16 * ks) * TB::STRIDE) + j * 64;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)q0);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * TB::STRIDE));
                    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    bv[j][h] = __builtin_bit_cast(frag, v);
                }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) SCH::mac(av[i], bv[j], acc[i][j]);
        }
    };

    for (int mt = m_begin; mt + 32 < m_end; mt += 32) {
        load_tile(mt + 32);
        compute_tile();
        __syncthreads();
        store_tile();
        __syncthreads();
    }
    if (m_begin < m_end) compute_tile();
    S_MFMA_DRAIN();

    float f1 = 1.f, f2 = 1.f;
    descale_factors<SCH>(p.x_exp, p.dy_exp, f1, f2);
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
                float v = acc[i][j][e];
                if constexpr (SCH::SCALED) v = (v * f1) * f2;
                if (k < p.K) dst[((size_t)k * p.T + t) * p.C + c] = v;
            }
        }
    }
}

template <class SCH, int BM, int BN>
struct wgrad_kernel_body {
    static constexpr int THREADS = 256;
    static constexpr bool MULTI = std::is_same<SCH, SchH2>::value && BM <= 128 && BN <= 128;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const WParams p) {
    wgrad_block_body<SCH, BM, BN>(p, blockIdx.x + gridDim.x * blockIdx.y, gridDim.y);
    }
};

// MANY weight gradients in one launch (semseg_conv2d_wgrad_multi_h2): the weight gradients of a backward pass are read by the
// optimizer only, so the host may hold the small ones back (their operands stay alive) and run their blocks side by side -- HRNet's
// 241 launches of 2 - 60 blocks each occupy a fraction of the 256 CUs one after the other; here the blocks of up to kWMulti
// problems form one grid.  Block b belongs to problem j with first[j] <= b < first[j + 1] and runs wgrad_kernel's block
// b - first[j] of that problem's (tiles, splits) grid UNCHANGED (same block order, same summation order: bit-identical results).
constexpr int kWMulti = 24;     // 24 x sizeof(WParams) + the offsets stay below the 4 KB kernel-argument segment
struct WMultiParams {
    WParams p[kWMulti];
    int first[kWMulti + 1];
    int n;
};
static_assert(sizeof(WMultiParams) <= 4000, "kernel arguments");

template <class SCH, int BM, int BN>
__global__ __launch_bounds__(256) void wgrad_multi_kernel(const WMultiParams mp) {
    const int b = blockIdx.x;
    int j = 0;
    while (j + 1 < mp.n && b >= mp.first[j + 1]) ++j;       // block-uniform: scalar loads from the argument segment
    const WParams& p = mp.p[j];
    wgrad_block_body<SCH, BM, BN>(p, b - mp.first[j], p.splits);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of a 3x3, stride-1, pad == dil convolution with ALL NINE TAPS in one block (wtile 10).
// wgrad_kernel gives every tap its own block, so the nine blocks of a pixel chunk each read the chunk's dy rows and (shifted) x
// rows: the 64 x 64 launches of the 64 - 128-channel layers fetched 3.5 - 5x their operands (profiles/r4_pmc_step_traffic_cfg1.txt:
// 3.97 GB per step for `wgrad_kernel<SchH2, 64, 64>`).  Here a block walks its pixel range in chunks of 32 consecutive output
// pixels of ONE image row (OW % 32 == 0): the dy tile [32 pixels][64 k] is staged once, the x tile as a HALO -- 3 image rows
// (oh - dil, oh, oh + dil) x (32 + 2 dil) pixels x 64 channels -- once, and tap (r, s) multiplies dy with the window
// [row r][s dil ... s dil + 31] of the halo: the same transposed LDS reads as wgrad_kernel at a row offset, nine accumulator
// fragments per wave (2 x 2 waves, 32 k x 32 c each: 144 accumulator registers).  102 staged pixel rows per chunk instead of 288,
// one launch block per chunk instead of nine.  Replaces the weight-gradient half of nn.Conv2d.backward at resnet.py:100-108,
// 61-66 (the 3x3 convs of the stem, layer1, layer2) and hrnet.py:26-29 (branch convs with OW % 32 == 0).
// ------------------------------------------------------------------------------------------------
constexpr int WT_STRIDE = 64 * 2 + 64;        // bytes per staged pixel row (64 channels + the 64-byte skew of WTile)
static inline int wtaps_halo_w(int dil) { return 32 + 2 * dil; }
static inline size_t wtaps_smem(int NP, int dil) { return (size_t)NP * (32 + 3 * wtaps_halo_w(dil)) * WT_STRIDE; }

template <class SCH>
__global__ __launch_bounds__(256, 2) void wgrad_taps_kernel(const WParams p) {
    constexpr int NP = SCH::NP;
    typedef typename SCH::frag frag;
    constexpr int XPASS = 4;                     // 3 x (32 + 2 dil) x 8 16-byte chunks <= 1024 for dil <= 5
    extern __shared__ __align__(16) unsigned char smem_t[];
    const int HWp = 32 + 2 * p.dil;
    unsigned char* As = smem_t;                                   // dy tile  [NP][32][WT_STRIDE]
    unsigned char* Xs = smem_t + NP * 32 * WT_STRIDE;             // x halo   [NP][3][HWp][WT_STRIDE]
    const int x_plane_lds = 3 * HWp * WT_STRIDE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.tiles_k * p.tiles_c;
    const WgradBlock wb = wgrad_block(blockIdx.x + gridDim.x * blockIdx.y, ntiles, gridDim.y);
    const int tc = wb.tile % p.tiles_c, tk = wb.tile / p.tiles_c;
    const int k0 = tk * 64, c0 = tc * 64;
    const int z = wb.z;
    const int m_begin = z * p.m_per_split;
    const int m_end = min(p.M, m_begin + p.m_per_split);

    const int q8 = tid & 7, rowa = tid >> 3;                       // dy: 32 rows x 8 chunks = one pass
    const bool ka_ok = (k0 + 8 * q8) < p.Kp;
    const bool cb_ok = (c0 + 8 * q8) < p.Cp;
    const int HWo = p.OH * p.OW;
    const int nx = 3 * HWp * 8;                                    // 16-byte chunks of the halo per plane

    uint4 ra[NP], rb[XPASS][NP];
    auto load_chunk = [&](int mt) {
        // the chunk's 32 output pixels: image n, row oh, columns ow0 ... ow0 + 31 (block-uniform)
        const int n = mt / HWo;
        const int rem = mt - n * HWo;
        const int oh = rem / p.OW;
        const int ow0 = rem - oh * p.OW;
        {
            const int m = mt + rowa;
            const bool ok = (m < m_end) && ka_ok;
            const uint32_t o = ok ? (uint32_t)m * (uint32_t)p.dypitch + k0 + 8 * q8 : 0u;
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                const uint4 v = *reinterpret_cast<const uint4*>(p.dys + (size_t)h * p.dy_plane + o);
                ra[h] = ok ? v : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            const int idx = tid + i * 256;
            const int ps = idx >> 3;                               // halo pixel slot: row r, column j
            const int r = ps / HWp, j = ps - r * HWp;
            const int ih = oh + (r - 1) * p.dil, iw = ow0 - p.dil + j;
            const bool ok = (idx < nx) & (ih >= 0) & (ih < p.H) & (iw >= 0) & (iw < p.W) & cb_ok;
            const uint32_t o = ok ? (uint32_t)((n * p.H + ih) * p.W + iw) * (uint32_t)p.xpitch + c0 + 8 * q8 : 0u;
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                const uint4 v = *reinterpret_cast<const uint4*>(p.xs + (size_t)h * p.x_plane + o);
                rb[i][h] = ok ? v : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int h = 0; h < NP; ++h) *reinterpret_cast<uint4*>(As + (h * 32 + rowa) * WT_STRIDE + 16 * q8) = ra[h];
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            const int idx = tid + i * 256;
            if (idx < nx) {
#pragma unroll
                for (int h = 0; h < NP; ++h)
                    *reinterpret_cast<uint4*>(Xs + h * x_plane_lds + (idx >> 3) * WT_STRIDE + 16 * q8) = rb[i][h];
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int G = lane >> 4, i16 = lane & 15;
    const int tr_row = 8 * (G >> 1) + (i16 >> 2);
    const int tr_col = 16 * (G & 1) + 4 * (i16 & 3);
    const unsigned char* a_base = As + tr_row * WT_STRIDE + (wm * 32 + tr_col) * 2;
    const unsigned char* b_base = Xs + tr_row * WT_STRIDE + (wn * 32 + tr_col) * 2;

    auto compute_chunk = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag av[NP];
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                const unsigned char* q0 = a_base + (h * 32 + 16 * ks) * WT_STRIDE;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)q0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * WT_STRIDE));
                av[h] = __builtin_bit_cast(frag, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2) {
                    frag bv[NP];
#pragma unroll
                    for (int h = 0; h < NP; ++h) {
                        const unsigned char* q0 = b_base + h * x_plane_lds + (r * HWp + s2 * p.dil + 16 * ks) * WT_STRIDE;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)q0);
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * WT_STRIDE));
                        bv[h] = __builtin_bit_cast(frag, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                    SCH::mac(av, bv, acc[r * 3 + s2]);
                }
        }
    };

    if (m_begin < m_end) {
        load_chunk(m_begin);
        store_chunk();
    }
    __syncthreads();
    for (int mt = m_begin; mt + 32 < m_end; mt += 32) {
        load_chunk(mt + 32);
        compute_chunk();
        __syncthreads();
        store_chunk();
        __syncthreads();
    }
    if (m_begin < m_end) compute_chunk();
    S_MFMA_DRAIN();

    float f1 = 1.f, f2 = 1.f;
    descale_factors<SCH>(p.x_exp, p.dy_exp, f1, f2);
    float* dst = (p.splits == 1) ? p.dw : p.partial + (size_t)z * p.K * 9 * p.C;
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    const int c = c0 + wn * 32 + col_l;
    if (c < p.C) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + wm * 32 + (e & 3) + 8 * (e >> 2) + row_l;
                float v = acc[t][e];
                if constexpr (SCH::SCALED) v = (v * f1) * f2;
                if (k < p.K) dst[((size_t)k * 9 + t) * p.C + c] = v;
            }
    }
}

static bool wtaps_eligible(int R, int S, int stride, int pad, int dil, int OW, int M) {
    return R == 3 && S == 3 && stride == 1 && pad == dil && dil >= 1 && dil <= 5 && OW % 32 == 0 && M % 32 == 0;
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA weight gradient: the same contraction as wgrad_kernel, but the 32-pixel x (BM | BN)-channel operand tiles go
// global -> LDS by buffer_load_dwordx4 ... lds (no staging VGPRs and, above all, no ds_write_b128: at ~79 B/clk/CU the
// LDS store path of the register-staged kernel costs more cycles per k-tile than its transposed reads).
// LDS image per operand: [plane][128-channel column block][32 pixel rows][256 B]; one DMA instruction ("piece") = 4 rows
// x 256 B = 1 KiB, lane l -> row l>>4, 16-byte chunk l&15.  The 64-byte row padding of wgrad_kernel (which keeps the four
// rows of a ds_read_b64_tr_b16 lane group on distinct banks) is replaced by an XOR swizzle applied on the DMA SOURCE side:
// the physical chunk p of row r holds the logical chunk p ^ ((r & 3) << 2).  Rows that are padding / outside the image /
// past the split fetch the zero tail of the plane buffer; every wave issues exactly LPT pieces per k-tile.
//   NSLOT == 2: wait(tile it) | barrier | multiply | barrier | issue tile it+2      (4 waves, 2 blocks per CU)
//   NSLOT == 3: wait(tile it) | barrier | issue tile it+2 | multiply                (8 waves, one barrier per k-tile)
// ------------------------------------------------------------------------------------------------
template <class SCH, int BM, int BN, int WGM, int WGN, int NSLOT>
struct wgrad_dma_kernel_body {
    static constexpr int THREADS = WGM * WGN * 64;
    static constexpr bool MULTI = std::is_same<SCH, SchH2>::value && BM <= 128 && BN <= 128;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const WParams p) {
    constexpr int NP = SCH::NP;
    typedef typename SCH::frag frag;
    constexpr int NW = WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int FM = WM / 32, FN = WN / 32;
    constexpr int CBA = BM / 128, CBB = BN / 128;          // 128-channel column blocks per operand
    // 16 waves (round 2; four per SIMD hide the DMA / LDS waits better than two, as in igemm_dma_kernel): the 8 four-row groups
    // of a k-tile go to waves 0-7 for the dy operand and to waves 8-15 for the x operand -- every wave still issues the same
    // number of DMA pieces (the vmcnt accounting depends on it), which needs CBA == CBB
    constexpr bool SPLIT_AB = NW == 16;
    constexpr int GPW = SPLIT_AB ? 1 : 8 / NW;             // 4-row groups per wave
    constexpr int LPT = SPLIT_AB ? NP * CBA : GPW * NP * (CBA + CBB);            // DMA pieces per wave per k-tile
    constexpr int A_BYTES = NP * CBA * 32 * 256, B_BYTES = NP * CBB * 32 * 256, BUF_BYTES = A_BYTES + B_BYTES;
    static_assert(BM % 128 == 0 && BN % 128 == 0 && (NW == 4 || NW == 8 || NW == 16) && FM >= 1 && FN >= 1, "tile");
    static_assert(!SPLIT_AB || CBA == CBB, "16 waves: equal operand widths");
    static_assert(WM % 32 == 0 && WN % 32 == 0 && (NSLOT == 2 || NSLOT == 3 || NSLOT == 12 || NSLOT == 25) && 2 * LPT < 64, "tile");
    static_assert(NSLOT != 25 || (CBA == CBB && NP == 2), "the half-tile ring takes square tiles of the h2 scheme");

    extern __shared__ __align__(16) unsigned char smem_w[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    // block -> (tile, z): the hardware deals the blocks of a launch to the XCDs in the order x + gridDim.x * y, so the remap
    // runs over (tile, split) TOGETHER: an XCD receives whole row chunks z, and the tiles of a chunk -- the taps / channel
    // blocks that read the same rows of x and dy -- meet in ONE L2 (a remap of x alone assumes gridDim.x % 8 == 0 and still
    // deals every chunk to all eight XCDs: profiles/r4_pmc_step_traffic_cfg1.txt, 9 x 56 blocks fetched 9x their operands)
    const int ntiles = p.tiles_k * p.tiles_c * p.T;
    const WgradBlock wb = wgrad_block(blockIdx.x + gridDim.x * blockIdx.y, ntiles, gridDim.y);      // block_order.h
    const int tile = wb.tile;
    const int t = tile % p.T;
    const int tc = (tile / p.T) % p.tiles_c;
    const int tk = tile / (p.T * p.tiles_c);
    const int k0 = tk * BM, c0 = tc * BN;
    const int r = t / p.S, s = t - r * p.S;
    const int z = wb.z;
    const int m_begin = z * p.m_per_split;
    const int m_end = min(p.M, m_begin + p.m_per_split);
    const int nk = (m_end - m_begin + 31) >> 5;

    const uint32_t a_plane_b = 2u * p.dy_plane, b_plane_b = 2u * p.x_plane;
    const uint32_t a_zero = NP * a_plane_b, b_zero = NP * b_plane_b;          // byte offsets of the zero tails
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.dys, 0, (int)(a_zero + SPLIT_ZERO_TAIL_BYTES), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.xs, 0, (int)(b_zero + SPLIT_ZERO_TAIL_BYTES), 0x00020000);

    // DMA lane constants: row inside a 4-row piece, logical 16-byte chunk this lane fetches (source-side swizzle)
    const int lrow = lane >> 4;
    const int cl = (lane & 15) ^ (lrow << 2);
    uint32_t a_cb_msk[CBA], b_cb_msk[CBB];                 // all ones: the chunk lies inside the padded channel range
#pragma unroll
    for (int cb = 0; cb < CBA; ++cb) a_cb_msk[cb] = (k0 + cb * 128 + 8 * cl) < p.Kp ? 0xffffffffu : 0u;
#pragma unroll
    for (int cb = 0; cb < CBB; ++cb) b_cb_msk[cb] = (c0 + cb * 128 + 8 * cl) < p.Cp ? 0xffffffffu : 0u;
    const uint32_t a_col_b = 2u * (uint32_t)k0 + 16u * cl, b_col_b = 2u * (uint32_t)c0 + 16u * cl;
    const uint32_t a_row_b = 2u * (uint32_t)p.dypitch, b_row_b = 2u * (uint32_t)p.xpitch;
    const int HWo = p.OH * p.OW;
    const float inv_hwo = 1.0f / (float)HWo, inv_ow = 1.0f / (float)p.OW;

    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem_w;
    auto issue = [&](int mt, int slot) {
        const uint32_t abuf = lds0 + (uint32_t)slot * BUF_BYTES;
        const uint32_t bbuf = abuf + A_BYTES;
        const bool live = mt < m_end;                      // wave-uniform
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) {
            const int g = SPLIT_AB ? (wave & 7) : wave + NW * gi;                  // wave-uniform 4-row group
            const bool do_a = !SPLIT_AB || wave < 8, do_b = !SPLIT_AB || wave >= 8;          // wave-uniform
            const int mr = mt + 4 * g + lrow;
            const bool rowok = live && (mr < m_end);
            const int m = min(mr, p.M - 1);
            // m -> (n, oh, ow): float-reciprocal division + one correction step (exact for m < 2^24)
            int n = (int)((float)m * inv_hwo);
            int rem = m - n * HWo;
            if (rem < 0) { --n; rem += HWo; } else if (rem >= HWo) { ++n; rem -= HWo; }
            int oh = (int)((float)rem * inv_ow);
            int ow = rem - oh * p.OW;
            if (ow < 0) { --oh; ow += p.OW; } else if (ow >= p.OW) { ++oh; ow -= p.OW; }
            const int ih = oh * p.stride - p.pad + r * p.dil;
            const int iw = ow * p.stride - p.pad + s * p.dil;
            const bool pixok = rowok & (ih >= 0) & (iw >= 0) & (ih < p.H) & (iw < p.W);
            const uint32_t a_msk = rowok ? 0xffffffffu : 0u, b_msk = pixok ? 0xffffffffu : 0u;
            const uint32_t brow = (uint32_t)blockIdx.z * (uint32_t)p.batch_rows;
            const uint32_t a_off = ((uint32_t)m + brow) * a_row_b + a_col_b;
            const uint32_t b_off = ((uint32_t)((n * p.H + ih) * p.W + iw) + brow) * b_row_b + b_col_b;
            if (do_a) {
#pragma unroll
            for (int cb = 0; cb < CBA; ++cb)
#pragma unroll
                for (int h = 0; h < NP; ++h) {
                    const uint32_t ok = a_msk & a_cb_msk[cb];
                    const uint32_t vo = ((a_off + cb * 256u + h * a_plane_b) & ok) | (a_zero & ~ok);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        rs_a, (lds_void*)(uintptr_t)(abuf + ((h * CBA + cb) * 32 + 4 * g) * 256), 16, vo, 0, 0, 0);
                }
            }
            if (do_b) {
#pragma unroll
            for (int cb = 0; cb < CBB; ++cb)
#pragma unroll
                for (int h = 0; h < NP; ++h) {
                    const uint32_t ok = b_msk & b_cb_msk[cb];
                    const uint32_t vo = ((b_off + cb * 256u + h * b_plane_b) & ok) | (b_zero & ~ok);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        rs_b, (lds_void*)(uintptr_t)(bbuf + ((h * CBB + cb) * 32 + 4 * g) * 256), 16, vo, 0, 0, 0);
                }
            }
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // transpose-read addressing (logical mapping of wgrad_kernel): lane -> pixel row 8*(G>>1) + (i16>>2) (+16*ks, +4 for the
    // second read) and channels 16*(G&1) + 4*(i16&3) .. +3 of the 32-channel fragment
    const int G = lane >> 4, i16 = lane & 15;
    const int tr_row = 8 * (G >> 1) + (i16 >> 2);
    const int swz = (i16 >> 2) << 2;                        // (row & 3) << 2: tr_row, 16*ks and 4*half keep row & 3
    const int chan_l = 16 * (G & 1) + 4 * (i16 & 3);
    uint32_t a_foff[FM], b_foff[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int cbase = wm * WM + i * 32, within = (cbase & 127) + chan_l;
        a_foff[i] = ((cbase >> 7) * 32 + tr_row) * 256 + (((within >> 3) ^ swz) << 4) + ((within & 7) << 1);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int cbase = wn * WN + j * 32, within = (cbase & 127) + chan_l;
        b_foff[j] = ((cbase >> 7) * 32 + tr_row) * 256 + (((within >> 3) ^ swz) << 4) + ((within & 7) << 1);
    }

    auto read_frags = [&](int slot, int ks, frag (&av)[FM][NP], frag (&bv)[FN][NP]) {
        const unsigned char* As = smem_w + slot * BUF_BYTES;
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                const unsigned char* q0 = As + a_foff[i] + (h * CBA * 32 + 16 * ks) * 256;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)q0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * 256));
                const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                av[i][h] = __builtin_bit_cast(frag, v);
            }
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                const unsigned char* q0 = Bs + b_foff[j] + (h * CBB * 32 + 16 * ks) * 256;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)q0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * 256));
                const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                bv[j][h] = __builtin_bit_cast(frag, v);
            }
    };
    auto mma = [&](const frag (&av)[FM][NP], const frag (&bv)[FN][NP]) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) SCH::mac(av[i], bv[j], acc[i][j]);
    };
    auto compute_tile = [&](int slot) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag av[FM][NP], bv[FN][NP];
            read_frags(slot, ks, av, bv);
            mma(av, bv);
        }
    };

    if constexpr (NSLOT == 25) {
        // ---- ring of five half tiles with spread DMA issue (round 5; see igemm_dma_kernel NSLOT == 25) ---------------------------
        // entry = the dy rows (even entries) or the x rows (odd) of a 32-pixel k-tile, A_BYTES each; tile it is multiplied from two
        // slots while three entries fly; the x entry of tile it+2 is issued behind the second MFMA group of tile it, the dy entry
        // of tile it+3 behind the first group of tile it+1.  16 waves: EVERY wave takes part in both entries (row group wave & 7,
        // column block / plane wave >> 3), so the vmcnt accounting is the same for all of them.
        constexpr int GPE = NW == 16 ? 1 : 8 / NW;                       // 4-row groups per wave and entry
        constexpr int PPE = NW == 16 ? NP * CBA / 2 : GPE * NP * CBA;     // DMA pieces per wave and entry
        constexpr int NENT = 5;
        static_assert(PPE >= 1 && 3 * PPE < 64, "ring");
        int i_ent = 0, i_slot = 0;
        auto issue_entry = [&]() {
            const int mt = m_begin + (i_ent >> 1) * 32;
            const bool is_b = i_ent & 1;
            const bool live = mt < m_end;                  // wave-uniform
            const uint32_t base = lds0 + (uint32_t)i_slot * A_BYTES;
#pragma unroll
            for (int gi = 0; gi < GPE; ++gi) {
                const int g = NW == 16 ? (wave & 7) : wave + NW * gi;
                const int mr = mt + 4 * g + lrow;
                const bool rowok = live && (mr < m_end);
                const int m = min(mr, p.M - 1);
                const uint32_t brow = (uint32_t)blockIdx.z * (uint32_t)p.batch_rows;
                uint32_t off, msk;
                if (!is_b) {
                    msk = rowok ? 0xffffffffu : 0u;
                    off = ((uint32_t)m + brow) * a_row_b + a_col_b;
                } else {
                    int n = (int)((float)m * inv_hwo);
                    int rem = m - n * HWo;
                    if (rem < 0) { --n; rem += HWo; } else if (rem >= HWo) { ++n; rem -= HWo; }
                    int oh = (int)((float)rem * inv_ow);
                    int ow = rem - oh * p.OW;
                    if (ow < 0) { --oh; ow += p.OW; } else if (ow >= p.OW) { ++oh; ow -= p.OW; }
                    const int ih = oh * p.stride - p.pad + r * p.dil;
                    const int iw = ow * p.stride - p.pad + s * p.dil;
                    const bool pixok = rowok & (ih >= 0) & (iw >= 0) & (ih < p.H) & (iw < p.W);
                    msk = pixok ? 0xffffffffu : 0u;
                    off = ((uint32_t)((n * p.H + ih) * p.W + iw) + brow) * b_row_b + b_col_b;
                }
#pragma unroll
                for (int cb = 0; cb < CBA; ++cb) {
                    if (NW == 16 && CBA == 2 && cb != (wave >> 3)) continue;           // wave-uniform
#pragma unroll
                    for (int h = 0; h < NP; ++h) {
                        if (NW == 16 && CBA == 1 && h != (wave >> 3)) continue;        // wave-uniform
                        const uint32_t dst = base + (uint32_t)(((h * CBA + cb) * 32 + 4 * g) * 256);
                        if (!is_b) {
                            const uint32_t ok = msk & a_cb_msk[cb];
                            const uint32_t vo = ((off + cb * 256u + h * a_plane_b) & ok) | (a_zero & ~ok);
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(uintptr_t)dst, 16, vo, 0, 0, 0);
                        } else {
                            const uint32_t ok = msk & b_cb_msk[cb];
                            const uint32_t vo = ((off + cb * 256u + h * b_plane_b) & ok) | (b_zero & ~ok);
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(uintptr_t)dst, 16, vo, 0, 0, 0);
                        }
                    }
                }
            }
            ++i_ent;
            i_slot = (i_slot == NENT - 1) ? 0 : i_slot + 1;
        };
        auto read_frags2 = [&](int sa, int sb, int ks, frag (&av)[FM][NP], frag (&bv)[FN][NP]) {
            const unsigned char* As = smem_w + sa * A_BYTES;
            const unsigned char* Bs = smem_w + sb * A_BYTES;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int h = 0; h < NP; ++h) {
                    const unsigned char* q0 = As + a_foff[i] + (h * CBA * 32 + 16 * ks) * 256;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)q0);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * 256));
                    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    av[i][h] = __builtin_bit_cast(frag, v);
                }
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int h = 0; h < NP; ++h) {
                    const unsigned char* q0 = Bs + b_foff[j] + (h * CBB * 32 + 16 * ks) * 256;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)q0);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * 256));
                    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    bv[j][h] = __builtin_bit_cast(frag, v);
                }
        };
#pragma unroll
        for (int j = 0; j < NENT; ++j) issue_entry();        // dy(0) x(0) dy(1) x(1) dy(2)
        frag a0[FM][NP], b0[FN][NP], a1[FM][NP], b1[FN][NP];
        wait_vm_barrier<3 * PPE>();                           // entries 0 and 1 have landed for every wave
        int sa = 0, sb = 1;
        read_frags2(sa, sb, 0, a0, b0);
        for (int it = 0; it < nk; ++it) {
            read_frags2(sa, sb, 1, a1, b1);
            mma(a0, b0);
            if (it) issue_entry();                            // dy rows of tile it + 2 into the second slot tile it - 1 has freed
            wait_vm_barrier<1 * PPE>();                       // my reads of tile it are done; tile it + 1 has landed
            const int na = (sb == NENT - 1) ? 0 : sb + 1;
            const int nb = (na == NENT - 1) ? 0 : na + 1;
            read_frags2(na, nb, 0, a0, b0);                   // past the last tile: zero tails, never multiplied
            mma(a1, b1);
            issue_entry();                                    // x rows of tile it + 2 into slot sa
            sa = na;
            sb = nb;
        }
    } else {
    issue(m_begin, 0);
    issue(m_begin + 32, 1);
    if constexpr (NSLOT == 12) {
        // software-pipelined 2-slot loop, one barrier per k-tile (see igemm_dma_kernel)
        frag a0[FM][NP], b0[FN][NP], a1[FM][NP], b1[FN][NP];
        wait_vm_barrier<LPT>();
        read_frags(0, 0, a0, b0);
        for (int it = 0; it < nk; ++it) {
            const int slot = it & 1;
            read_frags(slot, 1, a1, b1);
            mma(a0, b0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            issue(m_begin + (it + 2) * 32, slot);
            read_frags(slot ^ 1, 0, a0, b0);
            mma(a1, b1);
        }
    } else if constexpr (NSLOT == 2) {
        for (int it = 0; it < nk; ++it) {
            const int slot = it & 1;
            wait_vm_barrier<LPT>();
            compute_tile(slot);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done reading `slot`
            issue(m_begin + (it + 2) * 32, slot);
        }
    } else {
        int slot = 0, fill = 2;
        for (int it = 0; it < nk; ++it) {
            wait_vm_barrier<LPT>();
            issue(m_begin + (it + 2) * 32, fill);
            compute_tile(slot);
            slot = (slot == 2) ? 0 : slot + 1;
            fill = (fill == 2) ? 0 : fill + 1;
        }
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    S_MFMA_DRAIN();

    float f1 = 1.f, f2 = 1.f;
    descale_factors<SCH>(p.x_exp, p.dy_exp, f1, f2);
    const size_t slab = (size_t)p.K * p.T * p.C;
    const int nb = p.batches > 0 ? p.batches : 1;
    float* dst = (p.splits == 1) ? p.dw + (size_t)blockIdx.z * slab : p.partial + ((size_t)z * nb + blockIdx.z) * slab;
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
                float v = acc[i][j][e];
                if constexpr (SCH::SCALED) v = (v * f1) * f2;
                if (k < p.K) dst[((size_t)k * p.T + t) * p.C + c] = v;
            }
        }
    }
    }
};

struct split_wgrad_reduce_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ partial, float* __restrict__ dw, size_t total, int splits) {
    // float4 lanes (the slabs are 16-byte aligned and total % 4 == 0 is checked by the launcher; scalar tail otherwise)
    const size_t quads = total >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(partial);
    float4* o4 = reinterpret_cast<float4*>(dw);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = p4[i];
        for (int z = 1; z < splits; z += 4) {     // 4 loads in flight, additions in slab order (see split_gemm_reduce_kernel)
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (z + u < splits) v[u] = p4[(size_t)(z + u) * quads + i];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (z + u < splits) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        o4[i] = s;
    }
    }
};

struct split_wgrad_reduce_scalar_kernel_body {
    static constexpr int THREADS = 256;
    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, const semseg_batch::U3 gridDim,
                                               const float* __restrict__ partial, float* __restrict__ dw, size_t total,
                                                 int splits) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float s = partial[i];
        for (int z = 1; z < splits; ++z) s += partial[(size_t)z * total + i];
        dw[i] = s;
    }
    }
};

struct WPlan {
    int tile, BM, BN, tiles_k, tiles_c, splits, m_per_split;
};

// wgrad tiles (k x c): 0 = 128x128, 1 = 64x64 (register staged); h2 only: 2 = 128x128 LDS-DMA 2-slot (4 waves),
// 3 = 256x128 LDS-DMA 3-slot ring (8 waves), 4 = 256x256 LDS-DMA 2-slot (8 waves), 5 / 6 = 2 / 4 software pipelined -- chosen by
// the tuner / overrides only
// 10 = 64x64 with all nine taps of a 3x3 stride-1 conv in the block (wgrad_taps_kernel): tiles = tiles_k * tiles_c, not * T
// 11 ... 14 (round 5): the LDS-DMA tiles on the ring of five half tiles with spread DMA issue: 256 x 256 on 16 / 8 waves, 128 x 128 on 8 / 4 waves
// kWTiles: defined next to lookup_plan (the member plan above needs the tile extents)
constexpr int kWTileTaps = 10;

// tuning overrides: SEMSEG_W3_TILE=0..3, SEMSEG_W3_SPLIT=n
static WPlan plan_wgrad(int M, int K, int C, int T, int ov_tile = -1, int ov_split = 0) {
    WPlan pl;
    const int mtiles = ceil_div(M, 32);
    const double tile_cost[15] = {1.0, 0.32, 1.0, 2.0, 4.0, 1.0, 4.0, 4.0, 1.0, 1.0, 2.9, 4.0, 4.0, 1.0, 1.0};
    const int slots[15] = {512, 1024, 512, 256, 256, 512, 256, 256, 512, 512, 512, 256, 256, 512, 512};
    const int force_tile = ov_tile >= 0 ? ov_tile : env_int("SEMSEG_W3_TILE", -1);
    const int force_split = ov_split > 0 ? ov_split : env_int("SEMSEG_W3_SPLIT", 0);
    double best = 1e30;
    int best_t = 1, best_s = 1;
    for (int t = 0; t < 15; ++t) {
        if (force_tile >= 0 && t != force_tile) continue;
        if (force_tile < 0 && t >= 2) continue;
        if (t == 0 && (K < 128 || C < 128) && force_tile < 0) continue;
        const long tiles = (long)ceil_div(K, kWTiles[t][0]) * ceil_div(C, kWTiles[t][1]) * (t == kWTileTaps ? 1 : T);
        const int max_split = (t == kWTileTaps) ? 512 : 64;      // a cap of 256 measured nothing better (profiles/r3i-m_ab_tile_forms.txt)
        for (int sp = 1; sp <= max_split; ++sp) {
            if (force_split > 0 && sp != min(force_split, mtiles)) continue;
            if (force_split <= 0 && sp > 1 && mtiles / sp < 8) break;
            const int mps = ceil_div(mtiles, sp);
            if (ceil_div(mtiles, mps) != sp) continue;
            const long blocks = tiles * sp;
            const long rem = blocks % slots[t];
            const int per_cu = slots[t] / 256;
            const double frac = rem ? (double)min((long)per_cu, (rem + 255) / 256) / per_cu : 0.0;
            const double waves = (double)(blocks / slots[t]) + (rem ? 0.35 + 0.65 * frac : 0.0);
            double time = waves * (mps * tile_cost[t] + 6.0 * tile_cost[t]);
            if (sp > 1) time += (double)(sp + 1) * K * T * C * 4.0 / 4.0e12 / 1.6e-6;
            if (time < best) { best = time; best_t = t; best_s = sp; }
        }
    }
    pl.tile = best_t;
    pl.BM = kWTiles[best_t][0];
    pl.BN = kWTiles[best_t][1];
    pl.tiles_k = ceil_div(K, pl.BM);
    pl.tiles_c = ceil_div(C, pl.BN);
    pl.m_per_split = ceil_div(mtiles, best_s) * 32;
    pl.splits = ceil_div(M, pl.m_per_split);
    return pl;
}

static size_t wgrad_split_workspace_bytes(int M, int K, int C, int T, int ov_tile, int ov_split) {
    const WPlan pl = plan_wgrad(M, K, C, T, ov_tile, ov_split);
    return pl.splits > 1 ? (size_t)pl.splits * K * T * C * sizeof(float) : 0;
}

template <class SCH, int BT>
static int launch_wgrad(const WParams& p, hipStream_t st) {
    constexpr size_t smem = (size_t)2 * WTile<SCH::NP, BT>::BYTES;
    static SmemAttrCache attr_cache;
    dim3 grid(p.tiles_k * p.tiles_c * p.T, p.splits);
    SEMSEG_LAUNCH_BODY((wgrad_kernel_body<SCH, BT, BT>), grid, smem, st, p);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

template <class SCH, int BM, int BN, int WGM, int WGN, int NSLOT>
static int launch_wgrad_dma(const WParams& p, hipStream_t st) {
    constexpr size_t smem = NSLOT == 25 ? (size_t)5 * SCH::NP * (BM / 128) * 32 * 256           // five half tiles
                                        : (size_t)(NSLOT == 12 ? 2 : NSLOT) * SCH::NP * (BM / 128 + BN / 128) * 32 * 256;
    static_assert(smem <= 160 * 1024, "LDS");
    // 32-bit byte offsets in the buffer descriptors
    if ((size_t)2 * SCH::NP * p.x_plane + SPLIT_ZERO_TAIL_BYTES >= ((size_t)1 << 31) ||
        (size_t)2 * SCH::NP * p.dy_plane + SPLIT_ZERO_TAIL_BYTES >= ((size_t)1 << 31))
        return SEMSEG_EINVAL;
    static SmemAttrCache attr_cache;
    dim3 grid(p.tiles_k * p.tiles_c * p.T, p.splits, p.batches > 0 ? p.batches : 1);
    SEMSEG_LAUNCH_BODY((wgrad_dma_kernel_body<SCH, BM, BN, WGM, WGN, NSLOT>), grid, smem, st, p);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

// the kernel parameters and the launch plan of one weight gradient (everything of conv_wgrad before the launch)
template <class SCH>
static int wgrad_prepare(const void* xs, const void* dys, float* dw,
                         int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                         void* workspace, size_t workspace_bytes, int* splits_out, WParams& p, WPlan& pl) {
    if (!xs || !dys || (!dw && !splits_out) || N <= 0 || C <= 0 || K <= 0 || stride <= 0 || dil <= 0) return SEMSEG_EINVAL;
    if (!aligned16(xs) || !aligned16(dys)) return SEMSEG_EINVAL;
    const int OH = out_dim(H, R, stride, pad, dil), OW = out_dim(W, S, stride, pad, dil);
    if (OH <= 0 || OW <= 0) return SEMSEG_EINVAL;
    p = {};
    p.xs = (const uint16_t*)xs; p.dys = (const uint16_t*)dys; p.dw = dw;
    p.Cp = round_up32(C); p.Kp = round_up32(K); p.xpitch = split_pitch(C); p.dypitch = split_pitch(K);
    p.H = H; p.W = W; p.C = C; p.OH = OH; p.OW = OW; p.K = K;
    p.M = N * OH * OW;
    p.S = S; p.T = R * S;
    p.stride = stride; p.pad = pad; p.dil = dil;
    const size_t x_plane = (size_t)N * H * W * p.xpitch, dy_plane = (size_t)p.M * p.dypitch;
    if (x_plane >= ((size_t)1 << 31) || dy_plane >= ((size_t)1 << 31) || p.M >= (1 << 24)) return SEMSEG_EINVAL;
    p.x_plane = (uint32_t)x_plane; p.dy_plane = (uint32_t)dy_plane;
    if (SCH::SCALED) {
        p.x_exp = h2_exp_ptr(xs, (size_t)N * H * W, C);
        p.dy_exp = h2_exp_ptr(dys, (size_t)p.M, K);
    }
    int ov_tile = -1, ov_split = 0;
    lookup_plan(SCH::ID, 2, N, H, W, C, K, R, S, stride, pad, dil, &ov_tile, &ov_split);
    pl = plan_wgrad(p.M, K, C, p.T, ov_tile, ov_split);
    p.tiles_k = pl.tiles_k; p.tiles_c = pl.tiles_c;
    p.m_per_split = pl.m_per_split; p.splits = pl.splits;
    if (pl.splits > 1 || splits_out) {
        const size_t need = (size_t)pl.splits * K * p.T * C * sizeof(float);
        if (!workspace || workspace_bytes < need || !aligned16(workspace)) return SEMSEG_EWORKSPACE;
        p.partial = (float*)workspace;
        if (splits_out) {
            *splits_out = pl.splits;
            if (pl.splits == 1) p.dw = (float*)workspace;          // the one slab
        }
    }
    return 0;
}

// splits_out != nullptr: "slabs only" -- the partial sums of the `*splits_out` pixel chunks are left in `workspace` as
// [splits][K*T*C] fp32 (a plan without a split writes its one slab there too) and NO reduce is launched: the caller sums the slabs
// of many weight gradients in one multi-tensor launch (semseg_reduce_slabs_multi); `dw` is not touched.
template <class SCH>
static int conv_wgrad(const void* xs, const void* dys, float* dw,
                      int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                      void* workspace, size_t workspace_bytes, void* stream, int* splits_out = nullptr) {
    hipStream_t st = (hipStream_t)stream;
    WParams p;
    WPlan pl;
    const int prc = wgrad_prepare<SCH>(xs, dys, dw, N, H, W, C, K, R, S, stride, pad, dil, workspace, workspace_bytes, splits_out,
                                       p, pl);
    if (prc) return prc;
    const int OW = p.OW;
    int rc = SEMSEG_EINVAL;
    switch (pl.tile) {
        case 0: rc = launch_wgrad<SCH, 128>(p, st); break;
        case 1: rc = launch_wgrad<SCH, 64>(p, st); break;
        case 2:
            if constexpr (SCH::NP == 2) rc = launch_wgrad_dma<SCH, 128, 128, 2, 2, 2>(p, st);
            break;
        case 3:
            if constexpr (SCH::NP == 2) rc = launch_wgrad_dma<SCH, 256, 128, 4, 2, 3>(p, st);
            break;
        case 4:
            if constexpr (SCH::NP == 2) rc = launch_wgrad_dma<SCH, 256, 256, 2, 4, 2>(p, st);
            break;
        case 5:
            if constexpr (SCH::NP == 2) rc = launch_wgrad_dma<SCH, 128, 128, 2, 2, 12>(p, st);
            break;
        case 6:
            if constexpr (SCH::NP == 2) rc = launch_wgrad_dma<SCH, 256, 256, 2, 4, 12>(p, st);
            break;
        case 7:
            if constexpr (SCH::NP == 2) rc = launch_wgrad_dma<SCH, 256, 256, 4, 4, 12>(p, st);     // 16 waves
            break;
        case 8:
            if constexpr (SCH::NP == 2) rc = launch_wgrad_dma<SCH, 128, 128, 4, 2, 12>(p, st);     // 8 waves, 32x64 per wave
            break;
        case 9:
            if constexpr (SCH::NP == 2) rc = launch_wgrad_dma<SCH, 128, 128, 4, 4, 12>(p, st);     // 16 waves, 32x32 per wave
            break;
        case 11:
            if constexpr (SCH::NP == 2) rc = launch_wgrad_dma<SCH, 256, 256, 4, 4, 25>(p, st);     // ring of five half tiles, 16 waves
            break;
        case 12:
            if constexpr (SCH::NP == 2) rc = launch_wgrad_dma<SCH, 256, 256, 2, 4, 25>(p, st);     // ... 8 waves
            break;
        case 13:
            if constexpr (SCH::NP == 2) rc = launch_wgrad_dma<SCH, 128, 128, 4, 2, 25>(p, st);     // 80 KiB: two blocks per CU
            break;
        case 14:
            if constexpr (SCH::NP == 2) rc = launch_wgrad_dma<SCH, 128, 128, 2, 2, 25>(p, st);
            break;
        case kWTileTaps: {
            if (!wtaps_eligible(R, S, stride, pad, dil, OW, p.M)) return SEMSEG_EINVAL;
            const size_t smem = wtaps_smem(SCH::NP, dil);
            static SmemAttrCache attr_cache;
            if (int e = ensure_smem_attr(attr_cache, (const void*)wgrad_taps_kernel<SCH>, smem)) return e;
            hipLaunchKernelGGL((wgrad_taps_kernel<SCH>), dim3(p.tiles_k * p.tiles_c, p.splits), dim3(256), smem, st, p);
            SEMSEG_LAUNCH_CHECK();
            rc = 0;
            break;
        }
    }
    if (rc) return rc;
    if (pl.splits > 1 && !splits_out) {
        const size_t total = (size_t)K * p.T * C;
        if (total % 4 == 0 && aligned16(p.partial) && aligned16(dw)) {
            const int blocks = (int)min((size_t)2048, ceil_div_sz(total / 4, 256));
            SEMSEG_LAUNCH_BODY((split_wgrad_reduce_kernel_body), dim3(blocks), 0, st, p.partial, dw, total, pl.splits);
        } else {
            const int blocks = (int)min((size_t)2048, ceil_div_sz(total, 256));
            SEMSEG_LAUNCH_BODY((split_wgrad_reduce_scalar_kernel_body), dim3(blocks), 0, st, p.partial, dw, total, pl.splits);
        }
        SEMSEG_LAUNCH_CHECK();
    }
    return 0;
}

// the 16 weight-gradient GEMMs of a Winograd convolution (winograd.hip) as one batched launch of the LDS-DMA wgrad kernel:
// dU[f] ([K][C]) = dM[f]^T ([tiles][K] planes) x V[f] ([tiles][C] planes), reduction over the tiles
extern "C" int semseg_winograd_wgrad_gemm_h2(const void* v_planes, const void* dm_planes, float* dU, int tiles, int C, int K,
                                             void* workspace, size_t workspace_bytes, void* stream) {
    if (!v_planes || !dm_planes || !dU || tiles <= 0 || C <= 0 || K <= 0 || !aligned16(v_planes) || !aligned16(dm_planes))
        return SEMSEG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    WParams p = {};
    p.xs = (const uint16_t*)v_planes; p.dys = (const uint16_t*)dm_planes; p.dw = dU;
    p.Cp = round_up32(C); p.Kp = round_up32(K); p.xpitch = split_pitch(C); p.dypitch = split_pitch(K);
    p.H = 1; p.W = tiles; p.C = C; p.OH = 1; p.OW = tiles; p.K = K;
    p.M = tiles;
    p.S = 1; p.T = 1;
    p.stride = 1; p.pad = 0; p.dil = 1;
    const size_t x_plane = (size_t)16 * tiles * p.xpitch, dy_plane = (size_t)16 * tiles * p.dypitch;
    if (x_plane >= ((size_t)1 << 31) || dy_plane >= ((size_t)1 << 31)) return SEMSEG_EINVAL;
    p.x_plane = (uint32_t)x_plane; p.dy_plane = (uint32_t)dy_plane;
    p.x_exp = h2_exp_ptr(v_planes, (size_t)16 * tiles, C);
    p.dy_exp = h2_exp_ptr(dm_planes, (size_t)16 * tiles, K);
    p.batches = 16; p.batch_rows = tiles;
    const bool big = K >= 256 && C >= 256;
    const int BM = big ? 256 : 128, BN = big ? 256 : 128;
    p.tiles_k = ceil_div(K, BM); p.tiles_c = ceil_div(C, BN);
    const int mtiles = ceil_div(tiles, 32);
    // enough blocks for two waves of the chip; splits over the tiles only when the 16 batches do not provide them
    int splits = 1;
    while (splits < 8 && (long)p.tiles_k * p.tiles_c * 16 * splits < 384 && mtiles / (splits * 2) >= 8) splits *= 2;
    p.m_per_split = ceil_div(mtiles, splits) * 32;
    p.splits = ceil_div(tiles, p.m_per_split);
    const size_t slab = (size_t)16 * K * C;
    if (p.splits > 1) {
        if (!workspace || workspace_bytes < (size_t)p.splits * slab * sizeof(float)) return SEMSEG_EWORKSPACE;
        p.partial = (float*)workspace;
    }
    // form of the batched launch: 0 = the plain 2-slot loop of rounds 2-4; 1 / 2 = the ring of five half tiles on 8 / 16 waves (round 5)
    static const int form = env_int("SEMSEG_WINO_WGRAD_FORM", 1);
    int rc;
    if (!big) rc = form ? launch_wgrad_dma<SchH2, 128, 128, 2, 2, 25>(p, st) : launch_wgrad_dma<SchH2, 128, 128, 2, 2, 2>(p, st);
    else if (form == 2) rc = launch_wgrad_dma<SchH2, 256, 256, 4, 4, 25>(p, st);
    else if (form == 1) rc = launch_wgrad_dma<SchH2, 256, 256, 2, 4, 25>(p, st);
    else rc = launch_wgrad_dma<SchH2, 256, 256, 2, 4, 2>(p, st);
    if (rc) return rc;
    if (p.splits > 1) {
        const int blocks = (int)min((size_t)2048, ceil_div_sz(slab / 4, 256));
        SEMSEG_LAUNCH_BODY((split_wgrad_reduce_kernel_body), dim3(blocks), 0, st, p.partial, dU, slab, p.splits);
        SEMSEG_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" size_t semseg_winograd_wgrad_workspace_bytes(int tiles, int C, int K) {
    return (size_t)8 * 16 * K * C * sizeof(float);      // at most 8 splits
}

extern "C" int semseg_conv2d_wgrad_s3(const void* xs, const void* dys, float* dw,
                                      int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    return conv_wgrad<SchS3>(xs, dys, dw, N, H, W, C, K, R, S, stride, pad, dil, workspace, workspace_bytes, stream);
}
extern "C" int semseg_conv2d_wgrad_h2(const void* xs, const void* dys, float* dw,
                                      int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    return conv_wgrad<SchH2>(xs, dys, dw, N, H, W, C, K, R, S, stride, pad, dil, workspace, workspace_bytes, stream);
}

// The weight gradient as SLABS: the kernel of semseg_conv2d_wgrad_h2 on the same launch plan, its per-chunk partial sums left in
// `slabs` ([*splits_out][K*R*S*C] fp32, KRSC order; slabs_bytes >= semseg_conv2d_wgrad_slabs_bytes), no reduce launch.
extern "C" size_t semseg_conv2d_wgrad_slabs_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0 || dil <= 0) return 0;
    const int OH = out_dim(H, R, stride, pad, dil), OW = out_dim(W, S, stride, pad, dil);
    if (OH <= 0 || OW <= 0) return 0;
    int t2 = -1, s2 = 0;
    lookup_plan(SchH2::ID, 2, N, H, W, C, K, R, S, stride, pad, dil, &t2, &s2);
    const WPlan pl = plan_wgrad(N * OH * OW, K, C, R * S, t2, s2);
    return (size_t)pl.splits * K * R * S * C * sizeof(float);
}
extern "C" int semseg_conv2d_wgrad_slabs_h2(const void* xs, const void* dys, float* slabs, size_t slabs_bytes, int* splits_out,
                                            int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                            void* stream) {
    if (!splits_out) return SEMSEG_EINVAL;
    return conv_wgrad<SchH2>(xs, dys, nullptr, N, H, W, C, K, R, S, stride, pad, dil, slabs, slabs_bytes, stream, splits_out);
}

// MANY weight gradients as slabs in ONE launch (wgrad_multi_kernel): problem i leaves its `splits` partial sums in
// problems[i].slabs exactly as semseg_conv2d_wgrad_slabs_h2 would -- same plan, same blocks, same bits.  Only problems whose launch
// plan is the register-staged 64 x 64 tile (semseg_conv2d_wgrad_tile_h2 == 1: the small layers, the ones that do not fill the chip
// on their own) can be batched; any other plan -> SEMSEG_EINVAL before anything is launched.
extern "C" int semseg_conv2d_wgrad_tile_h2(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0 || dil <= 0) return SEMSEG_EINVAL;
    const int OH = out_dim(H, R, stride, pad, dil), OW = out_dim(W, S, stride, pad, dil);
    if (OH <= 0 || OW <= 0) return SEMSEG_EINVAL;
    int t2 = -1, s2 = 0;
    lookup_plan(SchH2::ID, 2, N, H, W, C, K, R, S, stride, pad, dil, &t2, &s2);
    return plan_wgrad(N * OH * OW, K, C, R * S, t2, s2).tile;
}
extern "C" int semseg_conv2d_wgrad_multi_h2(semseg_wgrad_problem* problems_host, int n, void* stream) {
    if (n < 0 || (n > 0 && !problems_host)) return SEMSEG_EINVAL;
    if (n == 0) return 0;
    std::vector<WParams> ps((size_t)n);
    for (int i = 0; i < n; ++i) {
        semseg_wgrad_problem& q = problems_host[i];
        WPlan pl;
        const int rc = wgrad_prepare<SchH2>(q.xs, q.dys, nullptr, q.N, q.H, q.W, q.C, q.K, q.R, q.S, q.stride, q.pad, q.dil,
                                            q.slabs, q.slabs_bytes, &q.splits, ps[i], pl);
        if (rc) return rc;
        if (pl.tile != 1) return SEMSEG_EINVAL;
    }
    constexpr size_t smem = (size_t)2 * WTile<SchH2::NP, 64>::BYTES;
    static SmemAttrCache attr_cache;
    if (int e = ensure_smem_attr(attr_cache, (const void*)wgrad_multi_kernel<SchH2, 64, 64>, smem)) return e;
    // longest blocks first: a block's run time is its pixel range (m_per_split rows through a 64 x 64 tile), and the hardware starts
    // the blocks of a launch in order -- in the order of the backward pass the stem's long blocks would start last and be the tail.
    // (ONE launch for all problems, the table in device memory, was measured no faster than 24 per launch: DESIGN, item 11.)
    std::vector<int> order((size_t)n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ps[a].m_per_split > ps[b].m_per_split; });
    for (int base = 0; base < n; base += kWMulti) {
        WMultiParams mp = {};
        mp.n = n - base < kWMulti ? n - base : kWMulti;
        long blocks = 0;
        for (int i = 0; i < mp.n; ++i) {
            mp.p[i] = ps[order[base + i]];
            mp.first[i] = (int)blocks;
            blocks += (long)mp.p[i].tiles_k * mp.p[i].tiles_c * mp.p[i].T * mp.p[i].splits;
        }
        if (blocks >= ((long)1 << 31)) return SEMSEG_EINVAL;
        mp.first[mp.n] = (int)blocks;
        hipLaunchKernelGGL((wgrad_multi_kernel<SchH2, 64, 64>), dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream, mp);
        SEMSEG_LAUNCH_CHECK();
    }
    return 0;
}

// out[i] = slabs[0][i] + slabs[1][i] + ... (slab order: the order split_wgrad_reduce_kernel adds in) for MANY tensors in one launch:
// blockIdx.y = tensor, the blocks of a row walk the tensor's float4 lanes.  One launch per training step instead of one per weight
// gradient with a split plan (58 on configs[1], 270 on HRNetV2: launches of 4 - 7 us each at the launch floor).
constexpr int SLAB_MAX_TENSORS = 64;
struct SlabBatch {
    semseg_slab_tensor t[SLAB_MAX_TENSORS];
};
__global__ __launch_bounds__(256) void reduce_slabs_multi_kernel(const SlabBatch b) {
    const semseg_slab_tensor t = b.t[blockIdx.y];
    const size_t total = (size_t)t.numel;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool vec = (total % 4 == 0) && (((reinterpret_cast<uintptr_t>(t.slabs) | reinterpret_cast<uintptr_t>(t.out)) & 15) == 0);
    if (vec) {
        const size_t quads = total >> 2;
        const float4* p4 = reinterpret_cast<const float4*>(t.slabs);
        float4* o4 = reinterpret_cast<float4*>(t.out);
        for (size_t i = tid; i < quads; i += stride) {
            float4 s = p4[i];
            for (int z = 1; z < t.splits; z += 4) {       // 4 loads in flight, additions in slab order
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (z + u < t.splits) v[u] = p4[(size_t)(z + u) * quads + i];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (z + u < t.splits) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            }
            o4[i] = s;
        }
    } else {
        for (size_t i = tid; i < total; i += stride) {
            float s = t.slabs[i];
            for (int z = 1; z < t.splits; ++z) s += t.slabs[(size_t)z * total + i];
            t.out[i] = s;
        }
    }
}
extern "C" int semseg_reduce_slabs_multi(const semseg_slab_tensor* tensors_host, int n, void* stream) {
    if (n < 0 || (n > 0 && !tensors_host)) return SEMSEG_EINVAL;
    for (int base = 0; base < n; base += SLAB_MAX_TENSORS) {
        const int cnt = n - base < SLAB_MAX_TENSORS ? n - base : SLAB_MAX_TENSORS;
        SlabBatch b;
        size_t most = 0;
        for (int i = 0; i < cnt; ++i) {
            const semseg_slab_tensor& t = tensors_host[base + i];
            if (!t.slabs || !t.out || t.numel <= 0 || t.splits < 1) return SEMSEG_EINVAL;
            b.t[i] = t;
            if ((size_t)t.numel > most) most = (size_t)t.numel;
        }
        int gx = (int)ceil_div_sz(most / 4 + 1, 256 * 4);          // ~4 float4 lanes per thread for the largest tensor of the batch
        gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
        hipLaunchKernelGGL(reduce_slabs_multi_kernel, dim3(gx, cnt), dim3(256), 0, (hipStream_t)stream, b);
        SEMSEG_LAUNCH_CHECK();
    }
    return 0;
}

static size_t split_conv_workspace_bytes(int sch, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0 || dil <= 0) return 0;
    const int OH = out_dim(H, R, stride, pad, dil), OW = out_dim(W, S, stride, pad, dil);
    if (OH <= 0 || OW <= 0) return 0;
    const int T = R * S;
    int t0 = -1, s0 = 0, t1 = -1, s1 = 0, t2 = -1, s2 = 0;
    lookup_plan(sch, 0, N, H, W, C, K, R, S, stride, pad, dil, &t0, &s0);
    lookup_plan(sch, 1, N, H, W, C, K, R, S, stride, pad, dil, &t1, &s1);
    lookup_plan(sch, 2, N, H, W, C, K, R, S, stride, pad, dil, &t2, &s2);
    const size_t a = gemm_workspace_bytes(sch, N * OH * OW, K, C, T, t0, s0);      // forward
    const size_t b = gemm_workspace_bytes(sch, N * H * W, C, K, T, t1, s1);        // dgrad
    const size_t c = wgrad_split_workspace_bytes(N * OH * OW, K, C, T, t2, s2);    // wgrad
    const size_t d = (size_t)256 * K * sizeof(double);                    // bias gradient partials
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    return m > d ? m : d;
}

extern "C" size_t semseg_conv2d_s3_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                                                   int dil) {
    return split_conv_workspace_bytes(SchS3::ID, N, H, W, C, K, R, S, stride, pad, dil);
}
extern "C" size_t semseg_conv2d_h2_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                                                   int dil) {
    return split_conv_workspace_bytes(SchH2::ID, N, H, W, C, K, R, S, stride, pad, dil);
}
