// Multi-tensor preparation of the convolution weights for the h2 path: for EVERY conv weight of the model, in two
// launches per 64 tensors, produce
//   * the KRSC split planes (rows (k, tap), columns c)  -- the B operand of the forward implicit GEMM
//   * the CRSK split planes (rows (c, tap), columns k)  -- the B operand of the data gradient
// with one shared per-tensor exponent.  Replaces, per conv and per training step, 2 x (absmax + split_h2) launches and
// one weight_krsc_to_crsk launch (63 convs x 5 = 315 launches of ~6 us each in R50dilated+PPM_deepsup) -- the weights
// only change in the optimiser step (train.py:117-126), so the engine calls this once after semseg_sgd_step.
// Results are bit-identical to semseg_split_h2 applied to the weight and to its CRSK transpose (tests/test_gpu_ops.py).
#include "common.h"
#include "split_layout.h"

constexpr int WPREP_MAX_TENSORS = 48;      // 48 x 72 B of descriptors stay below the 4 KiB kernel-argument limit
constexpr int WPREP_CHUNKS = 256;          // absmax partial blocks per tensor (<= H2_MAX_PARTIALS)
constexpr int WPREP_SLOTS = 512;           // partial slots the split kernel reduces: 256 from wprep_absmax_kernel (the rest zero) or up to 512
                                           // from the fused SGD kernel (csrc/head.hip sgd_fused_kernel)

struct WPrepTensor {
    const float* w;        // [K][T][C] fp32
    uint16_t* krsc;        // h2 split buffer, rows K*T, channels C
    uint16_t* crsk;        // h2 split buffer, rows C*T, channels K
    int K, T, C;
    int tile_base;         // first flat tile id of this tensor
    uint16_t* wino;        // optional (T == 9): h2 split buffer of the Winograd-transformed weights, rows 16*K, channels C
    int wino_base;         // first flat block id of this tensor in wprep_wino_kernel
    uint16_t* wino_t;      // optional (T == 9): the same for the DATA GRADIENT (flipped taps, roles of C and K swapped): rows 16*C, channels K
    int wino_t_base;       // first flat block id of this tensor in wprep_wino_t_kernel
};
struct WPrepBatch {
    WPrepTensor t[WPREP_MAX_TENSORS];
    int n;
};

__host__ __device__ __forceinline__ uint32_t* wprep_partials(const WPrepTensor& t, size_t krsc_plane) {
    unsigned char* base = reinterpret_cast<unsigned char*>(t.krsc) + (size_t)H2_NP * krsc_plane * 2 + SPLIT_ZERO_TAIL_BYTES;
    return reinterpret_cast<uint32_t*>(base) + 64;      // 256 B after the exponent word
}

__global__ __launch_bounds__(256) void wprep_absmax_kernel(const WPrepBatch b, int pitch_pad) {
    const WPrepTensor t = b.t[blockIdx.y];
    const size_t total = (size_t)t.K * t.T * t.C;
    uint32_t m = 0;
    const size_t quads = total >> 2;                 // t.w is 16-byte aligned (checked on the host)
    const float4* w4 = reinterpret_cast<const float4*>(t.w);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (size_t)WPREP_CHUNKS * 256) {
        const float4 v = w4[i];
        m = max(m, max(max(absbits(v.x), absbits(v.y)), max(absbits(v.z), absbits(v.w))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(total & 3)) m = max(m, absbits(t.w[(quads << 2) + threadIdx.x]));
    m = block_max_u32(m);
    const int Cp = (t.C + 31) & ~31;
    const int pitch = ((Cp * 2) % 2048 == 0) ? Cp + pitch_pad : Cp;
    if (threadIdx.x == 0) {
        uint32_t* slots = wprep_partials(t, (size_t)t.K * t.T * pitch);
        slots[blockIdx.x] = m;
        slots[WPREP_CHUNKS + blockIdx.x] = 0u;      // the upper half of the WPREP_SLOTS slots (only the fused SGD kernel fills it)
    }
}

// one block = one 64(k) x 64(c) tile of one tap of one tensor
__global__ __launch_bounds__(256) void wprep_split_kernel(const WPrepBatch b, int pitch_pad) {
    __shared__ _Float16 tr[2][64][66];      // [part][c][k]: CRSK staging (transposed)
    int ti = 0;
    while (ti + 1 < b.n && (int)blockIdx.x >= b.t[ti + 1].tile_base) ++ti;      // block-uniform
    const WPrepTensor t = b.t[ti];
    const int K = t.K, T = t.T, C = t.C;
    const int Cp = (C + 31) & ~31, Kp = (K + 31) & ~31;
    const int pitch_c = ((Cp * 2) % 2048 == 0) ? Cp + pitch_pad : Cp;
    const int pitch_k = ((Kp * 2) % 2048 == 0) ? Kp + pitch_pad : Kp;
    const size_t plane_krsc = (size_t)K * T * pitch_c;
    const size_t plane_crsk = (size_t)C * T * pitch_k;
    const int tiles_c = (Cp + 63) >> 6;
    int tile = (int)blockIdx.x - t.tile_base;
    const int tap = tile % T;
    tile /= T;
    const int ct = tile % tiles_c;
    const int kt = tile / tiles_c;
    const int k0 = kt * 64, c0 = ct * 64;

    // exponent from the partial maxima of wprep_absmax_kernel
    const uint32_t* partial = wprep_partials(t, plane_krsc);
    static_assert(WPREP_SLOTS == 2 * WPREP_CHUNKS && WPREP_CHUNKS == 256 && WPREP_SLOTS <= H2_MAX_PARTIALS, "two partials per thread");
    uint32_t m = max(partial[threadIdx.x], partial[WPREP_CHUNKS + threadIdx.x]);
    m = block_max_u32(m);
    const int ex = h2_exponent(m);
    const float sc = pow2i(ex);
    if ((int)blockIdx.x == t.tile_base) {       // first tile of the tensor: zero tails + exponent words of both buffers
        unsigned char* ka = reinterpret_cast<unsigned char*>(t.krsc) + (size_t)H2_NP * plane_krsc * 2;
        unsigned char* ca = reinterpret_cast<unsigned char*>(t.crsk) + (size_t)H2_NP * plane_crsk * 2;
        if (threadIdx.x < SPLIT_ZERO_TAIL_BYTES / 16) {
            reinterpret_cast<uint4*>(ka)[threadIdx.x] = make_uint4(0, 0, 0, 0);
            reinterpret_cast<uint4*>(ca)[threadIdx.x] = make_uint4(0, 0, 0, 0);
        }
        if (threadIdx.x == 0) {
            *reinterpret_cast<int*>(ka + SPLIT_ZERO_TAIL_BYTES) = ex;
            *reinterpret_cast<int*>(ca + SPLIT_ZERO_TAIL_BYTES) = ex;
        }
    }

    // load + split: thread = 4 consecutive channels of one k row, 4 passes of 16 rows
    const int cg = threadIdx.x & 15;
    const int r0 = threadIdx.x >> 4;
    const int c = c0 + cg * 4;
    // all 4 passes' weights are REQUESTED first (unconditional loads at clamped indices, one 16-byte load per pass where the
    // channel count allows), then consumed: as 16 conditional scalar loads each waited for its own round trip (tools/isa_wait_lint.py)
    float wv[4][4];
    const bool vec4 = (C & 3) == 0;               // block-uniform: rows of 4 channels are 16-byte aligned
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int k = k0 + r0 + ps * 16;
        const int kc = k < K ? k : K - 1;
        const size_t row = ((size_t)kc * T + tap) * C;
        if (vec4) {
            const int c4 = c < C ? c : C - 4;
            const float4 q = *reinterpret_cast<const float4*>(t.w + row + c4);
            wv[ps][0] = q.x; wv[ps][1] = q.y; wv[ps][2] = q.z; wv[ps][3] = q.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) wv[ps][e] = t.w[row + (c + e < C ? c + e : C - 1)];
        }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) asm volatile("" : "+v"(wv[ps][0]), "+v"(wv[ps][1]), "+v"(wv[ps][2]), "+v"(wv[ps][3]));
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int kl = r0 + ps * 16;
        const int k = k0 + kl;
        f16x4 h0, h1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = (k < K && c + e < C) ? wv[ps][e] : 0.f;
            _Float16 a, r;
            h2_split_of(v * sc, a, r);
            h0[e] = a;
            h1[e] = r;
            tr[0][cg * 4 + e][kl] = a;
            tr[1][cg * 4 + e][kl] = r;
        }
        if (k < K && c < Cp) {                  // Cp is a multiple of 4: the 4 columns are all inside or all outside
            const size_t o = ((size_t)k * T + tap) * pitch_c + c;
            *reinterpret_cast<f16x4*>(t.krsc + o) = h0;
            *reinterpret_cast<f16x4*>(t.krsc + plane_krsc + o) = h1;
        }
    }
    __syncthreads();
    // CRSK rows: thread = 16 consecutive k of one channel row
    const int cl = threadIdx.x >> 2;
    const int kq = (threadIdx.x & 3) * 16;
    const int cc = c0 + cl;
    if (cc < C) {
#pragma unroll
        for (int part = 0; part < 2; ++part) {
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int kk = k0 + kq + hlf * 8;
                if (kk < Kp) {                  // Kp is a multiple of 8
                    f16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = tr[part][cl][kq + hlf * 8 + e];
                    *reinterpret_cast<f16x8*>(t.crsk + (size_t)part * plane_crsk + ((size_t)cc * T + tap) * pitch_k + kk) = v;
                }
            }
        }
    }
}

// U = G g G^T of every 3x3 weight that asked for it (winograd.hip): one thread = 4 channels of one k; |U| <= 2.25 max|g|
// (G has rows of absolute sum <= 1.5), so the exponent comes from the absmax partials that are already there.
__global__ __launch_bounds__(256) void wprep_wino_kernel(const WPrepBatch b, int pitch_pad) {
    int ti = -1;
    for (int i = 0; i < b.n; ++i)
        if (b.t[i].wino && (int)blockIdx.x >= b.t[i].wino_base) ti = i;         // block-uniform; bases ascend
    if (ti < 0) return;
    const WPrepTensor t = b.t[ti];
    const int K = t.K, C = t.C;
    const int Cp = (C + 31) & ~31;
    const int pitch = ((Cp * 2) % 2048 == 0) ? Cp + pitch_pad : Cp;
    const size_t plane_krsc = (size_t)K * t.T * pitch;
    const size_t plane = (size_t)16 * K * pitch;
    const uint32_t* partial = wprep_partials(t, plane_krsc);
    uint32_t m = max(partial[threadIdx.x], partial[WPREP_CHUNKS + threadIdx.x]);       // WPREP_SLOTS slots, two per thread
    m = block_max_u32(m);
    const float bound = 2.25f * __uint_as_float(m);
    const int ex = h2_exponent((m >> 23) == 255 ? m : __float_as_uint(bound));
    const float sc = pow2i(ex);
    if ((int)blockIdx.x == t.wino_base) {
        unsigned char* tail = reinterpret_cast<unsigned char*>(t.wino) + (size_t)H2_NP * plane * 2;
        if (threadIdx.x < SPLIT_ZERO_TAIL_BYTES / 16) reinterpret_cast<uint4*>(tail)[threadIdx.x] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) *reinterpret_cast<int*>(tail + SPLIT_ZERO_TAIL_BYTES) = ex;
    }
    const int G4 = Cp >> 2;
    const size_t item = (size_t)((int)blockIdx.x - t.wino_base) * 256 + threadIdx.x;
    if (item >= (size_t)K * G4) return;
    const int k = (int)(item / G4);
    const int c = (int)(item - (size_t)k * G4) << 2;
    float g[3][3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) g[r][q][e] = (c + e < C) ? t.w[((size_t)k * 9 + r * 3 + q) * C + c + e] : 0.f;
    // G g (rows), then (.) G^T (columns):  u0 = g0, u1 = (g0 + g1 + g2)/2, u2 = (g0 - g1 + g2)/2, u3 = g2
    float a[4][3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float g0 = g[0][q][e], g1 = g[1][q][e], g2 = g[2][q][e];
            a[0][q][e] = g0;
            a[1][q][e] = 0.5f * (g0 + g1 + g2);
            a[2][q][e] = 0.5f * (g0 - g1 + g2);
            a[3][q][e] = g2;
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float u[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = a[i][0][e], a1 = a[i][1][e], a2 = a[i][2][e];
            u[0][e] = a0;
            u[1][e] = 0.5f * (a0 + a1 + a2);
            u[2][e] = 0.5f * (a0 - a1 + a2);
            u[3][e] = a2;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f16x4 p0, p1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 hi, lo;
                h2_split_of(u[j][e] * sc, hi, lo);
                p0[e] = hi;
                p1[e] = lo;
            }
            const size_t off = ((size_t)(i * 4 + j) * K + k) * pitch + c;
            *reinterpret_cast<f16x4*>(t.wino + off) = p0;
            *reinterpret_cast<f16x4*>(t.wino + plane + off) = p1;
        }
    }
}

// U' = G g' G^T for the data gradient in the Winograd domain: dx = conv(dz, g') with g'[c][r][s][k] = g[k][2-r][2-s][c] (the
// flipped taps, channels and filters swapped) is a 3x3 stride-1 convolution of the same geometry, so it runs through the same
// input transform / batched GEMM / output transform as the forward.  Read from the CRSK planes this launch sequence has just
// written (rows (c, tap), k contiguous: coalesced; p0 + p1 is the weight to 22 bits, the precision the direct data gradient
// multiplies with), transformed in the scaled domain with a factor 1/4 (|U'| <= 2.25 max|g|: stays inside fp16), exponent = e - 2.
// One thread = 4 filters k of one channel c.
__global__ __launch_bounds__(256) void wprep_wino_t_kernel(const WPrepBatch b, int pitch_pad) {
    int ti = -1;
    for (int i = 0; i < b.n; ++i)
        if (b.t[i].wino_t && (int)blockIdx.x >= b.t[i].wino_t_base) ti = i;     // block-uniform; bases ascend
    if (ti < 0) return;
    const WPrepTensor t = b.t[ti];
    const int K = t.K, C = t.C;
    const int Kp = (K + 31) & ~31;
    const int pitch_k = ((Kp * 2) % 2048 == 0) ? Kp + pitch_pad : Kp;
    const size_t plane_crsk = (size_t)C * 9 * pitch_k;
    const size_t plane = (size_t)16 * C * pitch_k;
    const unsigned char* ctail = reinterpret_cast<const unsigned char*>(t.crsk) + (size_t)H2_NP * plane_crsk * 2;
    const int ex = *reinterpret_cast<const int*>(ctail + SPLIT_ZERO_TAIL_BYTES);
    if ((int)blockIdx.x == t.wino_t_base) {
        unsigned char* tail = reinterpret_cast<unsigned char*>(t.wino_t) + (size_t)H2_NP * plane * 2;
        if (threadIdx.x < SPLIT_ZERO_TAIL_BYTES / 16) reinterpret_cast<uint4*>(tail)[threadIdx.x] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) *reinterpret_cast<int*>(tail + SPLIT_ZERO_TAIL_BYTES) = ex - 2;
    }
    const int G4 = Kp >> 2;
    const size_t item = (size_t)((int)blockIdx.x - t.wino_t_base) * 256 + threadIdx.x;
    if (item >= (size_t)C * G4) return;
    const int c = (int)(item / G4);
    const int k = (int)(item - (size_t)c * G4) << 2;
    float g[3][3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const size_t o = ((size_t)c * 9 + (8 - (r * 3 + q))) * pitch_k + k;         // flipped tap
            const f16x4 p0 = *reinterpret_cast<const f16x4*>(t.crsk + o);
            const f16x4 p1 = *reinterpret_cast<const f16x4*>(t.crsk + plane_crsk + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[r][q][e] = (k + e < K) ? ((float)p0[e] + (float)p1[e]) * 0.25f : 0.f;
        }
    float a[4][3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float g0 = g[0][q][e], g1 = g[1][q][e], g2 = g[2][q][e];
            a[0][q][e] = g0;
            a[1][q][e] = 0.5f * (g0 + g1 + g2);
            a[2][q][e] = 0.5f * (g0 - g1 + g2);
            a[3][q][e] = g2;
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float u[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = a[i][0][e], a1 = a[i][1][e], a2 = a[i][2][e];
            u[0][e] = a0;
            u[1][e] = 0.5f * (a0 + a1 + a2);
            u[2][e] = 0.5f * (a0 - a1 + a2);
            u[3][e] = a2;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f16x4 p0, p1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 hi, lo;
                h2_split_of(u[j][e], hi, lo);
                p0[e] = hi;
                p1[e] = lo;
            }
            const size_t off = ((size_t)(i * 4 + j) * C + c) * pitch_k + k;
            *reinterpret_cast<f16x4*>(t.wino_t + off) = p0;
            *reinterpret_cast<f16x4*>(t.wino_t + plane + off) = p1;
        }
    }
}

// device address of the WPREP_SLOTS partial-maximum slots of a conv weight's KRSC plane buffer (what the fused SGD kernel fills)
extern "C" void* semseg_weights_absmax_slots(void* krsc_planes, int K, int T, int C) {
    if (!krsc_planes || K <= 0 || T <= 0 || C <= 0) return nullptr;
    WPrepTensor t = {};
    t.krsc = (uint16_t*)krsc_planes;
    return wprep_partials(t, (size_t)K * T * split_pitch(C));
}

static int weights_prepare(const semseg_wprep_tensor* tensors_host, int n, int have_absmax, void* stream);
extern "C" int semseg_weights_prepare_h2(const semseg_wprep_tensor* tensors_host, int n, void* stream) {
    return weights_prepare(tensors_host, n, 0, stream);
}
// have_absmax != 0: the partial maxima of EVERY tensor are in place already (semseg_sgd_step_fused wrote them while it updated the
// weights): the pass over the weights that only finds their maximum is not launched
extern "C" int semseg_weights_prepare_h2_after_sgd(const semseg_wprep_tensor* tensors_host, int n, int have_absmax, void* stream) {
    return weights_prepare(tensors_host, n, have_absmax, stream);
}
static int weights_prepare(const semseg_wprep_tensor* tensors_host, int n, int have_absmax, void* stream) {
    if (!tensors_host || n < 0) return SEMSEG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int pitch_pad = split_pitch(1024) - 1024;      // the skew split_layout.h applies to power-of-two pitches
    for (int base = 0; base < n; base += WPREP_MAX_TENSORS) {
        WPrepBatch b;
        b.n = min(WPREP_MAX_TENSORS, n - base);
        int tiles = 0, wino_blocks = 0, wino_t_blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            const semseg_wprep_tensor& s = tensors_host[base + i];
            if (!s.w || !s.krsc || !s.crsk || s.K <= 0 || s.T <= 0 || s.C <= 0) return SEMSEG_EINVAL;
            if (!aligned16(s.krsc) || !aligned16(s.crsk) || !aligned16(s.w)) return SEMSEG_EINVAL;
            b.t[i].w = s.w;
            b.t[i].krsc = (uint16_t*)s.krsc;
            b.t[i].crsk = (uint16_t*)s.crsk;
            b.t[i].K = s.K;
            b.t[i].T = s.T;
            b.t[i].C = s.C;
            b.t[i].tile_base = tiles;
            tiles += ((s.K + 63) / 64) * ((round_up32(s.C) + 63) / 64) * s.T;
            b.t[i].wino = nullptr;
            b.t[i].wino_base = wino_blocks;
            if (s.wino) {
                if (s.T != 9 || !aligned16(s.wino)) return SEMSEG_EINVAL;
                b.t[i].wino = (uint16_t*)s.wino;
                wino_blocks += (int)ceil_div_sz((size_t)s.K * (round_up32(s.C) / 4), 256);
            }
            b.t[i].wino_t = nullptr;
            b.t[i].wino_t_base = wino_t_blocks;
            if (s.wino_t) {
                if (s.T != 9 || !aligned16(s.wino_t)) return SEMSEG_EINVAL;
                b.t[i].wino_t = (uint16_t*)s.wino_t;
                wino_t_blocks += (int)ceil_div_sz((size_t)s.C * (round_up32(s.K) / 4), 256);
            }
        }
        if (b.n == 0) break;
        if (!have_absmax) {
            hipLaunchKernelGGL(wprep_absmax_kernel, dim3(WPREP_CHUNKS, b.n), dim3(256), 0, st, b, pitch_pad);
            SEMSEG_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(wprep_split_kernel, dim3(tiles), dim3(256), 0, st, b, pitch_pad);
        SEMSEG_LAUNCH_CHECK();
        if (wino_blocks > 0) {
            hipLaunchKernelGGL(wprep_wino_kernel, dim3(wino_blocks), dim3(256), 0, st, b, pitch_pad);
            SEMSEG_LAUNCH_CHECK();
        }
        if (wino_t_blocks > 0) {            // after wprep_split_kernel: reads the CRSK planes it wrote
            hipLaunchKernelGGL(wprep_wino_t_kernel, dim3(wino_t_blocks), dim3(256), 0, st, b, pitch_pad);
            SEMSEG_LAUNCH_CHECK();
        }
    }
    return 0;
}
