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

constexpr int WPREP_MAX_TENSORS = 64;
constexpr int WPREP_CHUNKS = 256;          // absmax partial blocks per tensor (<= H2_MAX_PARTIALS)

struct WPrepTensor {
    const float* w;        // [K][T][C] fp32
    uint16_t* krsc;        // h2 split buffer, rows K*T, channels C
    uint16_t* crsk;        // h2 split buffer, rows C*T, channels K
    int K, T, C;
    int tile_base;         // first flat tile id of this tensor
};
struct WPrepBatch {
    WPrepTensor t[WPREP_MAX_TENSORS];
    int n;
};

__device__ __forceinline__ uint32_t* wprep_partials(const WPrepTensor& t, size_t krsc_plane) {
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
    if (threadIdx.x == 0) wprep_partials(t, (size_t)t.K * t.T * pitch)[blockIdx.x] = m;
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
    static_assert(WPREP_CHUNKS <= 256 && WPREP_CHUNKS <= H2_MAX_PARTIALS, "one partial per thread");
    uint32_t m = threadIdx.x < WPREP_CHUNKS ? partial[threadIdx.x] : 0u;
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
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int kl = r0 + ps * 16;
        const int k = k0 + kl;
        f16x4 h0, h1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = 0.f;
            if (k < K && c + e < C) v = t.w[((size_t)k * T + tap) * C + c + e];
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

extern "C" int semseg_weights_prepare_h2(const semseg_wprep_tensor* tensors_host, int n, void* stream) {
    if (!tensors_host || n < 0) return SEMSEG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int pitch_pad = split_pitch(1024) - 1024;      // the skew in effect (SEMSEG_S3_PITCH_PAD)
    for (int base = 0; base < n; base += WPREP_MAX_TENSORS) {
        WPrepBatch b;
        b.n = min(WPREP_MAX_TENSORS, n - base);
        int tiles = 0;
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
        }
        if (b.n == 0) break;
        hipLaunchKernelGGL(wprep_absmax_kernel, dim3(WPREP_CHUNKS, b.n), dim3(256), 0, st, b, pitch_pad);
        SEMSEG_LAUNCH_CHECK();
        hipLaunchKernelGGL(wprep_split_kernel, dim3(tiles), dim3(256), 0, st, b, pitch_pad);
        SEMSEG_LAUNCH_CHECK();
    }
    return 0;
}
