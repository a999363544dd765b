// Layout of the 16-bit split planes shared by every kernel that produces or consumes them (conv_split.hip: the split
// kernels and the convolutions; bn.hip: the BN kernels that emit planes directly; weights_prep.hip).
//   [NP][rows][pitch] 16-bit elements | SPLIT_ZERO_TAIL_BYTES of zeros | h2 only: header (exponent word, partial maxima)
#pragma once
#include "common.h"
#include <stdlib.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

static inline int round_up32(int c) { return (c + 31) & ~31; }

// Row pitch (16-bit elements) of a split plane.  A power-of-two pitch of >= 2 KB walks the gathered rows of an operand
// tile (128 rows x 64 B) over a fraction of the L2 channels only, so such pitches are skewed by 256 B.
// (conv_last fwd 83 -> 104 TFLOP/s in the pad sweep of round 1)
inline int split_pitch(int C) {
    const int Cp = round_up32(C);
    return ((Cp * 2) % 2048 == 0) ? Cp + 128 : Cp;
}

// The planes are followed by SPLIT_ZERO_TAIL_BYTES of zeros: the LDS-DMA conv kernel points the lanes of padded /
// out-of-range rows at it (a direct-to-LDS load cannot select a zero afterwards).  h2 buffers continue with a header:
// int32 exponent e (the planes hold 2^e * x), then H2_MAX_PARTIALS uint32 partial |x| maxima (bit patterns).
#define SPLIT_ZERO_TAIL_BYTES 256
#define H2_MAX_PARTIALS 1024
#define H2_HDR_BYTES (256 + 4 * H2_MAX_PARTIALS)
#define H2_NP 2

static inline size_t h2_plane_elems(size_t rows, int C) { return rows * (size_t)split_pitch(C); }
static inline size_t h2_bytes(size_t rows, int C) {
    return (size_t)H2_NP * h2_plane_elems(rows, C) * sizeof(uint16_t) + SPLIT_ZERO_TAIL_BYTES + H2_HDR_BYTES;
}
// device address of the exponent word of an h2 split buffer
static inline const int* h2_exp_ptr(const void* xs, size_t rows, int C) {
    return reinterpret_cast<const int*>(reinterpret_cast<const unsigned char*>(xs) +
                                        (size_t)H2_NP * h2_plane_elems(rows, C) * sizeof(uint16_t) + SPLIT_ZERO_TAIL_BYTES);
}

__device__ __forceinline__ float pow2i(int e) { return __int_as_float((127 + e) << 23); }     // -126 <= e <= 127
__device__ __forceinline__ uint32_t absbits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

// exponent e with 2^e * max in [2^14, 2^15); clamped to +-100 (tensors whose max is below 2^-86 keep fewer bits);
// inf / NaN maxima: e = 0 (they propagate through fp16 as inf / NaN).  `maxbits` may be any UPPER BOUND of max|x|:
// a bound that is loose by 2^k only moves the fp16-subnormal floor of the low part up by k bits (conv_split.hip header).
__device__ __forceinline__ int h2_exponent(uint32_t maxbits) {
    const int ef = (int)(maxbits >> 23);
    if (ef == 255) return 0;
    return max(-100, min(100, 141 - ef));
}

// s (already scaled by 2^e) -> fp16 high part + fp16 residual
__device__ __forceinline__ void h2_split_of(float s, _Float16& h0, _Float16& h1) {
    h0 = (_Float16)s;
    h1 = (_Float16)(s - (float)h0);      // the subtraction is exact
}

// max over the block (any block size that is a multiple of 64, <= 1024); every thread gets the result
__device__ __forceinline__ uint32_t block_max_u32(uint32_t m) {
    __shared__ uint32_t red_u32__[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    const int nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red_u32__[threadIdx.x >> 6] = m;
    __syncthreads();
    uint32_t r = 0;
    for (int i = 0; i < nw; ++i) r = max(r, red_u32__[i]);
    __syncthreads();
    return r;
}
