// Probe of gfx950 ds_read_b64_tr_b16 and mfma_f32_32x32x16_bf16 operand layout (prints the lane/element maps).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void tr_probe(int* out, int stride_elems) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    // address pattern: within each 16-lane group, lane i points at row (i>>2), col chunk (i&3) of a [4][16] block;
    // group g uses rows 4g..4g+3 of a [16][stride] image
    const int g = l >> 4, i = l & 15;
    const int elem = (4 * g + (i >> 2)) * stride_elems + 4 * (i & 3);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + elem));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}

__global__ void mfma_probe(float* out) {
    // A[i][k] = i*100 + k ; B[k][j] = (k == ksel) ? 1 : 0 ... instead: run 16 MFMAs each with B = onehot(k==kk, j==0)
    const int l = threadIdx.x;
    for (int kk = 0; kk < 16; ++kk) {
        bf8 a, b;
        for (int e = 0; e < 8; ++e) {
            a[e] = (__bf16)(float)((l >> 5) * 8 + e + 1);          // value = assumed k index + 1 (row independent)
            b[e] = (__bf16)(((l & 31) == 0 && ((l >> 5) * 8 + e) == kk) ? 1.0f : 0.0f);
        }
        f16v c = {0};
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        // D[i][0] = sum_k A[i][k] B[k][0] = A[i][kk]; lane 0 reg 0 holds D[row 0][col 0]
        if (l == 0) out[kk] = c[0];
    }
}

int main() {
    int* d; hipMalloc(&d, 64 * 4 * 4);
    int h[256];
    for (int stride : {16, 64}) {
        tr_probe<<<1, 64>>>(d, stride);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("tr16_b64 stride=%d: lane -> 4 values (element index = row*stride+col)\n", stride);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" r%d c%2d |", h[l * 4 + j] / stride, h[l * 4 + j] % stride);
            printf("\n");
        }
    }
    float* f; hipMalloc(&f, 64); float hf[16];
    mfma_probe<<<1, 64>>>(f);
    hipMemcpy(hf, f, sizeof(hf), hipMemcpyDeviceToHost);
    printf("mfma 32x32x16 bf16: B onehot at assumed k=kk -> D[0][0] (expect kk+1 if k = 8*(lane>>5)+e on both operands):");
    for (int k = 0; k < 16; ++k) printf(" %g", hf[k]);
    printf("\n");
    return 0;
}
