// TEST INFRASTRUCTURE: host enumeration of whole launch grids through the block-order functions the GEMM-shaped kernels call
// (csrc/block_order.h), for tests/test_block_order_cpu.py.  Never shipped, never loaded by the product.
#include "block_order.h"

// one row per block in HARDWARE order b = x + gx * (y + gy * z): out[b] = {tm, tn, split, batch}
extern "C" void emulate_gemm_grid(int tiles_m, int tiles_n, int splits, int batches, int tn_fast, int* out) {
    const int gx = tiles_m * tiles_n, gy = splits, gz = batches;
    for (int z = 0; z < gz; ++z)
        for (int y = 0; y < gy; ++y)
            for (int x = 0; x < gx; ++x) {
                const int b = x + gx * (y + gy * z);
                const GemmBlock g = gemm_block(b, tiles_m, tiles_n, splits, batches, tn_fast);
                out[4 * b + 0] = g.tm; out[4 * b + 1] = g.tn; out[4 * b + 2] = g.z; out[4 * b + 3] = g.batch;
            }
}

// out[b] = {tile, split} for b = x + ntiles * y
extern "C" void emulate_wgrad_grid(int ntiles, int splits, int* out) {
    for (int y = 0; y < splits; ++y)
        for (int x = 0; x < ntiles; ++x) {
            const int b = x + ntiles * y;
            const WgradBlock w = wgrad_block(b, ntiles, splits);
            out[2 * b + 0] = w.tile; out[2 * b + 1] = w.z;
        }
}
