// Which tile a block computes: the block-order arithmetic of the GEMM-shaped kernels (conv_split.hip, conv_igemm.hip,
// conv_wgrad.hip) as plain inline functions, so that tests/native/block_order_emulate.cpp enumerates whole grids with the SAME
// code on the host (tests/test_block_order_cpu.py: every (tile, split, batch) exactly once, and the XCD properties below).
//
// MI355X deals the blocks of a launch to its 8 XCDs round-robin in the order  b = x + gridDim.x * (y + gridDim.y * z),
// and each XCD has its own 4 MB L2.  xcd_remap gives the blocks of one XCD (b, b + 8, b + 16, ...) CONSECUTIVE logical ids, so
// "neighbouring logical ids" means "same L2"; the functions below decide what neighbours share.
#pragma once

#ifndef SEMSEG_HD
#ifdef __HIPCC__
#define SEMSEG_HD __host__ __device__ __forceinline__
#else
#define SEMSEG_HD static inline
#endif
#endif

// bijection of [0, nblocks): hardware order b -> logical id; XCD x (b % 8 == x) owns one contiguous range of logical ids
SEMSEG_HD int xcd_remap(int b, int nblocks) {
    const int xcd = b & 7, q = nblocks >> 3, r = nblocks & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

// Forward / data-gradient GEMM (igemm_rs_kernel, igemm_dma_kernel): grid (tiles_m * tiles_n, splits, batches).
//   batch  slowest: the batches of a batched GEMM (the 16 Winograd positions) go to the XCDs WHOLE -- both operands of a batch
//          stay in one L2;
//   split  next (a split-K slice is a channel range of both operands);
//   inside: row tiles fastest (default) -- the blocks of an XCD share ONE weight slice and the halo rows of neighbouring pixel
//          tiles (3x3 convs) -- or, `tn_fast`, column tiles fastest -- all column tiles of a row tile side by side, the pixel rows
//          are fetched once and the small weight matrix streams through every L2 (1x1 convs; run_gemm decides).
struct GemmBlock { int tm, tn, z, batch; };
SEMSEG_HD GemmBlock gemm_block(int hw_linear, int tiles_m, int tiles_n, int splits, int batches, int tn_fast) {
    const int per_batch = tiles_m * tiles_n * splits;
    const int lin = xcd_remap(hw_linear, per_batch * batches);
    GemmBlock b;
    // one division instead of three for the plain launch (no batches, no split-K: most of a step's GEMMs) -- on the device every
    // division by a run-time divisor is a ~30-instruction dependent chain in front of the block's first DMA piece
    b.batch = batches > 1 ? lin / per_batch : 0;
    const int lid = lin - b.batch * per_batch;
    if (tn_fast) {
        const int tmz = lid / tiles_n;
        b.tn = lid - tmz * tiles_n;
        if (splits > 1) {
            b.z = tmz / tiles_m;
            b.tm = tmz - b.z * tiles_m;
        } else {
            b.z = 0;
            b.tm = tmz;
        }
    } else {
        const int tnz = lid / tiles_m;
        b.tm = lid - tnz * tiles_m;
        if (splits > 1) {
            b.z = tnz / tiles_n;
            b.tn = tnz - b.z * tiles_n;
        } else {
            b.z = 0;
            b.tn = tnz;
        }
    }
    return b;
}

// Fused Winograd GEMM + output transform (wino_fused_kernel): grid tiles_m * tiles_n, every block walks all 16 frequencies in
// step with its neighbours, so what an XCD's L2 has to hold per frequency is (row tiles resident) x BM x C of V + (column tiles
// resident) x BN x C of U.  An XCD runs ~32 blocks at a time: they are dealt as SR x SC = 4 x 8 super-tiles (12 operand tiles per
// 32 blocks instead of the 18 of a 16 x 2 strip); grids that the super-tile does not divide fall back to rows fastest.
struct WinoFusedBlock { int tm, tn; };
SEMSEG_HD WinoFusedBlock wino_fused_block(int hw_linear, int tiles_m, int tiles_n) {
    const int lin = xcd_remap(hw_linear, tiles_m * tiles_n);
    WinoFusedBlock b;
    const int SR = 4, SC = 8;
    if (tiles_m % SR == 0 && tiles_n % SC == 0) {
        const int sup = lin / (SR * SC), in = lin - sup * (SR * SC);
        const int sups_m = tiles_m / SR;
        b.tm = (sup % sups_m) * SR + in % SR;
        b.tn = (sup / sups_m) * SC + in / SR;
    } else {
        b.tm = lin % tiles_m;
        b.tn = lin / tiles_m;
    }
    return b;
}

// Weight-gradient kernels (wgrad_kernel, wgrad_dma_kernel): grid (ntiles, splits[, batches]); the remap runs over (tile, split)
// TOGETHER: an XCD receives whole row chunks z, and the tiles of a chunk -- the taps / channel blocks that read the same rows of
// x and dy -- meet in one L2.  (A remap of x alone is an XCD map only when gridDim.x % 8 == 0 and deals every chunk to all XCDs.)
struct WgradBlock { int tile, z; };
SEMSEG_HD WgradBlock wgrad_block(int hw_linear_xy, int ntiles, int splits) {
    const int lin = xcd_remap(hw_linear_xy, ntiles * splits);
    WgradBlock b;
    b.z = lin / ntiles;
    b.tile = lin - b.z * ntiles;
    return b;
}
