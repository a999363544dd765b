"""Block order of the GEMM-shaped kernels (csrc/block_order.h), enumerated on the host with the SAME functions the kernels call
(tests/native/block_order_emulate.cpp).  Correctness of a launch needs exactly one thing from this arithmetic -- every
(tile, split, batch) computed by exactly one block -- and its speed needs the XCD properties the header states (MI355X deals block b
of a launch to XCD b % 8; each XCD has its own L2)."""
import ctypes
import itertools
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def emu(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('block_order') / 'libblock_order_emulate.so')
    subprocess.run(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-I' + os.path.join(ROOT, 'semantic-segmentation-pytorch_amd', 'csrc'),
                    os.path.join(ROOT, 'tests', 'native', 'block_order_emulate.cpp'), '-o', out], check=True)
    return ctypes.CDLL(out)


def gemm_grid(emu, tiles_m, tiles_n, splits, batches, tn_fast):
    out = np.zeros((tiles_m * tiles_n * splits * batches, 4), dtype=np.int32)
    emu.emulate_gemm_grid(tiles_m, tiles_n, splits, batches, tn_fast, out.ctypes.data_as(ctypes.c_void_p))
    return out


def wgrad_grid(emu, ntiles, splits):
    out = np.zeros((ntiles * splits, 2), dtype=np.int32)
    emu.emulate_wgrad_grid(ntiles, splits, out.ctypes.data_as(ctypes.c_void_p))
    return out


GEMM_SHAPES = [(1, 1, 1, 1), (8, 2, 1, 16), (8, 16, 1, 16), (64, 4, 1, 1), (128, 4, 1, 1), (32, 8, 1, 1), (5, 3, 1, 1), (7, 1, 3, 1),
               (128, 1, 3, 1), (8, 16, 16, 1), (3, 5, 7, 2), (1, 9, 1, 16), (2048, 1, 1, 1), (17, 3, 2, 16)]


@pytest.mark.parametrize('tn_fast', [0, 1])
@pytest.mark.parametrize('shape', GEMM_SHAPES, ids=str)
def test_gemm_blocks_cover_every_tile_once(emu, shape, tn_fast):
    tiles_m, tiles_n, splits, batches = shape
    g = gemm_grid(emu, tiles_m, tiles_n, splits, batches, tn_fast)
    want = set(itertools.product(range(tiles_m), range(tiles_n), range(splits), range(batches)))
    got = [tuple(int(v) for v in row) for row in g]
    assert len(set(got)) == len(got) == len(want) and set(got) == want


@pytest.mark.parametrize('shape', [(9, 56), (36, 6), (36, 12), (64, 8), (16, 16), (24, 20), (1, 64), (256, 1), (18, 56), (5, 3), (1, 1)], ids=str)
def test_wgrad_blocks_cover_every_tile_once(emu, shape):
    ntiles, splits = shape
    g = wgrad_grid(emu, ntiles, splits)
    got = [tuple(int(v) for v in row) for row in g]
    assert len(set(got)) == len(got) and set(got) == set(itertools.product(range(ntiles), range(splits)))


def xcd_of(n):
    return np.arange(n) % 8


def test_whole_winograd_batches_land_on_one_xcd(emu):
    """layer4's batched GEMM (8 x 2 tiles, 16 positions = 256 blocks, one per CU): each XCD holds exactly two positions, all 16
    blocks of each -- before, the 16 blocks of a position sat on 8 XCDs (profiles/r4_ab_xcd_block_order.txt)"""
    g = gemm_grid(emu, 8, 2, 1, 16, 0)
    x = xcd_of(len(g))
    for batch in range(16):
        assert len(set(x[g[:, 3] == batch])) == 1
    for xcd in range(8):
        assert len(set(g[x == xcd, 3])) == 2


def test_column_tiles_of_a_row_tile_share_an_xcd_when_tn_fast(emu):
    """layer3's 1024 -> 256 conv as 64 x 64 tiles (128 row tiles x 4 column tiles): column tiles fastest puts the four blocks that
    read the same pixel rows on ONE XCD next to each other in dispatch order; rows fastest spreads them over four XCDs"""
    for tiles_m, tiles_n in ((128, 4), (64, 8), (32, 8)):
        fast, slow = gemm_grid(emu, tiles_m, tiles_n, 1, 1, 1), gemm_grid(emu, tiles_m, tiles_n, 1, 1, 0)
        x = xcd_of(len(fast))
        b = np.arange(len(fast))
        for tm in range(tiles_m):
            sel = fast[:, 0] == tm
            assert len(set(x[sel])) == 1
            assert b[sel].max() - b[sel].min() <= 8 * (tiles_n - 1)          # consecutive slots of that XCD
            assert len(set(x[slow[:, 0] == tm])) == min(tiles_n, 8)
        # and rows fastest keeps what it is for: one weight slice per XCD at a time
        for xcd in range(8):
            assert len(set(slow[x == xcd, 1])) <= -(-tiles_n // 8) + 1


def test_the_tiles_of_a_weight_gradient_chunk_share_an_xcd(emu):
    """stem_conv2's weight gradient (9 tap tiles x 56 row chunks; gridDim.x = 9 is no multiple of 8): all tap tiles of a chunk on one
    XCD, every XCD with 7 whole chunks"""
    for ntiles, splits in ((9, 56), (36, 6 * 8), (18, 56), (64, 8)):
        g = wgrad_grid(emu, ntiles, splits)
        x = xcd_of(len(g))
        for z in range(splits):
            assert len(set(x[g[:, 1] == z])) == 1
        assert all(len(set(g[x == xcd, 1])) == splits // 8 for xcd in range(8))
