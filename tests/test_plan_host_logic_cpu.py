"""Host side of the launch plans in the BUILT library, without a GPU: semseg_conv2d_h2_set_plan's refusals, the plan queries the
Python layer decides with (which weight gradients are batched, how many slabs a plan leaves), and the batched weight-gradient
entry refusing its arguments BEFORE it reaches a HIP call.  Nothing here launches or allocates on a device."""
import ctypes

import pytest

from mit_semseg import _native

EINVAL = -1


@pytest.fixture()
def L():
    return _native.lib()


def unpin(L, pas, geom):
    L.semseg_conv2d_h2_set_plan(pas, *geom, -1, 0)


def test_wgrad_plan_query_follows_the_pinned_plan(L):
    geom = (2, 64, 64, 64, 64, 3, 3, 1, 1, 1)               # N H W C K R S stride pad dil
    try:
        assert L.semseg_conv2d_wgrad_tile_h2(*geom) in (0, 1)                      # the library's heuristic: register-staged tiles
        for tile, split in ((1, 4), (5, 2), (10, 8)):
            assert L.semseg_conv2d_h2_set_plan(2, *geom, tile, split) == 0
            assert L.semseg_conv2d_wgrad_tile_h2(*geom) == tile
            nbytes = L.semseg_conv2d_wgrad_slabs_bytes(*geom)
            per = 64 * 3 * 3 * 64 * 4
            assert nbytes % per == 0 and 1 <= nbytes // per <= split              # the plan's chunks (never more than asked for)
    finally:
        unpin(L, 2, geom)
    assert L.semseg_conv2d_wgrad_tile_h2(0, 64, 64, 64, 64, 3, 3, 1, 1, 1) == EINVAL
    assert L.semseg_conv2d_wgrad_slabs_bytes(2, 64, 64, 64, 64, 3, 3, 0, 1, 1) == 0


def test_set_plan_refuses_tiles_a_geometry_cannot_take(L):
    # the all-taps weight-gradient kernel (wtile 10): 3x3, stride 1, pad == dil, output rows of whole 32-pixel chunks
    for bad in ((1, 24, 24, 64, 64, 3, 3, 1, 1, 1), (1, 32, 32, 64, 64, 3, 3, 2, 1, 1), (1, 32, 32, 64, 64, 1, 1, 1, 0, 1)):
        assert L.semseg_conv2d_h2_set_plan(2, *bad, 10, 1) != 0, bad
    ok = (1, 32, 32, 64, 64, 3, 3, 1, 2, 2)
    try:
        assert L.semseg_conv2d_h2_set_plan(2, *ok, 10, 1) == 0
    finally:
        unpin(L, 2, ok)
    # 64-deep k-tiles (22 - 24): forward / data gradient take any reduction since round 6 (an odd count of 32-channel chunks: the
    # missing half of the last 64-channel chunk is fetched as zeros, igemm_dma64_kernel); the batched Winograd GEMM (pass 3)
    # still wants whole 64-channel chunks
    for tile in (22, 23, 24):
        for geom in ((2, 16, 16, 96, 64, 1, 1, 1, 0, 1), (2, 16, 16, 128, 64, 1, 1, 1, 0, 1)):      # C = 96: one and a half chunks
            for pass_id in (0, 1):
                try:
                    assert L.semseg_conv2d_h2_set_plan(pass_id, *geom, tile, 1) == 0, (tile, geom)
                finally:
                    unpin(L, pass_id, geom)
    wino_odd, wino_even = (128, 1, 1, 96, 64, 3, 3, 1, 1, 1), (128, 1, 1, 128, 64, 3, 3, 1, 1, 1)
    assert L.semseg_conv2d_h2_set_plan(3, *wino_odd, 22, 1) != 0
    try:
        assert L.semseg_conv2d_h2_set_plan(3, *wino_even, 22, 1) == 0
    finally:
        unpin(L, 3, wino_even)
    # tiles beyond the tables, unknown passes
    geom = (2, 16, 16, 64, 64, 3, 3, 1, 1, 1)
    assert L.semseg_conv2d_h2_set_plan(0, *geom, 26, 1) == 0 and L.semseg_conv2d_h2_set_plan(0, *geom, -1, 0) == 0
    assert L.semseg_conv2d_h2_set_plan(0, *geom, 27, 1) != 0
    assert L.semseg_conv2d_h2_set_plan(2, *geom, 15, 1) != 0
    assert L.semseg_conv2d_h2_set_plan(7, *geom, 0, 1) != 0


def test_batched_weight_gradients_refuse_before_any_launch(L):
    """no GPU in this process: an argument error must come back as SEMSEG_EINVAL from the host checks, not as a HIP error from a
    launch (the plan check of EVERY problem runs before the first launch)"""
    assert L.semseg_conv2d_wgrad_multi_h2(None, 0, None) == 0
    assert L.semseg_conv2d_wgrad_multi_h2(None, 3, None) == EINVAL
    arr = (_native.WgradProblem * 2)()
    assert L.semseg_conv2d_wgrad_multi_h2(arr, -1, None) == EINVAL
    geom = (2, 16, 16, 64, 64, 3, 3, 1, 1, 1)
    fake = 1 << 20                                           # 16-byte aligned, never dereferenced on the host
    for q in arr:
        q.xs, q.dys, q.slabs, q.slabs_bytes = fake, fake, fake, 1 << 30
        q.N, q.H, q.W, q.C, q.K, q.R, q.S, q.stride, q.pad, q.dil = geom
    try:
        assert L.semseg_conv2d_h2_set_plan(2, *geom, 0, 1) == 0            # the 128 x 128 tile: not batchable
        assert L.semseg_conv2d_wgrad_multi_h2(arr, 2, None) == EINVAL
        assert L.semseg_conv2d_h2_set_plan(2, *geom, 1, 2) == 0
        arr[1].xs = fake + 4                                               # misaligned planes
        assert L.semseg_conv2d_wgrad_multi_h2(arr, 2, None) == EINVAL
        arr[1].xs = fake
        arr[1].slabs_bytes = 16                                            # slabs too small for the plan's chunks
        assert L.semseg_conv2d_wgrad_multi_h2(arr, 2, None) == -2          # SEMSEG_EWORKSPACE
        assert arr[0].splits == 2                                          # problem 0 had been planned: 2 chunks
    finally:
        unpin(L, 2, geom)
    st = _native.SlabTensor * 1
    assert L.semseg_reduce_slabs_multi(None, 0, None) == 0
    assert L.semseg_reduce_slabs_multi(None, 2, None) == EINVAL
    bad = st()
    bad[0].slabs, bad[0].out, bad[0].numel, bad[0].splits = fake, fake, 0, 1
    assert L.semseg_reduce_slabs_multi(bad, 1, None) == EINVAL
