"""TEST INFRASTRUCTURE (run by tests/test_gpu_ddp.py::test_peer_exchange_world8_in_one_process in a child process with
GPU_MAX_HW_QUEUES=16): the one-node peer exchange (csrc/peer.hip, csrc/peer_dev.h) at its FULL world of 8 ranks without 8 GPUs --
eight contexts of this process on the one GPU (semseg_peer_attach_local), one HIP stream each (every stream on its own
hardware queue: a rank's kernel spins until the other seven have pushed, so the eight kernels must run concurrently).
Checks: (1) the stand-alone exchange over 13 exchanges (> 4 slots, > 8 lanes) returns the RANK-ORDERED sum bit-identically on
all eight ranks (an order-sensitive payload: a permuted order gives different bits); (2) the fused BN forward finish kernels
(semseg_bn_fwd_stats_fused_peer, multi-block grids) on eight unequal shards produce on every rank exactly the coefficients the
unfused entry points produce from the rank-ordered host sum of the shards' statistics; the backward pair likewise;
(3) a rank that never shows up: the other seven poison their result with NaN and raise the status within the timeout."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

W = 8
vp = ctypes.c_void_p


def P(t):
    return vp(t.data_ptr()) if t is not None else vp(0)


def make_world(L, cap, timeout_s, world=W):
    peers = [vp() for _ in range(world)]
    for r in range(world):
        assert L.semseg_peer_create(r, world, cap, timeout_s, ctypes.byref(peers[r])) == 0
    for r in range(world):
        for o in range(world):
            if o != r:
                assert L.semseg_peer_attach_local(peers[r], o, peers[o]) == 0
    return peers


def main():
    from mit_semseg import _native
    L = _native.lib()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    assert L.semseg_peer_max_world() >= W
    streams = [torch.cuda.Stream() for _ in range(W)]
    cap = 2 * 512 + 1
    peers = make_world(L, cap, 15.0)

    def on_all(fn):
        for r in range(W):
            with torch.cuda.stream(streams[r]):
                fn(r, vp(streams[r].cuda_stream))

    # (1) stand-alone exchange, rank-ordered sum
    g = torch.Generator().manual_seed(8)
    for it, n in enumerate((1, 3, 129, 1025, 2, 513, 77, 1024, 5, 64, 257, cap, 9)):
        host = [torch.randn(n, dtype=torch.float64, generator=g) * 10.0 ** float(torch.randint(-6, 7, (1,), generator=g)) for _ in range(W)]
        want = host[0].clone()
        for r in range(1, W):
            want = want + host[r]                       # rank order 0, 1, ..., 7: fp64 addition is not associative
        rev = host[W - 1].clone()
        for r in range(W - 2, -1, -1):
            rev = rev + host[r]
        bufs = [h.to(dev) for h in host]
        torch.cuda.synchronize()
        on_all(lambda r, st: _native.check(L.semseg_peer_allreduce_sum_f64(peers[r], P(bufs[r]), n, st), 'peer_allreduce'))
        torch.cuda.synchronize()
        for r in range(W):
            assert L.semseg_peer_status(peers[r]) == 0, ('timeout', it, r)
            assert torch.equal(bufs[r].cpu(), want), ('exchange %d rank %d' % (it, r))
        if n >= 129:
            assert not torch.equal(rev, want), 'payload is not order sensitive'
    print('WORLD8_EXCHANGE_OK 13 exchanges, rank-ordered sums bit-identical on 8 ranks', flush=True)

    # (2) fused BN kernels at world 8 on unequal shards
    C = 64
    rows = [512, 384, 640, 512, 256, 768, 512, 448]
    gz = torch.Generator().manual_seed(9)
    z = [(torch.randn(p, C, generator=gz) * (1 + r) * 0.3 + 0.1 * r).to(dev) for r, p in enumerate(rows)]
    gamma = (torch.rand(C, generator=gz) + 0.5).to(dev)
    beta = torch.randn(C, generator=gz).to(dev)
    mom, eps = 0.1, 1e-5

    def bufs():
        return dict(stats=torch.empty(2 * C + 1, dtype=torch.float64, device=dev), zmm=torch.empty(2 * C, device=dev),
                    rm=torch.zeros(C, device=dev), rv=torch.ones(C, device=dev), nbt=torch.zeros((), dtype=torch.int64, device=dev),
                    coef=torch.empty(4, C, device=dev), bb=torch.empty((C + 15) // 16, dtype=torch.int32, device=dev),
                    absmax=torch.zeros(1, device=dev),
                    ws=torch.empty(L.semseg_bn_mm_workspace_bytes(max(rows), C), dtype=torch.uint8, device=dev))
    B = [bufs() for _ in range(W)]
    torch.cuda.synchronize()

    def fwd(r, st):
        b = B[r]
        _native.check(L.semseg_bn_fwd_stats_fused_peer(
            P(z[r]), rows[r], C, P(b['stats']), P(b['zmm']), P(gamma), P(beta), P(b['rm']), P(b['rv']), P(b['nbt']), mom, eps, 1,
            vp(0), P(b['coef'][0]), P(b['coef'][1]), P(b['coef'][2]), P(b['coef'][3]), P(b['bb']), P(b['ws']), b['ws'].numel(), st,
            peers[r], vp(0)), 'bn_fwd_stats_fused_peer')
    on_all(fwd)
    torch.cuda.synchronize()
    # the unfused path: per-shard statistics, rank-ordered host sum, finalize
    ref = bufs()
    tot = None
    for r in range(W):
        s = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
        zm = torch.empty(2 * C, device=dev)
        _native.check(L.semseg_bn_stats_mm(P(z[r]), rows[r], C, P(s), P(zm), P(ref['ws']), ref['ws'].numel(),
                                           vp(torch.cuda.current_stream().cuda_stream)), 'bn_stats_mm')
        torch.cuda.synchronize()
        tot = s.cpu() if tot is None else tot + s.cpu()
    for r in range(W):
        assert L.semseg_peer_status(peers[r]) == 0
        assert torch.equal(B[r]['stats'].cpu(), tot), ('fused peer stats != rank-ordered sum of the shards', r)
        assert torch.equal(B[r]['coef'].cpu(), B[0]['coef'].cpu()), ('coefficients differ between ranks', r)
        assert int(B[r]['nbt'].item()) == 1
    ref['stats'].copy_(tot)
    zmm0 = B[0]['zmm']                                  # min / max are rank-local: finalize rank 0's view
    _native.check(L.semseg_bn_finalize_mm(P(ref['stats']), P(zmm0), C, P(gamma), P(beta), P(ref['rm']), P(ref['rv']), P(ref['nbt']),
                                          mom, eps, 1, vp(0), P(ref['coef'][0]), P(ref['coef'][1]), P(ref['coef'][2]),
                                          P(ref['coef'][3]), P(ref['absmax']), vp(0), rows[0],
                                          vp(torch.cuda.current_stream().cuda_stream)), 'bn_finalize_mm')
    torch.cuda.synchronize()
    assert torch.equal(B[0]['coef'].cpu(), ref['coef'].cpu()), 'fused peer coefficients != unfused finalize of the summed statistics'
    assert torch.equal(B[3]['rm'].cpu(), ref['rm'].cpu()) and torch.equal(B[3]['rv'].cpu(), ref['rv'].cpu())
    print('WORLD8_FUSED_BN_OK fused BN statistics over 8 unequal shards == unfused path, identical on 8 ranks', flush=True)

    # (3) a rank that never shows up
    for p in peers:
        L.semseg_peer_destroy(p)
    peers = make_world(L, 64, 1.5)
    bufs3 = [torch.ones(16, dtype=torch.float64, device=dev) for _ in range(W)]
    for r in range(W - 1):                              # rank 7 stays away
        with torch.cuda.stream(streams[r]):
            _native.check(L.semseg_peer_allreduce_sum_f64(peers[r], P(bufs3[r]), 16, vp(streams[r].cuda_stream)), 'peer_allreduce')
    torch.cuda.synchronize()
    for r in range(W - 1):
        assert L.semseg_peer_status(peers[r]) != 0 and torch.isnan(bufs3[r]).all(), ('no timeout poisoning', r)
    assert L.semseg_peer_status(peers[W - 1]) == 0
    print('WORLD8_TIMEOUT_OK seven ranks poisoned their result and raised the status', flush=True)
    for p in peers:
        L.semseg_peer_destroy(p)


if __name__ == '__main__':
    main()
