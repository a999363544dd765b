"""Latency of one peer exchange (csrc/peer.hip) on ONE GPU: two contexts of this process (attach_local), one stream each, a
hipGraph of `reps` back-to-back exchanges per rank, replayed concurrently.  What it measures: kernel launch inside a graph +
push + poll of uncached device memory -- the protocol's floor; across GPUs the xGMI store latency comes on top."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd'))
import torch  # noqa: E402


def main():
    from mit_semseg import _native
    L = _native.lib()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    world, reps = 2, 200
    cap = 2 * 4096 + 1
    peers = [ctypes.c_void_p() for _ in range(world)]
    for r in range(world):
        assert L.semseg_peer_create(r, world, cap, 10.0, ctypes.byref(peers[r])) == 0
    for r in range(world):
        for o in range(world):
            if o != r:
                assert L.semseg_peer_attach_local(peers[r], o, peers[o]) == 0
    # different priorities = different hardware queues (rank 0's kernel waits for rank 1's)
    streams = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]
    for n in (129, 513, 1025, 4097, 8193):
        bufs = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(world)]
        graphs = []
        torch.cuda.synchronize()
        for r in range(world):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[r]):
                st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                for _ in range(reps):
                    assert L.semseg_peer_allreduce_sum_f64(peers[r], ctypes.c_void_p(bufs[r].data_ptr()), n, st) == 0
            graphs.append(g)
        best = 1e9
        for it in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    graphs[r].replay()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        assert all(L.semseg_peer_status(p) == 0 for p in peers)
        # the same kernel count without a partner to wait for: a world-1 context (push / poll loops are empty)
        print('payload %5d doubles: %.2f us per exchange (2 ranks on one GPU, %d exchanges per graph)' % (n, best / reps * 1e6, reps), flush=True)
    for p in peers:
        L.semseg_peer_destroy(p)


if __name__ == '__main__':
    main()
