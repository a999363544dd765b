"""Does an HBM-streaming pass (the optimiser: SGD + weight preparation, ~2 GB per step) hide behind the GEMMs of the rest of backward
when it is issued on a SECOND stream of the same hipGraph?  Round 5 measured GEMM beside GEMM (large weight gradients on a side
stream): 6 - 8 % slower.  This probe measures streaming beside GEMM: a chain A of layer3 / layer2 forward + data-gradient launches (what
backward still has to run once layer4 and the decoder are done) and a chain B of float4 copies (semseg_probe_copy), captured as
A then B on one stream and as A || B on two, replayed 20 times each.

    gpurun -- 'python tools/probes/overlap_probe.py'
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

vp = ctypes.c_void_p
LAYERS = [  # N  C     H   W   K    ks st pad dil  repeats
    (2, 256, 64, 64, 256, 3, 1, 2, 2, 5),
    (2, 1024, 64, 64, 256, 1, 1, 0, 1, 5),
    (2, 256, 64, 64, 1024, 1, 1, 0, 1, 6),
    (2, 128, 64, 64, 128, 3, 1, 1, 1, 3),
    (2, 128, 64, 64, 512, 1, 1, 0, 1, 4),
    (2, 64, 128, 128, 64, 3, 1, 1, 1, 3),
]


def main():
    ge.build()
    from mit_semseg import _native
    L = _native.lib()
    dev = torch.device('cuda:0')
    P = lambda t: vp(t.data_ptr())  # noqa: E731
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    calls = []
    s0 = vp(torch.cuda.current_stream().cuda_stream)

    def split(t, rows, ch):
        out = torch.empty(L.semseg_split_h2_bytes(rows, ch), dtype=torch.uint8, device=dev)
        _native.check(L.semseg_split_h2(P(t), ch, P(out), rows, ch, s0), 'split')
        return out
    keep = []
    for n, c, h, w, k, ks, st, pad, dil, rep in LAYERS:
        x = torch.randn(n, h, w, c, device=dev)
        wt = torch.randn(k, ks, ks, c, device=dev) * 0.02
        wtt = torch.randn(c, ks, ks, k, device=dev) * 0.02
        dy = torch.randn(n, h, w, k, device=dev) * 1e-3
        y = torch.empty(n, h, w, k, device=dev)
        dx = torch.empty(n, h, w, c, device=dev)
        xs, wss, wts, dys = split(x, n * h * w, c), split(wt, k * ks * ks, c), split(wtt, c * ks * ks, k), split(dy, n * h * w, k)
        keep += [x, wt, wtt, dy, y, dx, xs, wss, wts, dys]
        g = (n, h, w, c, k, ks, ks, st, pad, dil)
        for _ in range(rep):
            calls.append(lambda s, a=(P(xs), P(wss), vp(0), P(y), k) + g: L.semseg_conv2d_fwd_h2(*a, P(ws), ws.numel(), s))
            calls.append(lambda s, a=(P(dys), P(wts), P(dx), c) + g: L.semseg_conv2d_dgrad_h2(*a, P(ws), ws.numel(), s))
    nbytes = 256 << 20
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ncopy = 4                                    # 4 x (256 MiB read + 256 MiB written) = 2 GiB of traffic: the optimiser pass of configs[1]

    def chain_a(stream):
        s = vp(stream.cuda_stream)
        for f in calls:
            _native.check(f(s), 'conv')

    def chain_b(stream):
        s = vp(stream.cuda_stream)
        for _ in range(ncopy):
            _native.check(L.semseg_probe_copy(P(src), P(dst), nbytes, s), 'copy')

    main_s, side = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(main_s):              # warm: launch plans are looked up / timed outside any capture
        chain_a(main_s)
        chain_b(main_s)
    torch.cuda.synchronize()

    def capture(mode):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main_s):
            if mode == 'a':
                chain_a(main_s)
            elif mode == 'b':
                chain_b(main_s)
            elif mode == 'serial':
                chain_a(main_s)
                chain_b(main_s)
            else:
                side.wait_stream(main_s)
                chain_b(side)
                chain_a(main_s)
                main_s.wait_stream(side)
        return g

    def timed(g, reps=20):
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    res = {m: timed(capture(m)) for m in ('a', 'b', 'serial', 'parallel')}

    # the same two chains (1) launched eagerly on two streams, (2) as TWO graphs replayed on two streams
    def eager_two(reps=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            side.wait_stream(main_s)
            chain_b(side)
            chain_a(main_s)
            main_s.wait_stream(side)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def eager_one(reps=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            chain_a(main_s)
            chain_b(main_s)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    res['eager 1 stream'] = eager_one()
    res['eager 2 streams'] = eager_two()
    ga, gb = capture('a'), capture('b')

    def two_graphs(reps=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                gb.replay()
            with torch.cuda.stream(main_s):
                ga.replay()
            main_s.wait_stream(side)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    res['2 graphs, 2 streams'] = two_graphs()
    print('launches in chain A: %d   copies in chain B: %d x %d MiB' % (len(calls), ncopy, nbytes >> 20))
    for m, t in res.items():
        print('%-20s %.3f ms' % (m, t))
    print('A + B = %.3f ms; parallel saves %.3f ms of the %.3f ms of B' % (res['a'] + res['b'], res['serial'] - res['parallel'], res['b']))


if __name__ == '__main__':
    main()
