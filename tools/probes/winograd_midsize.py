"""Where would Winograd F(2x2,3x3) pay below 1024 input channels?  (north_star's named kernels, the 3x3 dilated
convs of layer3 / layer4.)  Times, per layer geometry, every piece separately through the C ABI with HIP events:
  direct   : semseg_conv2d_fwd_stats_h2 under the tuner's plan (+ the statistics sweep when that plan splits K)
  wino in  : semseg_winograd_input_h2 (fp32 x -> V planes; a fused producer would pay only the extra 8 B / element of V)
  wino gemm: semseg_winograd_gemm_h2 for every tile form (plan pass 3)
  wino out : semseg_winograd_output (M -> z)
so that the best case of a fully fused Winograd path (gemm + the extra traffic of the transforms) can be put next to the direct
convolution it would replace.    python tools/probes/winograd_midsize.py
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

vp = ctypes.c_void_p
LAYERS = [('layer4 512->512 d4', 2, 64, 64, 512, 512, 4), ('layer4 512->512 d2', 2, 64, 64, 512, 512, 2),
          ('layer3 256->256 d2', 2, 64, 64, 256, 256, 2), ('layer3 256->256 d1', 2, 64, 64, 256, 256, 1),
          ('deepsup 1024->512 d1', 2, 64, 64, 1024, 512, 1), ('upernet fpn_out 512->512 @128 d1', 2, 128, 128, 512, 512, 1)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def main():
    import __graft_entry__ as ge
    ge.build()
    from mit_semseg import _native, ops, tuner
    L = _native.lib()
    dev = torch.device('cuda:0')
    P = lambda t: vp(t.data_ptr())                    # noqa: E731
    st = lambda: vp(torch.cuda.current_stream().cuda_stream)   # noqa: E731
    print('%-34s %9s | %8s %8s %8s | %s' % ('layer', 'direct us', 'wino in', 'wino out', 'sweep', 'wino gemm us per tile id'))
    for name, n, h, w, c, k, dil in LAYERS:
        geom = (n, h, w, c, k, 3, 3, 1, dil, dil)
        M = n * h * w
        g = torch.Generator().manual_seed(1)
        x = torch.randn(n, h, w, c, generator=g).to(dev)
        wt = (torch.randn(k, 3, 3, c, generator=g) * 0.02).to(dev)
        xp = ops.SCHEMES['h2'].split(x, M, c, c)
        wp = ops.SCHEMES['h2'].split(wt, k * 9, c, c)
        z = torch.empty(M, k, device=dev)
        bound = torch.full((1,), float(x.abs().max()), device=dev)
        stats_bytes = L.semseg_conv2d_fwd_stats_bytes(k)
        parts = ctypes.c_int(0)

        def direct():
            cb = (L.semseg_conv2d_h2_workspace_bytes(*geom) + 255) & ~255
            ws = ops.workspace(cb + stats_bytes, dev)
            _native.check(L.semseg_conv2d_fwd_stats_h2(P(xp), P(wp), P(z), k, *geom, vp(ws.data_ptr()), cb, vp(ws.data_ptr() + cb),
                                                       stats_bytes, P(bound), ctypes.byref(parts), st()), 'conv')
            if parts.value == 0:
                _native.check(L.semseg_bn_stats_mm_partial(P(z), M, k, vp(ws.data_ptr() + cb), stats_bytes, st()), 'sweep')
        tuner.ensure('h2', 0, geom, direct)
        t_direct = timeit(direct)
        plan = tuner.tuned_plans().get(('h2', 0) + geom)
        sweep_ws = torch.empty(stats_bytes, dtype=torch.uint8, device=dev)
        t_sweep = timeit(lambda: _native.check(L.semseg_bn_stats_mm_partial(P(z), M, k, P(sweep_ws), stats_bytes, st()), 'sweep'))
        tiles = L.semseg_winograd_tiles(n, h, w, dil)
        v = torch.empty(L.semseg_split_h2_bytes(16 * tiles, c), dtype=torch.uint8, device=dev)
        m = torch.empty(16 * tiles, k, device=dev)
        # U planes: any planes of the right shape do for timing
        u = ops.SCHEMES['h2'].split((torch.randn(16 * k, c, generator=g) * 0.02).to(dev), 16 * k, c, c)
        bp = (vp * 1)(bound.data_ptr())
        t_in = timeit(lambda: _native.check(L.semseg_winograd_input_h2(P(x), c, bp, 1, P(v), n, h, w, c, dil, st()), 'in'))
        t_out = timeit(lambda: _native.check(L.semseg_winograd_output(P(m), P(z), k, n, h, w, k, dil, st()), 'out'))
        gemms = {}
        for tile in (6, 7, 8, 9, 10, 14):
            if L.semseg_conv2d_h2_set_plan(3, tiles, 1, 1, c, k, 3, 3, 1, 1, 1, tile, 1) != 0:
                continue
            try:
                gemms[tile] = timeit(lambda: _native.check(L.semseg_winograd_gemm_h2(P(v), P(u), P(m), tiles, c, k, st()), 'gemm'))
            except RuntimeError:
                pass
        L.semseg_conv2d_h2_set_plan(3, tiles, 1, 1, c, k, 3, 3, 1, 1, 1, -1, 0)
        best = min(gemms.values())
        gf = 2.0 * M * c * k * 9 * 1e-9
        print('%-34s %9.1f | %8.1f %8.1f %8.1f | %s   plan %s parts %d | direct %.0f TF, wino gemm alone %.0f TF-alg, '
              'fully fused best case (gemm + extra 8 B/elem of V + out - sweep) %.1f us' % (
                  name, t_direct, t_in, t_out, t_sweep, ' '.join('%d:%.1f' % kv for kv in sorted(gemms.items())),
                  list(plan[:2]) if plan else None, parts.value, gf / t_direct * 1e3, gf / best * 1e3,
                  best + 0.5 * t_in + t_out - t_sweep), flush=True)


if __name__ == '__main__':
    main()
