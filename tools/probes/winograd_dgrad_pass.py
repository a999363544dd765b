"""The dominant pass of the configs[1] step since round 3 -- the Winograd-domain data gradient of decoder.conv_last.0 (3x3, 4096 <- 512
@ 64 x 64, N = 2) -- launched `--iters` times through the C ABI, for rocprofv3 (kernel trace / PMC passes, tools/gpu_pmc_wino.sh):
input transform from the planes of dy, the batched GEMM on its tuned tile, output transform."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--tile', type=int, default=14)
    ap.add_argument('--geom', default='2,64,64,4096,512,1', help='n,h,w,c,k,dil of the 3x3 stride-1 conv (default: conv_last)')
    ap.add_argument('--mode', default='dgrad', choices=('dgrad', 'fwd'))
    ap.add_argument('--form', type=int, default=0,
                    help='dgrad: 0 = batched GEMM + output transform, 1 / 2 / 3 = the fused kernel on its 3- / 4- / 5-slot ring')
    ap.add_argument('--probes', action='store_true', help='with --time: also the DMA-only / multiply-only probes of the fused kernel')
    ap.add_argument('--time', action='store_true', help='dgrad: HIP-event time of every form on this geometry (ms per pass)')
    args = ap.parse_args()
    from mit_semseg import ops, _native
    L = _native.lib()
    dev = torch.device('cuda:0')
    n, h, w, c, k, dil = (int(v) for v in args.geom.split(','))
    geom = (n, h, w, c, k, 3, 3, 1, dil, dil)
    if args.mode == 'fwd':
        # the forward of the same layer family (e.g. layer4's dilated 512 -> 512 convs: --geom 2,64,64,512,512,4): input transform
        # of x, batched GEMM on the pinned tile, output transform
        from mit_semseg import tuner
        tuner.ENABLED = False
        wparam = torch.nn.Parameter((torch.randn(k, 3, 3, c, device=dev) * 0.01).permute(0, 3, 1, 2))
        ops.prepare_conv_weights([wparam])
        x = torch.randn(n, c, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        bound = x.abs().max().reshape(1)
        z = ops.empty_nhwc(n, k, h, w, dev)
        tiles = L.semseg_winograd_tiles(n, h, w, dil)
        _native.check(L.semseg_conv2d_h2_set_plan(3, tiles, 1, 1, c, k, 3, 3, 1, 1, 1, args.tile, 1), 'set_plan')
        for _ in range(args.iters):
            ops._winograd_fwd(L, x, (bound,), ops.weight_wino(wparam), z, geom)
        torch.cuda.synchronize()
        print('done: %d forward passes, tiles %d, gemm tile %d' % (args.iters, tiles, args.tile))
        return
    wparam = torch.nn.Parameter((torch.randn(k, 3, 3, c, device=dev) * 0.01).permute(0, 3, 1, 2))
    ops.prepare_conv_weights([wparam])
    ut = ops.weight_wino_t(wparam)
    dy = torch.randn(n, h, w, k, device=dev) * 1e-3
    dyp = ops.SCHEMES['h2'].split(dy, n * h * w, k, k)
    tiles = L.semseg_winograd_tiles(n, h, w, dil)
    from mit_semseg import tuner
    tuner.ENABLED = False
    _native.check(L.semseg_conv2d_h2_set_plan(3, tiles, 1, 1, k, c, 3, 3, 1, 1, 1, args.tile, 1), 'set_plan')
    if args.time:
        gflop = 2.0 * n * h * w * c * k * 9 * 1e-9
        forms = list(range(1 + ops.WINOGRAD_FUSED_FORMS)) + ([101, 102, 103, 104] if args.probes else [])
        for form in forms:
            if form > 100:      # 101 = the fused kernel's DMA stream alone, 102 = its fragment reads + MFMAs alone (garbage results)
                tiles = L.semseg_winograd_tiles(n, h, w, dil)
                v = torch.empty(L.semseg_split_h2_bytes(16 * tiles, k), dtype=torch.uint8, device=dev)
                _native.check(L.semseg_winograd_input_planes_h2(ops._p(dyp), ops._p(v), n, h, w, k, dil, ops._st()), 'input')
                dx = ops.empty_nhwc(n, c, h, w, dev)
                run = lambda: _native.check(L.semseg_winograd_gemm_output_h2(ops._p(v), ops._p(ut), ops._p(dx), c, n, h, w, k, c, dil,   # noqa: E731
                                                                             form - 1, ops._st()), 'probe')
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                what = {101: 'DMA stream only', 102: 'fragment reads + MFMAs only',
                        103: 'the piece stream by plain 16-byte loads into registers, no LDS',
                        104: 'A pieces by LDS-DMA, B pieces by plain loads into registers'}[form]
                print('geom %s probe %d (%s): %.4f ms per launch' % (args.geom, form, what, e0.elapsed_time(e1) / args.iters))
                continue
            for _ in range(3):
                ops._winograd_dgrad(L, dyp, ut, geom, form=form)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                ops._winograd_dgrad(L, dyp, ut, geom, form=form)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            print('geom %s form %d: %.4f ms per pass (input transform + GEMM + output), %.1f algorithmic TFLOP/s'
                  % (args.geom, form, ms, gflop / ms))
        return
    for _ in range(args.iters):
        ops._winograd_dgrad(L, dyp, ut, geom, form=args.form)
    torch.cuda.synchronize()
    print('done: %d passes, tiles %d, gemm tile %d, form %d' % (args.iters, tiles, args.tile, args.form))


if __name__ == '__main__':
    main()
