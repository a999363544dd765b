"""Micro-benchmark of the conv kernels through the C ABI on the layer shapes of
ade20k-resnet50dilated-ppm_deepsup (bs 2, 512x512): per layer and per pass (fwd / dgrad / wgrad), HIP-event
timing and achieved TFLOP/s for a list of forced tile / split configurations (env overrides read by the
library).  Used to set the launch heuristics and to produce the PMC profiles under profiles/.

    python tools/conv_bench.py [--layers conv_last,l4_conv2_d4,...] [--passes fwd,dgrad,wgrad] [--sweep]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd'))
import torch  # noqa: E402
from mit_semseg import _native  # noqa: E402

vp = ctypes.c_void_p
#            name            N  C     H   W   K     ks st pad dil   count/step
LAYERS = [
    ('conv_last',     2, 4096, 64, 64, 512,  3, 1, 1, 1, 1),
    ('deepsup',       2, 1024, 64, 64, 512,  3, 1, 1, 1, 1),
    ('l4_conv2_d4',   2, 512,  64, 64, 512,  3, 1, 4, 4, 2),
    ('l4_conv3',      2, 512,  64, 64, 2048, 1, 1, 0, 1, 3),
    ('l3_conv2_d2',   2, 256,  64, 64, 256,  3, 1, 2, 2, 5),
    ('l4_conv1',      2, 2048, 64, 64, 512,  1, 1, 0, 1, 2),
    ('l4_down',       2, 1024, 64, 64, 2048, 1, 1, 0, 1, 1),
    ('l3_conv3',      2, 256,  64, 64, 1024, 1, 1, 0, 1, 6),
    ('l3_conv1',      2, 1024, 64, 64, 256,  1, 1, 0, 1, 5),
    ('stem_conv3',    2, 64,  256, 256, 128, 3, 1, 1, 1, 1),
    ('stem_conv2',    2, 64,  256, 256, 64,  3, 1, 1, 1, 1),
    ('l1_conv2',      2, 64,  128, 128, 64,  3, 1, 1, 1, 3),
    ('l2_conv2',      2, 128, 64, 64, 128,   3, 1, 1, 1, 3),
    ('ppm_1x1_s6',    2, 2048, 6, 6, 512,    1, 1, 0, 1, 1),
    ('cls',           2, 512, 64, 64, 150,   1, 1, 0, 1, 2),
    # HRNetV2-W48 branches (hrnet.py: 4 x BasicBlock per branch and module)
    ('hr_48',         2, 48, 128, 128, 48,   3, 1, 1, 1, 64),
    ('hr_96',         2, 96,  64, 64, 96,    3, 1, 1, 1, 64),
    ('hr_192',        2, 192, 32, 32, 192,   3, 1, 1, 1, 56),
    ('hr_384',        2, 384, 16, 16, 384,   3, 1, 1, 1, 24),
    ('l1_conv1',      2, 256, 128, 128, 64,  1, 1, 0, 1, 2),
    ('l1_conv3',      2, 64, 128, 128, 256,  1, 1, 0, 1, 3),
    ('l2_conv3',      2, 128, 64, 64, 512,   1, 1, 0, 1, 4),
]


def run(layer, which, iters, L, ws, mode='f32'):
    name, n, c, h, w, k, ks, st, pad, dil, _ = layer
    dev = torch.device('cuda:0')
    oh = (h + 2 * pad - dil * (ks - 1) - 1) // st + 1
    ow = (w + 2 * pad - dil * (ks - 1) - 1) // st + 1
    torch.manual_seed(7)
    x = torch.randn(n, h, w, c, device=dev)
    wt = torch.randn(k, ks, ks, c, device=dev) * 0.02
    wtt = torch.randn(c, ks, ks, k, device=dev) * 0.02
    y = torch.empty(n, oh, ow, k, device=dev)
    dy = torch.randn(n, oh, ow, k, device=dev) * 1e-3
    dx = torch.empty(n, h, w, c, device=dev)
    dw = torch.empty(k, ks, ks, c, device=dev)
    s = vp(torch.cuda.current_stream().cuda_stream)
    P = lambda t: vp(t.data_ptr())  # noqa: E731

    if mode in ('s3', 'h2'):
        tag = mode
        f_bytes = getattr(L, 'semseg_split3_bytes' if mode == 's3' else 'semseg_split_h2_bytes')
        f_split = getattr(L, 'semseg_split3' if mode == 's3' else 'semseg_split_h2')
        f_fwd, f_dgrad, f_wgrad = (getattr(L, 'semseg_conv2d_%s_%s' % (w_, tag)) for w_ in ('fwd', 'dgrad', 'wgrad'))

        def split(t, rows, ch):
            out = torch.empty(f_bytes(rows, ch), dtype=torch.uint8, device=dev)
            _native.check(f_split(P(t), ch, P(out), rows, ch, s), 'split')
            return out
        xs, wss, wts, dys = split(x, n * h * w, c), split(wt, k * ks * ks, c), split(wtt, c * ks * ks, k), split(dy, n * oh * ow, k)

    def call():
        if mode in ('s3', 'h2'):
            if which == 'fwd':
                rc = f_fwd(P(xs), P(wss), vp(0), P(y), k, n, h, w, c, k, ks, ks, st, pad, dil, P(ws), ws.numel(), s)
            elif which == 'dgrad':
                rc = f_dgrad(P(dys), P(wts), P(dx), c, n, h, w, c, k, ks, ks, st, pad, dil, P(ws), ws.numel(), s)
            elif which == 'wgrad':
                rc = f_wgrad(P(xs), P(dys), P(dw), n, h, w, c, k, ks, ks, st, pad, dil, P(ws), ws.numel(), s)
            else:   # 'split': the per-conv split traffic of one training step (x once, dy once, w twice)
                rc = f_split(P(x), c, P(xs), n * h * w, c, s) or f_split(P(dy), k, P(dys), n * oh * ow, k, s) \
                    or f_split(P(wt), c, P(wss), k * ks * ks, c, s) or f_split(P(wtt), k, P(wts), c * ks * ks, k, s)
        elif which == 'fwd':
            rc = L.semseg_conv2d_fwd(P(x), c, P(wt), vp(0), P(y), k, n, h, w, c, k, ks, ks, st, pad, dil, P(ws), ws.numel(), s)
        elif which == 'dgrad':
            rc = L.semseg_conv2d_dgrad(P(dy), k, P(wtt), P(dx), c, n, h, w, c, k, ks, ks, st, pad, dil, P(ws), ws.numel(), s)
        else:
            rc = L.semseg_conv2d_wgrad(P(x), c, P(dy), k, P(dw), vp(0), n, h, w, c, k, ks, ks, st, pad, dil, P(ws), ws.numel(), s)
        _native.check(rc, which)
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        call()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    gflop = 2.0 * n * oh * ow * k * c * ks * ks * 1e-9
    out = {'fwd': y, 'dgrad': dx, 'wgrad': dw}.get(which)
    return ms, gflop / ms, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', default='')
    ap.add_argument('--passes', default='fwd,dgrad,wgrad')
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--sweep', action='store_true', help='also try forced tile/split configurations')
    ap.add_argument('--mode', default='f32', choices=['f32', 's3', 'h2'],
                    help='exact-fp32 MFMA kernels, the split-bf16 (s3) or the split-fp16 (h2) kernels')
    ap.add_argument('--tile', type=int, default=-1, help="force this tile id for the 'default' configuration (split families)")
    ap.add_argument('--split', type=int, default=0, help="force this split-K / split-M factor for the 'default' configuration")
    ap.add_argument('--verify', action='store_true',
                    help='compare every configuration with the exact-fp32 MFMA kernel on the same (seeded) inputs: max|err|/rms')
    ap.add_argument('--tiles', default='', help='--sweep: only these tile ids (comma separated)')
    ap.add_argument('--splits', default='1,2,4,8,16', help='--sweep: these split factors')
    args = ap.parse_args()
    L = _native.lib()
    ws = torch.empty(1 << 30, dtype=torch.uint8, device='cuda:0')
    sel = [l for l in LAYERS if not args.layers or l[0] in args.layers.split(',')]
    tot = {}
    for layer in sel:
        for which in args.passes.split(','):
            split_mode = args.mode in ('s3', 'h2')
            forced = {}
            if args.tile >= 0:
                forced[('SEMSEG_W3' if split_mode else 'SEMSEG_WGRAD') + '_TILE' if which == 'wgrad' else
                       ('SEMSEG_S3' if split_mode else 'SEMSEG_IGEMM') + '_TILE'] = str(args.tile)
            if args.split > 0:
                forced[('SEMSEG_W3' if split_mode else 'SEMSEG_WGRAD') + '_SPLIT' if which == 'wgrad' else
                       ('SEMSEG_S3' if split_mode else 'SEMSEG_IGEMM') + '_SPLITK'] = str(args.split)
            cfgs = [('default', forced)]
            splits = tuple(int(v) for v in args.splits.split(','))
            if args.sweep:
                pre = 'SEMSEG_W3' if split_mode else 'SEMSEG_WGRAD'
                pre2 = 'SEMSEG_S3' if split_mode else 'SEMSEG_IGEMM'
                if which == 'wgrad':
                    wtiles = tuple(range(15)) if args.mode == 'h2' else (0, 1)
                    if args.tiles:
                        wtiles = tuple(int(t) for t in args.tiles.split(','))
                    cfgs += [('t%d_s%d' % (t, sp), {pre + '_TILE': str(t), pre + '_SPLIT': str(sp)})
                             for t in wtiles for sp in ((32, 64, 128, 256, 512) if t == 10 else splits)]
                elif which != 'split':
                    tiles = tuple(range(27)) if args.mode == 'h2' else (0, 1, 2, 3)
                    if args.tiles:
                        tiles = tuple(int(t) for t in args.tiles.split(','))
                    cfgs += [('t%d_s%d' % (t, sp), {pre2 + '_TILE': str(t), pre2 + '_SPLITK': str(sp)})
                             for t in tiles for sp in splits]
            ref = None
            if args.verify and which != 'split':
                for kk in list(os.environ):
                    if kk.startswith('SEMSEG_') and kk.endswith(('_TILE', '_SPLITK', '_SPLIT')):
                        os.environ.pop(kk)
                ref = run(layer, which, 1, L, ws, 'f32')[2].double()
                ref_rms = ref.pow(2).mean().sqrt().item() + 1e-30
            res = []
            worst_err = 0.0
            for cname, env in cfgs:
                for k in ('SEMSEG_IGEMM_TILE', 'SEMSEG_IGEMM_SPLITK', 'SEMSEG_WGRAD_TILE', 'SEMSEG_WGRAD_SPLIT',
                          'SEMSEG_S3_TILE', 'SEMSEG_S3_SPLITK', 'SEMSEG_W3_TILE', 'SEMSEG_W3_SPLIT'):
                    os.environ.pop(k, None)
                os.environ.update(env)
                try:
                    ms, tf, out = run(layer, which, args.iters, L, ws, args.mode)
                    if ref is not None:
                        err = (out.double() - ref).abs().max().item() / ref_rms
                        if not err < 1e-4:           # NaN-safe
                            print('  VERIFY FAIL %s %s %s: max|err|/rms = %.3e' % (layer[0], which, cname, err), flush=True)
                            tf = -tf                  # a wrong kernel never wins the sweep
                        worst_err = max(worst_err, err) if err == err else float('nan')
                    res.append((cname, ms, tf))
                except RuntimeError as e:
                    res.append((cname, float('nan'), 0.0))
            d = res[0]
            best = max(res, key=lambda r: r[2])
            tot.setdefault(which, [0.0, 0.0])
            tot[which][0] += d[1] * layer[-1]
            tot[which][1] += best[1] * layer[-1]
            line = '%-12s %-5s default %8.3f ms %6.1f TF' % (layer[0], which, d[1], d[2])
            if args.sweep:
                line += ' | best %-8s %8.3f ms %6.1f TF | ' % best + ' '.join('%s:%.0f' % (r[0], r[2]) for r in res[1:])
            if ref is not None:
                line += ' | verify max|err|/rms %.1e' % worst_err
            print(line, flush=True)
    for k, v in tot.items():
        print('sum over listed layers x count: %-5s default %.2f ms  best-of-sweep %.2f ms' % (k, v[0], v[1]))


if __name__ == '__main__':
    main()
