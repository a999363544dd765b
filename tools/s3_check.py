"""Correctness sweep of the operand-split convolution entry points (h2: 2 x fp16, s3: 3 x bf16) through the C ABI
against a float64 CPU convolution, next to the exact-fp32 MFMA kernels and torch's CPU fp32 convolution on the same
inputs.  Two error units per result: max|err| / rms(ref) (printed), and the COMPONENTWISE backward-stable measure
max_i |err_i| / (sum |a||b|)_i -- the quantity the fp32 error analysis of a dot product bounds (gamma_n ~ n 2^-24),
insensitive to a few outlier elements dominating rms(ref).  Pass criterion: the componentwise error of h2 / s3 is within
4x of the worse of the two fp32 implementations (or below 2^-20).
Also runs every case with operands rescaled by 2^-20 / 2^+12 (gradient-like and large magnitudes: the h2 scale).

    python tools/s3_check.py            # on the GPU box
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from mit_semseg import _native  # noqa: E402

vp = ctypes.c_void_p
P = lambda t: vp(t.data_ptr()) if t is not None else vp(0)  # noqa: E731

#        N  C    H   W   K    ks st pad dil bias
CASES = [
    (2, 64, 16, 16, 64, 3, 1, 1, 1, False),
    (2, 256, 16, 16, 256, 3, 1, 2, 2, False),
    (1, 512, 24, 24, 512, 3, 1, 4, 4, False),
    (2, 128, 17, 19, 96, 3, 2, 1, 1, False),      # stride 2, odd size, K not multiple of 32
    (2, 3, 32, 32, 64, 3, 2, 1, 1, False),        # stem
    (2, 512, 16, 16, 150, 1, 1, 0, 1, True),      # classifier (+bias)
    (2, 2048, 6, 6, 512, 1, 1, 0, 1, False),      # PPM branch, tiny M
    (2, 2048, 1, 1, 512, 1, 1, 0, 1, False),
    (2, 48, 20, 20, 48, 3, 1, 1, 1, False),       # HRNet widths
    (2, 720, 12, 12, 180, 3, 1, 1, 1, False),
    (2, 1024, 32, 32, 512, 3, 1, 1, 1, False),    # long K (9216) -> split-K
]


def run_case(L, case, dev, xscale=1.0, dyscale=1.0):
    n, c, h, w, k, ks, st, pad, dil, has_bias = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(n, c, h, w, generator=g).relu() * 1.5 * xscale
    wt = torch.randn(k, c, ks, ks, generator=g) * (2.0 / (c * ks * ks)) ** 0.5
    b = torch.randn(k, generator=g) * xscale if has_bias else None
    oh = (h + 2 * pad - dil * (ks - 1) - 1) // st + 1
    ow = (w + 2 * pad - dil * (ks - 1) - 1) // st + 1
    dy = torch.randn(n, k, oh, ow, generator=g) * dyscale
    # a few outliers 200x the bulk, as ReLU'd activations / sparse gradients have (the h2 scale follows the max)
    x.view(-1)[::9973] *= 200.0
    dy.view(-1)[::7919] *= 200.0
    xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    bd = b.double().requires_grad_(True) if has_bias else None
    yd = F.conv2d(xd, wd, bd, st, pad, dil)
    yd.backward(dy.double())
    ref = dict(y=yd.detach(), dx=xd.grad, dw=wd.grad, db=bd.grad if has_bias else None)
    # the yardstick: torch's own CPU fp32 convolution (what the reference computes with) against float64
    x32, w32 = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    b32 = b.clone().requires_grad_(True) if has_bias else None
    y32 = F.conv2d(x32, w32, b32, st, pad, dil)
    y32.backward(dy)
    cpu32 = dict(y=y32.detach(), dx=x32.grad, dw=w32.grad, db=b32.grad if has_bias else None)
    # sum |a||b| of every output element: the same conv / gradients on the absolute values
    xa, wa = x.double().abs().requires_grad_(True), wt.double().abs().requires_grad_(True)
    ba = b.double().abs().requires_grad_(True) if has_bias else None
    ya = F.conv2d(xa, wa, ba, st, pad, dil)
    ya.backward(dy.double().abs())
    mag = dict(y=ya.detach(), dx=xa.grad, dw=wa.grad, db=ba.grad if has_bias else None)

    s = vp(torch.cuda.current_stream().cuda_stream)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev)            # NHWC
    wg = wt.permute(0, 2, 3, 1).contiguous().to(dev)           # KRSC
    dyg = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    bg = b.to(dev) if has_bias else None
    wtg = torch.empty(c, ks, ks, k, device=dev)
    _native.check(L.semseg_weight_krsc_to_crsk(P(wg), P(wtg), k, ks * ks, c, s), 'transpose')
    wsb = max(L.semseg_conv2d_s3_workspace_bytes(n, h, w, c, k, ks, ks, st, pad, dil),
              L.semseg_conv2d_workspace_bytes(n, h, w, c, k, ks, ks, st, pad, dil), 1 << 20)
    wsb = max(wsb, L.semseg_conv2d_h2_workspace_bytes(n, h, w, c, k, ks, ks, st, pad, dil))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

    def split(mode, t, rows, ch):
        fb = L.semseg_split3_bytes if mode == 's3' else L.semseg_split_h2_bytes
        fs = L.semseg_split3 if mode == 's3' else L.semseg_split_h2
        out = torch.empty(fb(rows, ch), dtype=torch.uint8, device=dev)
        _native.check(fs(P(t), ch, P(out), rows, ch, s), 'split')
        return out

    res = {}
    for key in ('y', 'dx', 'dw', 'db'):
        if ref[key] is not None:
            r = ref[key]
            res[('cpu', key)] = (cpu32[key].double() - r).abs().max().item() / (r.pow(2).mean().sqrt().item() + 1e-30)
            res[('cpu', key, 'cw')] = ((cpu32[key].double() - r).abs() / (mag[key] + 1e-300)).max().item()
    for mode in ('h2', 's3', 'f32'):
        y = torch.full((n, oh, ow, k), float('nan'), device=dev)
        dx = torch.full((n, h, w, c), float('nan'), device=dev)
        dw = torch.full((k, ks, ks, c), float('nan'), device=dev)
        db = torch.full((k,), float('nan'), device=dev) if has_bias else None
        geo = (n, h, w, c, k, ks, ks, st, pad, dil)
        if mode in ('s3', 'h2'):
            xs = split(mode, xg, n * h * w, c)
            wss = split(mode, wg, k * ks * ks, c)
            wts = split(mode, wtg, c * ks * ks, k)
            dys = split(mode, dyg, n * oh * ow, k)
            f = lambda w_: getattr(L, 'semseg_conv2d_%s_%s' % (w_, mode))  # noqa: E731
            _native.check(f('fwd')(P(xs), P(wss), P(bg), P(y), k, *geo, P(ws), ws.numel(), s), 'fwd_' + mode)
            _native.check(f('dgrad')(P(dys), P(wts), P(dx), c, *geo, P(ws), ws.numel(), s), 'dgrad_' + mode)
            _native.check(f('wgrad')(P(xs), P(dys), P(dw), *geo, P(ws), ws.numel(), s), 'wgrad_' + mode)
            if has_bias:
                _native.check(L.semseg_bias_grad(P(dyg), k, P(db), n * oh * ow, k, P(ws), ws.numel(), s), 'bias_grad')
        else:
            _native.check(L.semseg_conv2d_fwd(P(xg), c, P(wg), P(bg), P(y), k, *geo, P(ws), ws.numel(), s), 'fwd')
            _native.check(L.semseg_conv2d_dgrad(P(dyg), k, P(wtg), P(dx), c, *geo, P(ws), ws.numel(), s), 'dgrad')
            _native.check(L.semseg_conv2d_wgrad(P(xg), c, P(dyg), k, P(dw), P(db), *geo, P(ws), ws.numel(), s), 'wgrad')
        torch.cuda.synchronize()
        got = dict(y=y.permute(0, 3, 1, 2), dx=dx.permute(0, 3, 1, 2), dw=dw.permute(0, 3, 1, 2), db=db)
        for key in ('y', 'dx', 'dw', 'db'):
            if ref[key] is None:
                continue
            r = ref[key]
            e = (got[key].cpu().double() - r).abs().max().item() / (r.pow(2).mean().sqrt().item() + 1e-30)
            res[(mode, key)] = e
            res[(mode, key, 'cw')] = ((got[key].cpu().double() - r).abs() / (mag[key] + 1e-300)).max().item()
    return res


def main():
    L = _native.lib()
    dev = torch.device('cuda:0')
    bad = 0
    for xscale, dyscale in ((1.0, 1.0), (2.0 ** 12, 2.0 ** -20)):
        print('--- operand scales: x * %g, dy * %g' % (xscale, dyscale))
        for case in CASES:
            res = run_case(L, case, dev, xscale, dyscale)
            line = 'N%d C%-4d %2dx%-2d K%-4d k%d s%d p%d d%d b%d |' % case
            for key in ('y', 'dx', 'dw', 'db'):
                if ('s3', key) in res:
                    h2, a, b, cpu = res[('h2', key)], res[('s3', key)], res[('f32', key)], res[('cpu', key)]
                    cw = [res[(m, key, 'cw')] for m in ('h2', 's3', 'f32', 'cpu')]
                    lim = max(4 * max(cw[2], cw[3]), 2.0 ** -20)
                    ok = all((v == v) and v < lim for v in cw[:2]) and (h2 == h2) and (a == a)   # NaN-safe
                    bad += 0 if ok else 1
                    line += ' %s h2 %.1e s3 %.1e f32 %.1e cpu32 %.1e [cw %.1e %.1e %.1e %.1e]%s |' % (
                        (key, h2, a, b, cpu) + tuple(cw) + ('' if ok else ' <-- BAD',))
            print(line, flush=True)
    print('s3_check: %s' % ('OK' if bad == 0 else '%d FAILURES' % bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
