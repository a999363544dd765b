"""Control experiment for the gradient / post-step acceptance bands of tests/util.py (VERDICT r2 "What's weak" 1b): the SAME
anchor-ratio measurement -- deviation of the native result from the float64 anchor of the unmodified reference, in units of the
reference's own fp32 reproducibility band -- once on the default h2 convolution path (2-way fp16 split, 2^-22 per product) and
once on the exact-fp32 MFMA path (SEMSEG_CONV=f32).  If the split products were what pushes tensors towards the acceptance
limit, the f32 column would sit visibly lower.  Also prints, for the tensors with NO ReLU gate between them and the loss (the
classifier convs), the plain elementwise error relative to the tensor's scale.

    python tools/probes/anchor_control.py            # runs itself under both modes, prints / writes one table per mode
    python tools/probes/anchor_control.py --cases r50_upernet_128_train --modes h2 f32 s3 h2:SEMSEG_WINOGRAD=0 h2:SEMSEG_TUNE=0
        # which approximation carries an outlier: any number of modes `conv[:ENV=V[,ENV=V]]`, one column per mode
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

GRAD_CASES = ['r18d_ppmds_64_train', 'r50d_ppmds_64_train', 'r50_upernet_128_train', 'hrnetv2_c1_64_train', 'hrnetv2_c1_128_train',
              'mnv2d_c1ds_64_train', 'mnv2d_c1ds_192_train', 'r18d_ppmds_64_trainedlike_train']


def summarize(ratios):
    rs = sorted(ratios)
    return rs[len(rs) // 2][0], rs[min(len(rs) - 1, int(len(rs) * 0.95))][0], rs[-1][0], rs[-1][1], len(rs)


def worker():
    import torch
    from tests import util
    from tests.test_gpu_models import build_native, _native_grads
    from oracle import semseg_oracle as O
    from mit_semseg import ops, tuner
    dev = torch.device('cuda:0')
    rows = []
    cases = [c for c in os.environ.get('ANCHOR_CONTROL_CASES', '').split(',') if c] or GRAD_CASES
    for name in cases:
        if name in util.HEURISTIC_PLAN_GOLDEN or name.startswith('mnv2d'):
            tuner.ENABLED = False
        g = util.load_golden(name)
        sm = _native_grads(g, dev)
        items, heads = [], []
        for mod, want, side in ((sm.encoder, g['anchor_grads_enc'], 'enc.'), (sm.decoder, g['anchor_grads_dec'], 'dec.')):
            for k, p in mod.named_parameters():
                items.append((side + k, p.grad, want[k]))
                if side == 'dec.' and util.is_head_tensor(k, p):
                    heads.append((util.scale_error(p.grad, want[k]), side + k))
        med, p95, mx, worst, n = summarize(util.anchor_ratios(items))
        raw = max(util.anchor_ratio(t, rec, k) for k, t, rec in items)          # without the case-wide lower limit of the band
        worst_scale = max(util.scale_error(t, rec) for k, t, rec in items)
        try:
            direction = util.check_directions(items, name, min_cos=-2.0, median_cos=-2.0, median_dot=float('inf'))
        except AssertionError as e:
            direction = 'FAILED: %s' % (e,)
        print('DIRECTION %s %s' % (ops.CONV_MODE, direction), flush=True)
        rows.append(dict(case=name, what='gradients', n=n, median=med, p95=p95, max=mx, worst=worst, raw_max=raw,
                         rel_band=util.case_rel_band([rec for _, _, rec in items]), worst_scale_err=worst_scale,
                         head_rel_err_max=max(h[0] for h in heads) if heads else None))
        # post-step state of the same case through TrainStep
        from mit_semseg.engine import TrainStep
        m = g['meta']
        sm, _, _ = build_native(g, dev)
        img, lab = O.synth_batch(m['n'], m['h'], m['w'], m['seg_rate'], seed=304 + m['seed'])
        ts = TrainStep(sm, lr_encoder=m['lr'], lr_decoder=m['lr'], max_iters=10 ** 9)
        ts.step({'img_data': img.to(dev), 'seg_label': lab.to(dev)})
        torch.cuda.synchronize()
        items = []
        for mod, want, side in ((sm.encoder, g['anchor_after_enc'], 'enc.'), (sm.decoder, g['anchor_after_dec'], 'dec.')):
            sd = mod.state_dict()
            for k in want:
                if k.rsplit('.', 1)[-1] in ('_tmp_running_mean', '_tmp_running_var', '_running_iter'):
                    continue
                items.append((side + k, sd[k], want[k]))
        med, p95, mx, worst, n = summarize(util.anchor_ratios(items, util.post_step_bands(g, m['lr'])))
        raw = max(util.anchor_ratio(t, rec, k) for k, t, rec in items)
        rows.append(dict(case=name, what='after-step', n=n, median=med, p95=p95, max=mx, worst=worst, raw_max=raw,
                         rel_band=util.case_rel_band([rec for _, _, rec in items]),
                         worst_scale_err=max(util.scale_error(t, rec) for k, t, rec in items), head_rel_err_max=None))
        tuner.ENABLED = True
    print('ANCHOR_TABLE ' + json.dumps({'mode': ops.CONV_MODE, 'rows': rows}), flush=True)


def main():
    if os.environ.get('ANCHOR_CONTROL_WORKER') == '1':
        return worker()
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', nargs='*', default=[])
    ap.add_argument('--modes', nargs='*', default=['h2', 'f32'])
    ap.add_argument('--tag', default='anchor_control_h2_vs_f32')
    args = ap.parse_args()
    out_dir = os.path.join(ROOT, 'gpurun_out', 'anchor_control')
    os.makedirs(out_dir, exist_ok=True)
    tables = {}
    directions = []            # tests/util.check_directions per (mode, case): the well-conditioned twin of the band statistics
    for spec in args.modes:
        conv, _, envs = spec.partition(':')
        env = dict(os.environ, SEMSEG_CONV=conv, ANCHOR_CONTROL_WORKER='1', ANCHOR_CONTROL_CASES=','.join(args.cases))
        for kv in (e for e in envs.split(',') if e):
            k, _, v = kv.partition('=')
            env[k] = v
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=1500)
        directions.extend(ln[len('DIRECTION '):] for ln in r.stdout.splitlines() if ln.startswith('DIRECTION '))
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('ANCHOR_TABLE ')]
        if not line:
            print('mode %s failed:\n%s\n%s' % (spec, r.stdout[-1500:], r.stderr[-3000:]))
            continue
        tables[spec] = json.loads(line[-1][len('ANCHOR_TABLE '):])['rows']
    lines = ['deviation from the float64 anchor of the unmodified reference, in units of the reference\'s own fp32 band (no tensor held',
             'to a tighter relative band than the median tensor of its case, tests/util.anchor_ratio; `raw` = without that lower limit;',
             '`scale` = largest |err| / max|ref| over the tensors of the case); one block per mode: median p95 max (raw max) scale, worst tensor']
    fmt = lambda r: '%5.2f %5.2f %6.2f (%8.2f) %.1e' % (r['median'], r['p95'], r['max'], r['raw_max'], r['worst_scale_err'])   # noqa: E731
    first = next(iter(tables.values()), [])
    for i, row in enumerate(first):
        lines.append('%-31s %-10s n=%d relband %.1e' % (row['case'], row['what'], row['n'], row['rel_band']))
        for spec, rows in tables.items():
            r = rows[i]
            lines.append('    %-36s %s   %s%s' % (spec, fmt(r), r['worst'],
                                                 ('   classifier |err|/scale %.1e' % r['head_rel_err_max']) if r['head_rel_err_max'] is not None else ''))
    if directions:
        lines.append('')
        lines.append('direction of every gradient tensor against the float64 anchor (tests/util.check_directions), `mode case: ...`')
        lines.extend(directions)
    text = '\n'.join(lines)
    print(text)
    with open(os.path.join(out_dir, args.tag + '.txt'), 'w') as fh:
        fh.write(text + '\n')
    with open(os.path.join(out_dir, args.tag + '.json'), 'w') as fh:
        json.dump(tables, fh)


if __name__ == '__main__':
    main()
