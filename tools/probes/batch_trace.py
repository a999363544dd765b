"""Which launches of an HRNetV2 training step pair up in the side-by-side scopes (csrc/batch.h), and which kernel forces a flush:
runs two eager steps of a golden case with SEMSEG_BATCH_DEBUG=1 in a child and summarises the trace.
    python tools/probes/batch_trace.py [golden case]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def worker(case):
    import torch
    from tests import util
    from tests.test_gpu_models import build_native
    from oracle import semseg_oracle as O
    from mit_semseg.engine import TrainStep
    g = util.load_golden(case)
    m = g['meta']
    dev = torch.device('cuda:0')
    sm, _, _ = build_native(g, dev)
    img, lab = O.synth_batch(m['n'], m['h'], m['w'], m['seg_rate'], seed=304 + m['seed'])
    ts = TrainStep(sm, max_iters=1000)
    for _ in range(2):
        ts.step({'img_data': img.to(dev), 'seg_label': lab.to(dev)})
    torch.cuda.synchronize()


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'hrnetv2_c1_128_train'
    if os.environ.get('BATCH_TRACE_WORKER') == '1':
        return worker(case)
    if case.startswith('bench:'):
        # the BASELINE configuration at full size: bench.py records the step ONCE into its hipGraph, so the trace is one step's scopes
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--config', case[6:], '--steps', '3', '--warmup', '3', '--no-cpu-baseline',
               '--no-other-configs', '--no-box', '--no-scaling-model', '--repeats', '0']
        r = subprocess.run(cmd, env=dict(os.environ, SEMSEG_BATCH_DEBUG='1'), capture_output=True, text=True, timeout=900)
    else:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), case],
                           env=dict(os.environ, BATCH_TRACE_WORKER='1', SEMSEG_BATCH_DEBUG='1'), capture_output=True, text=True, timeout=900)
    groups, forced = collections.Counter(), collections.Counter()
    for ln in r.stderr.splitlines():
        m = re.match(r'\[semseg_batch\] op \d+: (\d+) x (.*) \(branches', ln)
        if m:
            groups[(m.group(2)[:70] if case.startswith('bench:') else re.sub(r'<.*', '', m.group(2)), int(m.group(1)))] += 1
        m = re.match(r'\[semseg_batch\] branch \d+ op \d+: direct launch of (.*) forces a flush', ln)
        if m:
            forced[m.group(1)[:100]] += 1
    print('case %s (%s): rc %d' % (case, 'one recorded step' if case.startswith('bench:') else 'two eager steps', r.returncode))
    if r.returncode:
        print(r.stderr[-3000:])
    print('launches issued by the zip, by (kernel body, problems in the launch):')
    for (k, n), c in sorted(groups.items()):
        print('  %-72s x%d  %5d' % (k, n, c))
    print('direct launches inside a scope (each forces a flush):')
    for k, c in forced.most_common():
        print('  %5d  %s' % (c, k))


if __name__ == '__main__':
    main()
