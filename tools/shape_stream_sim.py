"""Where the variable-size path (BASELINE configs[3]) lives over a real run: the batch-shape stream of the reference's multi-scale
rule (dataset.py:110-142) over the ADE20K size list, pushed through TrainStep's per-shape graph policies (recorded at first or second sight, LRU of `cap`
graphs, recording with or without a device synchronize) with the per-event costs a bench run measured (bench.py raw_stream leg) on
a two-clock model (host issues, device executes).  Pure host arithmetic.

    python tools/shape_stream_sim.py [--replay-ms 22.05 --eager-ms 47 --capture-ms 32]
"""
import argparse
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd'))


def stream(n, seed=304):
    from mit_semseg.dataset import batch_shape_stream
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'ade20k_train_sizes.npz'))
    g = batch_shape_stream(list(zip(d['width'].tolist(), d['height'].tolist())), d['count'], seed=seed)
    return [next(g) for _ in range(n)]


def simulate(shapes, warm, steps, cap, replay_ms, eager_ms, capture_ms, first_sight=True, sync_capture=False, host_replay_ms=0.3):
    """img/s (2 images per step) over steps [warm, warm + steps) and the events inside them.  Two clocks: the host issues steps (a replay
    costs it `host_replay_ms`, an eager pass `eager_ms`, a recording pass `capture_ms`), the device executes them in order, `replay_ms`
    each, never before they are issued.  `sync_capture`: the recording pass starts with a device synchronize (torch.cuda.graph's
    __enter__; rounds 2 - 5 until its last day) -- the device idles through every recording.  `first_sight`: a new shape is recorded
    the first time it is seen (round 5), else run eagerly once and recorded at its second sight."""
    seen, lru, ev = collections.Counter(), collections.OrderedDict(), collections.Counter()
    h = d = 0.0
    t_begin = 0.0
    for i, s in enumerate(shapes[:warm + steps]):
        if i == warm:
            h = d = t_begin = max(h, d)                 # the timed window opens on an idle device (bench.py synchronises)
        if s in lru:
            lru.move_to_end(s)
            h += host_replay_ms
            e = 'replayed'
        else:
            seen[s] += 1
            if first_sight or seen[s] >= 2:
                lru[s] = 1
                if len(lru) > cap:
                    lru.popitem(last=False)
                    ev['evicted'] += i >= warm
                if sync_capture:
                    h = max(h, d)
                h += capture_ms + host_replay_ms
                e = 'captured'
            else:
                h += eager_ms
                e = 'eager'
        d = max(d, h) + replay_ms
        if i >= warm:
            ev[e] += 1
    return 2e3 * steps / (max(h, d) - t_begin), dict(ev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--replay-ms', type=float, default=22.05, help='device time of a step (bench.py configs[3] steady state: 90.7 img/s)')
    ap.add_argument('--eager-ms', type=float, default=47.0, help='host time of an eager pass (raw_stream leg, gpurun r6W)')
    ap.add_argument('--capture-ms', type=float, default=32.0, help='host time of a recording pass without its synchronize (r6X)')
    a = ap.parse_args()
    shapes = stream(100000)
    c = collections.Counter(shapes[:20000])
    print('distinct shapes in 20000 / 100000 iterations: %d / %d' % (len(c), len(set(shapes))))
    tot, cover = 0, {}
    for i, (_, n) in enumerate(c.most_common()):
        tot += n
        if i + 1 in (5, 16, 64, 256):
            cover[i + 1] = tot / 20000.0
    print('coverage of the stream by its most frequent shapes: ' + ', '.join('%d: %.0f %%' % (k, 100 * v) for k, v in cover.items()))
    steady = 2e3 / a.replay_ms
    print('steady state %.1f img/s;  img/s (fraction of steady state) over a window [warm, warm + steps); host: replay 0.3 ms, eager pass %.0f ms, '
          'recording pass %.0f ms' % (steady, a.eager_ms, a.capture_ms))
    policies = (('LRU 16, 2nd sight, sync', dict(cap=16, first_sight=False, sync_capture=True)),
                ('LRU 512, 2nd sight, sync', dict(cap=512, first_sight=False, sync_capture=True)),
                ('LRU 512, 1st sight, sync', dict(cap=512, first_sight=True, sync_capture=True)),
                ('LRU 512, 1st sight, no sync', dict(cap=512, first_sight=True, sync_capture=False)))
    print('%8s %8s | ' % ('warm', 'steps') + ' | '.join('%27s' % n for n, _ in policies))
    for warm, steps in ((100, 300), (1000, 1000), (5000, 5000), (20000, 5000), (95000, 5000)):
        row = []
        for _, kw in policies:
            v, ev = simulate(shapes, warm, steps, replay_ms=a.replay_ms, eager_ms=a.eager_ms, capture_ms=a.capture_ms, **kw)
            row.append('%6.1f (%.2f) %4d recorded' % (v, v / steady, ev.get('captured', 0)))
        print('%8d %8d | ' % (warm, steps) + ' | '.join('%27s' % r for r in row))


if __name__ == '__main__':
    main()
