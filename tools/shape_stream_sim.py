"""Where the variable-size path (BASELINE configs[3]) lives over a real run: the batch-shape stream of the reference's multi-scale
rule (dataset.py:110-142) over the ADE20K size list, pushed through TrainStep's per-shape graph policy (first sight eager, second
sight capture, LRU of `cap` graphs) with the per-event costs a bench run measured (bench.py raw_stream leg).  Pure host arithmetic.

    python tools/shape_stream_sim.py [--replay-ms 24.6 --eager-ms 60 --capture-ms 150]
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


def simulate(shapes, warm, steps, cap, replay_ms, eager_ms, capture_ms):
    """img/s (2 images per step) over steps [warm, warm + steps) and the events inside them"""
    seen, lru, t, ev = collections.Counter(), collections.OrderedDict(), 0.0, collections.Counter()
    for i, s in enumerate(shapes[:warm + steps]):
        if s in lru:
            lru.move_to_end(s)
            c, e = replay_ms, 'replayed'
        else:
            seen[s] += 1
            if seen[s] >= 2:
                lru[s] = 1
                if len(lru) > cap:
                    lru.popitem(last=False)
                    ev['evicted'] += i >= warm
                c, e = capture_ms + replay_ms, 'captured'
            else:
                c, e = eager_ms, 'eager'
        if i >= warm:
            t += c
            ev[e] += 1
    return 2e3 * steps / t, dict(ev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--replay-ms', type=float, default=24.6)
    ap.add_argument('--eager-ms', type=float, default=60.0)
    ap.add_argument('--capture-ms', type=float, default=150.0)
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
    print('steady state %.1f img/s;  img/s (fraction of steady state) over a window [warm, warm + steps):' % steady)
    print('%8s %8s | %22s | %22s' % ('warm', 'steps', 'LRU 16', 'LRU 512'))
    for warm, steps in ((100, 300), (1000, 1000), (5000, 5000), (20000, 5000), (95000, 5000)):
        row = []
        for cap in (16, 512):
            v, ev = simulate(shapes, warm, steps, cap, a.replay_ms, a.eager_ms, a.capture_ms)
            row.append('%6.1f (%.2f) %4d capt' % (v, v / steady, ev.get('captured', 0)))
        print('%8d %8d | %22s | %22s' % (warm, steps, row[0], row[1]))


if __name__ == '__main__':
    main()
