"""Concurrency profile of a rocprofv3 --kernel-trace run of a replayed training step (rocpd .db or *_kernel_trace.csv): over the last
`steps` steps of the trace (a step = the dispatches between two launches of `marker`, default the SGD kernel) report per step the wall
span, the summed kernel time, the time with 0 / 1 / 2 / 3 / 4+ kernels in flight, and the kernel families ranked by EXCLUSIVE time
(time during which nothing else runs: what a shorter critical path would have to attack) and by total time.

    python tools/trace_concurrency.py gpurun_out/<tag>/prof_cfg4/bench_results.db [steps=8] [marker=sgd_kernel]
"""
import collections
import re
import sys

from trace_gaps import load


def family(name):
    n = name.replace('void ', '')
    m = re.match(r'semseg_batch::(one|many)_kernel<(\w+?)_body\b', n)
    if m:                                   # the generic kernels of csrc/batch.h: name the body they run ("+": many problems per launch)
        return m.group(2) + ('+' if m.group(1) == 'many' else '')
    n = re.sub(r'\(.*', '', n)
    n = re.sub(r'<.*', '', n)
    return n.split('::')[-1]


def main():
    rows = load(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    marker = sys.argv[3] if len(sys.argv) > 3 else 'sgd_kernel,sgd_fused_kernel'      # the optimiser kernel ends a step (either form)
    marks = [i for i, (s, e, n) in enumerate(rows) if any(m in n for m in marker.split(','))]
    # a step may launch the marker several times back to back: keep the LAST dispatch of every cluster
    ends = [i for k, i in enumerate(marks) if k + 1 == len(marks) or rows[marks[k + 1]][0] - rows[i][1] > 2_000_000]
    ends = ends[-(steps + 1):]
    if len(ends) < 2:
        print('fewer than two steps found')
        return
    seg = rows[ends[0] + 1:ends[-1] + 1]
    n_steps = len(ends) - 1
    ev = []
    for s, e, n in seg:
        ev.append((s, 1, n))
        ev.append((e, -1, n))
    ev.sort()
    level = 0
    active = collections.Counter()
    t_prev = ev[0][0]
    at = collections.Counter()
    excl = collections.Counter()
    for t, d, n in ev:
        dt = t - t_prev
        if dt > 0:
            at[min(level, 4)] += dt
            if level == 1:
                (only,) = [k for k, v in active.items() if v > 0]
                excl[only] += dt
        f = family(n)
        active[f] += d
        level += d
        t_prev = t
    span = seg[-1][1] - seg[0][0]
    ksum = sum(e - s for s, e, _ in seg)
    tot = collections.Counter()
    cnt = collections.Counter()
    for s, e, n in seg:
        tot[family(n)] += e - s
        cnt[family(n)] += 1
    ms = lambda v: v * 1e-6 / n_steps      # noqa: E731
    print('%d steps, %d dispatches per step: wall %.3f ms per step, summed kernel time %.3f ms, mean kernels in flight while busy %.2f'
          % (n_steps, len(seg) // n_steps, ms(span), ms(ksum), ksum / max(1, span - at[0])))
    print('time per step with k kernels in flight:  ' + '  '.join('%s: %.3f ms' % ('4+' if k == 4 else k, ms(at[k])) for k in range(5)))
    print('%-36s %10s %10s %8s' % ('kernel family', 'exclusive', 'total', 'launches'))
    for f, v in sorted(excl.items(), key=lambda kv: -kv[1])[:28]:
        print('%-36s %8.3f ms %8.3f ms %8.1f' % (f[:36], ms(v), ms(tot[f]), cnt[f] / n_steps))


if __name__ == '__main__':
    main()
