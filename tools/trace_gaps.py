"""GPU-busy analysis of a rocprofv3 --kernel-trace run (rocpd .db or *_kernel_trace.csv): over the steady-state part
of the trace (last `frac` of the dispatches) report wall span, summed kernel time, idle time between consecutive
kernels, overlap (concurrent kernels) and a histogram of the gaps -- tells whether a step is bound by kernel time or by
launch gaps (hipGraph replay vs eager).

    python tools/trace_gaps.py gpurun_out/<tag>/prof_graph/bench_results.db [frac=0.5]
"""
import csv
import sqlite3
import sys


def load(path):
    if path.endswith('.db'):
        db = sqlite3.connect(path)
        return sorted((s, e, n) for n, s, e in db.execute('select name, start, end from kernels'))
    with open(path) as f:
        return sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(f))


def main():
    rows = load(sys.argv[1])
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows = rows[int(len(rows) * (1 - frac)):]
    span = rows[-1][1] - rows[0][0]
    busy = 0
    cur_s, cur_e = rows[0][0], rows[0][1]
    gaps = []
    big = []
    prev = rows[0][2]
    for s, e, n in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            big.append((s - cur_e, prev, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        prev = n
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e, _ in rows)
    print('dispatches %d  span %.3f ms  union-busy %.3f ms (%.1f %%)  sum of kernel durations %.3f ms  idle %.3f ms'
          % (len(rows), span * 1e-6, busy * 1e-6, 100.0 * busy / span, ksum * 1e-6, (span - busy) * 1e-6))
    edges = [1000, 2000, 4000, 8000, 16000, 50000, 10 ** 12]
    hist = [0] * len(edges)
    tot = [0] * len(edges)
    for g in gaps:
        for i, e in enumerate(edges):
            if g < e:
                hist[i] += 1
                tot[i] += g
                break
    lo = 0
    for e, h, t in zip(edges, hist, tot):
        print('  gaps %6.1f..%-8s us: %6d  total %.3f ms' % (lo * 1e-3, ('%.1f' % (e * 1e-3)) if e < 10 ** 11 else 'inf', h, t * 1e-6))
        lo = e
    print('largest gaps (us): after kernel -> before kernel')
    for g, a, b in sorted(big, reverse=True)[:12]:
        print('  %8.1f  %s -> %s' % (g * 1e-3, a[:60], b[:60]))


def _tail(rows):
    pass


if __name__ == '__main__':
    main()
