"""Where does a kernel sit in the step?  For every dispatch whose name contains `pattern`, count the (previous kernel ->
next kernel) neighbourhoods over the steady-state part of a rocprofv3 --kernel-trace run (rocpd .db or *_kernel_trace.csv).

    python tools/trace_context.py <trace> <pattern> [frac=0.5]
"""
import sys
from collections import Counter

from trace_gaps import load


def short(n):
    n = n.replace('void ', '')
    return n[:70]


def main():
    rows = load(sys.argv[1])
    pat = sys.argv[2]
    frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
    rows = rows[int(len(rows) * (1 - frac)):]
    ctx = Counter()
    dur = Counter()
    for i, (s, e, n) in enumerate(rows):
        if pat in n:
            key = (short(rows[i - 1][2]) if i else '-', short(rows[i + 1][2]) if i + 1 < len(rows) else '-')
            ctx[key] += 1
            dur[key] += e - s
    print('%d dispatches match %r' % (sum(ctx.values()), pat))
    for key, c in ctx.most_common(40):
        print('%5d x %7.1f us   %s  ->  [%s]  ->  %s' % (c, dur[key] / c * 1e-3, key[0], pat, key[1]))


if __name__ == '__main__':
    main()
