"""Per-kernel HBM traffic of ONE training step from the rocprofv3 --pmc passes of tools/gpu_pmc_step.sh (one directory per pass:
fetch/ = FETCH_SIZE, write/ = WRITE_SIZE).  The workload (tools/probes/step_traffic.py) repeats the same eager step, so the last
step is the shortest suffix of the dispatch sequence that repeats; its dispatches are summed per kernel name
(a few dispatches that read the loss back may follow the last step).  FETCH_SIZE is
doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies the 128-byte requests of wide streaming reads at 64 bytes); both
counters are in KiB.    python tools/pmc_step_summary.py gpurun_out/<tag> [step_ms]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def dispatches(root, sub, counter):
    rows = {}
    for f in glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r['Counter_Name'] != counter:
                    continue
                d = int(r['Dispatch_Id'])
                name, v = rows.get(d, (r['Kernel_Name'], 0.0))
                rows[d] = (name, v + float(r['Counter_Value']))
    return [rows[d] for d in sorted(rows)]


def last_period(names, shortest=50, max_tail=16):
    """(p, tail): names[-p-tail:-tail] is the last full step; `tail` dispatches (reading the loss back) follow it"""
    for tail in range(max_tail + 1):
        body = names[:len(names) - tail]
        for p in range(shortest, len(body) // 2 + 1):
            if body[-p:] == body[-2 * p:-p]:
                return p, tail
    return None


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', name)
    return name[:64]


def main():
    root = sys.argv[1]
    step_ms = float(sys.argv[2]) if len(sys.argv) > 2 else None
    table = defaultdict(lambda: [0, 0.0, 0.0])
    for col, (sub, counter, scale) in enumerate((('fetch', 'FETCH_SIZE', 2.0), ('write', 'WRITE_SIZE', 1.0))):
        seq = dispatches(root, sub, counter)
        found = last_period([n for n, _ in seq])
        if found is None:
            print('%s: no repeating step in %d dispatches' % (sub, len(seq)))
            continue
        p, tail = found
        print('%s pass: %d dispatches, %d per step' % (sub, len(seq), p))
        for name, v in seq[len(seq) - tail - p:len(seq) - tail]:
            row = table[short(name)]
            if col == 0:
                row[0] += 1
            row[1 + col] += v * scale * 1024 / 1e6
    tot_f = sum(r[1] for r in table.values())
    tot_w = sum(r[2] for r in table.values())
    print('%-66s %7s %10s %10s %10s' % ('kernel', 'n/step', 'read MB', 'write MB', 'sum MB'))
    for name, (n, f, w) in sorted(table.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print('%-66s %7d %10.1f %10.1f %10.1f' % (name, n, f, w, f + w))
    print('%-66s %7d %10.1f %10.1f %10.1f' % ('TOTAL', sum(r[0] for r in table.values()), tot_f, tot_w, tot_f + tot_w))
    if step_ms:
        print('step of %.2f ms (hipGraph replay, bench.py): %.2f TB/s average over the step (8 TB/s peak, ~6.3 achievable)' % (
            step_ms, (tot_f + tot_w) / 1e6 / (step_ms * 1e-3)))


if __name__ == '__main__':
    main()
