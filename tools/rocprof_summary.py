"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db or *_kernel_trace.csv) into a per-kernel CSV
(the same columns as rocprofv3's kernel_stats.csv) so that a small text file can be committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/r1a/prof/bench_results.db profiles/r1_bench_kernel_stats.csv [--by-grid]

--by-grid keeps launches of one kernel with different grids apart (name suffix " grid=GX*GY*GZ/WG", .db input only): the
conv kernels are shared by many layers, the dominant layer (decoder.conv_last.0) is the row with its grid.
"""
import csv
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    db = sqlite3.connect(path)
    for name, start, end, gx, gy, gz, wx, vg, ag, sg, lds in db.execute(
            'select name, start, end, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, '
            'lds_size from kernels'):
        if BY_GRID:
            name = '%s grid=%d*%d*%d/%d' % (name, gx, gy, gz, wx)
        yield name, end - start, (vg, ag, sg, lds)


def rows_from_csv(path):
    with open(path) as f:
        for r in csv.DictReader(f):
            yield r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']), (
                r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('SGPR_Count'), r.get('LDS_Block_Size'))


BY_GRID = False


def main():
    global BY_GRID
    BY_GRID = '--by-grid' in sys.argv
    src, dst = sys.argv[1], sys.argv[2]
    it = rows_from_db(src) if src.endswith('.db') else rows_from_csv(src)
    agg = defaultdict(list)
    res = {}
    for name, dur, r in it:
        agg[name].append(dur)
        res[name] = r
    if BY_GRID:
        # one kernel + grid can still serve several layers (e.g. two 3x3 convs with 64 tiles x 4 splits): separate the
        # duration clusters (consecutive sorted durations more than 1.5x apart) so that every row is one launch shape
        split = {}
        for name, v in agg.items():
            v = sorted(v)
            cl = [[v[0]]]
            for d in v[1:]:
                if d > 1.5 * cl[-1][-1]:
                    cl.append([d])
                else:
                    cl[-1].append(d)
            for c in cl:
                key = name if len(cl) == 1 else '%s ~%dus' % (name, round(sum(c) / len(c) / 1e3))
                split[key] = c
                res[key] = res[name]
        agg = split
    total = sum(sum(v) for v in agg.values())
    with open(dst, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'VGPR', 'AGPR', 'SGPR',
                    'LDS'])
        for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / total, 3), min(v), max(v)] +
                       list(res[name]))
    print('wrote %s (%d kernels, %.3f ms total)' % (dst, len(agg), total * 1e-6))


if __name__ == '__main__':
    main()
