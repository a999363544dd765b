"""last N dispatches of a rocprofv3 kernel trace (rocpd .db or csv) as a small csv: start_us (relative), dur_us, queue / stream, name"""
import csv
import sqlite3
import sys


def main():
    path, out, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    if path.endswith('.db'):
        db = sqlite3.connect(path)
        cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
        extra = [c for c in ('queue_id', 'stream_id', 'queue', 'stream') if c in cols]
        rows = list(db.execute('select start, end, name%s from kernels order by start' % ''.join(', ' + c for c in extra)))
    else:
        with open(path) as f:
            rd = list(csv.DictReader(f))
        extra = [c for c in ('Queue_Id', 'Stream_Id') if c in rd[0]]
        rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) + tuple(r[c] for c in extra) for r in rd)
    rows = rows[-n:]
    t0 = rows[0][0]
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['start_us', 'dur_us', 'name'] + extra)
        for r in rows:
            w.writerow(['%.2f' % ((r[0] - t0) * 1e-3), '%.2f' % ((r[1] - r[0]) * 1e-3), r[2].split('(')[0][:70]] + list(r[3:]))


if __name__ == '__main__':
    main()
