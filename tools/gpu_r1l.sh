#!/bin/bash
TAG=${1:-r1l}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
timeout 400 python bench.py --steps 8 --warmup 6 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 6 --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 )
db=$(find $OUT/prof -name '*.db' | head -1); tr=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
src=${db:-$tr}
python tools/trace_gaps.py $src 0.3 | tee $OUT/trace_gaps_graph.txt
ls $OUT/prof/* | head; python - <<PY
import sqlite3,sys,glob
dbs=glob.glob('$OUT/prof/**/*.db', recursive=True)
if dbs:
    db=sqlite3.connect(dbs[0])
    print([r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")][:60])
    try:
        for r in db.execute("select name, count(*), avg(end-start), min(size), max(size), avg(size) from memory_copies group by name"): print(r)
    except Exception as e: print('memcopy query failed', e)
PY
rm -rf $OUT/prof
