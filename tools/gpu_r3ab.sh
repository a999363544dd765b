#!/bin/bash
# round 3: where does a forked hipGraph lose its time?  kernel traces of the configs[1] step, pyramid scales on one stream vs on side streams
TAG=${1:-r3ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/plans_c1.json
timeout 300 python bench.py --steps 10 --warmup 6 --no-cpu-baseline > /dev/null 2>&1     # fills the plan cache
for name in linear forked; do
  case $name in linear) export SEMSEG_PPM_STREAMS=0;; *) export SEMSEG_PPM_STREAMS=1;; esac
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $ROOT/$OUT/prof_$name -o bench -- python $ROOT/bench.py --steps 10 --warmup 6 --no-cpu-baseline > $ROOT/$OUT/rocprof_$name.log 2>&1 )
  grep -a "^{" $OUT/rocprof_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$name', d['ms_per_step'])"
  db=$(find $OUT/prof_$name -name '*.db' | head -1); tr=$(find $OUT/prof_$name -name '*kernel_trace.csv' | head -1); src=${db:-$tr}
  python tools/trace_gaps.py $src 0.3 > $OUT/trace_gaps_$name.txt; head -12 $OUT/trace_gaps_$name.txt
  python tools/trace_dump.py $src $OUT/trace_$name.csv 4000
  rm -rf $OUT/prof_$name
done
