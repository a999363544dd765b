#!/bin/bash
# round 3, GPU call 12: kernel statistics of every BASELINE config's bench command (final code)
TAG=${1:-r3l}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
ROOT=$PWD
for c in 1 2 3 4; do
  export SEMSEG_TUNE_CACHE=/tmp/plans_c$c.json
  timeout 600 python bench.py --config $c --steps 10 --warmup 6 --no-cpu-baseline > $OUT/warm_cfg$c.json 2> $OUT/warm_cfg$c.err     # fills the plan cache
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof$c -o bench -- python $ROOT/bench.py --config $c --steps 16 --warmup 6 --no-cpu-baseline > $ROOT/$OUT/rocprof_cfg$c.log 2>&1 ); echo "cfg $c rocprof rc=$?"
  db=$(find $OUT/prof$c -name '*.db' | head -1); tr=$(find $OUT/prof$c -name '*kernel_trace.csv' | head -1); src=${db:-$tr}
  python tools/rocprof_summary.py $src $OUT/kernel_stats_cfg$c.csv
  python tools/rocprof_summary.py $src $OUT/kernel_stats_by_grid_cfg$c.csv --by-grid
  rm -rf $OUT/prof$c
  grep -a "^{" $OUT/rocprof_cfg$c.log | cut -c1-200
  head -12 $OUT/kernel_stats_cfg$c.csv | cut -d, -f1-5 | cut -c1-170
done
