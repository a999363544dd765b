#!/bin/bash
# round 3, GPU call 2: full suite (anchor-based goldens, full-size configs 1-4, segmented DDP graphs), bench of every config,
# Winograd threshold A/B, kernel stats of the step
TAG=${1:-r3b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
echo "== bench config 1 (cool GPU first)"
timeout 600 python bench.py --steps 40 --warmup 6 > $OUT/bench_cfg1.json 2> $OUT/bench_cfg1.err; echo "rc=$?"; cut -c1-600 $OUT/bench_cfg1.json; tail -3 $OUT/bench_cfg1.err
echo "== full gpu suite"
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-300
grep -a "error / allowed\|max|dlogp|\|segmented\|2 ranks on 1 GPU\|FAILED\|Error" $OUT/pytest_gpu.log | cut -c1-260 | head -60
for c in 2 3 4; do
  echo "== bench config $c"
  timeout 600 python bench.py --config $c --steps 30 --warmup 8 --no-cpu-baseline > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; echo "rc=$?"; cut -c1-900 $OUT/bench_cfg$c.json; tail -3 $OUT/bench_cfg$c.err
done
echo "== config 3, long run (graph cache fills): 200 steps"
timeout 600 python bench.py --config 3 --steps 200 --warmup 100 --no-cpu-baseline > $OUT/bench_cfg3_long.json 2> $OUT/bench_cfg3_long.err; echo "rc=$?"; cut -c1-900 $OUT/bench_cfg3_long.json; tail -3 $OUT/bench_cfg3_long.err
echo "== A/B winograd threshold"
for v in base:X=1 wino512:SEMSEG_WINOGRAD_MIN_C=512,SEMSEG_TUNE_CACHE=/tmp/plans_w512.json base2:X=1 wino512b:SEMSEG_WINOGRAD_MIN_C=512,SEMSEG_TUNE_CACHE=/tmp/plans_w512.json; do
  name=${v%%:*}; kv=${v#*:}; IFS=, read -ra kvs <<< "$kv"
  env "${kvs[@]}" timeout 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(python -c "import json;d=json.load(open('$OUT/ab_$name.json'));print(d['ms_per_step'], d['value'])")"
done
echo "== kernel stats of the step"
ROOT=$PWD
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 16 --warmup 6 --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 ); echo "rocprof rc=$?"
db=$(find $OUT/prof -name '*.db' | head -1); tr=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
src=${db:-$tr}
python tools/rocprof_summary.py $src $OUT/kernel_stats.csv
python tools/rocprof_summary.py $src $OUT/kernel_stats_by_grid.csv --by-grid
head -30 $OUT/kernel_stats.csv | cut -c1-160
cp /tmp/semseg_plans_h2.json $OUT/plans_h2.json
rm -rf $OUT/prof
