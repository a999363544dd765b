#!/bin/bash
# round 3, GPU call 5: BN last-block finish v2 (thread per channel) A/B + kernel stats of both; fused inference head tests
TAG=${1:-r3e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
echo "== A/B tickets"
for v in base:SEMSEG_BN_TICKETS=0 tick:SEMSEG_BN_TICKETS=1 base2:SEMSEG_BN_TICKETS=0 tick2:SEMSEG_BN_TICKETS=1; do
  name=${v%%:*}; kv=${v#*:}; IFS=, read -ra kvs <<< "$kv"
  env "${kvs[@]}" timeout 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(python -c "import json;d=json.load(open('$OUT/ab_$name.json'));print(d['ms_per_step'], d['value'])")"
done
ROOT=$PWD
for v in 0 1; do
  echo "== kernel stats, SEMSEG_BN_TICKETS=$v"
  ( cd /tmp && SEMSEG_BN_TICKETS=$v timeout 200 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof$v -o bench -- python $ROOT/bench.py --steps 16 --warmup 6 --no-cpu-baseline > $ROOT/$OUT/rocprof$v.log 2>&1 ); echo "rocprof rc=$?"
  db=$(find $OUT/prof$v -name '*.db' | head -1); tr=$(find $OUT/prof$v -name '*kernel_trace.csv' | head -1); src=${db:-$tr}
  python tools/rocprof_summary.py $src $OUT/kernel_stats_tickets$v.csv
  rm -rf $OUT/prof$v
  grep -i "bn_\|Name" $OUT/kernel_stats_tickets$v.csv | cut -d, -f1-4 | cut -c1-150 | head -20
done
echo "== tests"
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_drivers.py tests/test_gpu_models.py tests/test_gpu_eval_loop.py tests/test_gpu_metrics.py -m gpu -q -k "bn_ or drivers or golden or conv_bn or upsample_softmax or evaluate or inference" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log | cut -c1-300
grep -a "FAILED\|Error" $OUT/pytest_gpu.log | cut -c1-300 | head -30
echo "== inference bench"
timeout 300 python tools/bench_infer.py 2>&1 | tail -5 | cut -c1-300
