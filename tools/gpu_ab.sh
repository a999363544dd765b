#!/bin/bash
# A/B of env switches inside one box: bash tools/gpu_ab.sh <tag> NAME1:VAR=VAL NAME2:VAR=VAL ... (each run twice, interleaved)
TAG=${1:-ab}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
b() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print(d['ms_per_step'], d['value'])")"; }
for rep in 1 2; do
  for spec in "$@"; do
    name=${spec%%:*}; kv=${spec#*:}
    b ${name}_$rep $kv
  done
done
