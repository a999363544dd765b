#!/bin/bash
# A/B of env switches inside one box: bash tools/gpu_ab.sh <tag> NAME1:VAR=VAL[,VAR2=VAL2] NAME2:VAR=VAL ... (each run twice,
# interleaved).  Variants that change what the tuner may pick need their own plan cache, e.g.
#   bash tools/gpu_ab.sh r3a base:X=1 wino512:SEMSEG_WINOGRAD_MIN_C=512 wsplit:SEMSEG_WGRAD_MAX_SPLIT=256,SEMSEG_TUNE_CACHE=/tmp/plans_wsplit.json
TAG=${1:-ab}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
b() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print(d['ms_per_step'], d['value'])")"; }
for rep in 1 2; do
  for spec in "$@"; do
    name=${spec%%:*}; kv=${spec#*:}
    IFS=, read -ra kvs <<< "$kv"
    b ${name}_$rep "${kvs[@]}"
  done
done
