#!/bin/bash
# graph_steps A/B inside one box
TAG=${1:-r1k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
b() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 32 --warmup 6 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print(d['ms_per_step'], d['value'], d['config']['final_loss'])")"; tail -2 $OUT/bench_$name.err | cut -c1-200; }
b s1 SEMSEG_GRAPH_STEPS=1
b s2 SEMSEG_GRAPH_STEPS=2
b s4 SEMSEG_GRAPH_STEPS=4
b s1b SEMSEG_GRAPH_STEPS=1
b s8 SEMSEG_GRAPH_STEPS=8
