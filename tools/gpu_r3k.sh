#!/bin/bash
# round 3, GPU call 11: 16-wave tile inside the Winograd GEMMs (forward: SEMSEG_WINO_TILE=14, weight gradient: SEMSEG_WINO_W16=1)
TAG=${1:-r3k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/plans.json
for v in base:X=1 wf14:SEMSEG_WINO_TILE=14 ww16:SEMSEG_WINO_W16=1 both:SEMSEG_WINO_TILE=14,SEMSEG_WINO_W16=1 base2:X=1 both2:SEMSEG_WINO_TILE=14,SEMSEG_WINO_W16=1; do
  name=${v%%:*}; kv=${v#*:}; IFS=, read -ra kvs <<< "$kv"
  env "${kvs[@]}" timeout 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(python -c "import json;d=json.load(open('$OUT/ab_$name.json'));print(d['ms_per_step'], d['value'])")"
done
SEMSEG_WINO_TILE=14 SEMSEG_WINO_W16=1 timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "winograd" 2>&1 | tail -2
