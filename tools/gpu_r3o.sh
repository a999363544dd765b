#!/bin/bash
# round 3, GPU call 15: weight-gradient split cap 64 vs 256 (the long-M stem layers)
TAG=${1:-r3o}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
for name in base ws256 base2 ws256b; do
  case $name in base*) unset SEMSEG_WGRAD_MAX_SPLIT; export SEMSEG_TUNE_CACHE=/tmp/plans_base.json;; *) export SEMSEG_WGRAD_MAX_SPLIT=256; export SEMSEG_TUNE_CACHE=/tmp/plans_ws.json;; esac
  timeout 600 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(python -c "import json;d=json.load(open('$OUT/ab_$name.json'));print(d['ms_per_step'], d['value'])")"
done
python - <<'PY'
import json
a=json.load(open('/tmp/plans_base.json')); b=json.load(open('/tmp/plans_ws.json'))
for k in sorted(b):
    if k.split(',')[1]=='2' and k in a and b[k][1]>64: print(k, 'base', a[k], 'ws256', b[k])
PY
