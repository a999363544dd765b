#!/bin/bash
# round 3: HRNet's parallel branches on side streams (ops.run_branches): golden parity + in-box A/B (SEMSEG_BRANCH_STREAMS 0/1), inference
TAG=${1:-r3w}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_eval_loop.py -m gpu -q -x -k "hrnet" 2>&1 | tail -4 | cut -c1-300
export SEMSEG_TUNE_CACHE=/tmp/plans_c4.json
for name in off on off2 on2; do
  case $name in off*) export SEMSEG_BRANCH_STREAMS=0;; *) export SEMSEG_BRANCH_STREAMS=1;; esac
  timeout 600 python bench.py --config 4 --steps 30 --warmup 6 --no-cpu-baseline > $OUT/ab_c4_$name.json 2> $OUT/ab_c4_$name.err
  echo "cfg4 $name: $(python -c "import json;d=json.load(open('$OUT/ab_c4_$name.json'));print(d['ms_per_step'], d['value'], d['config']['final_loss'])")"; grep -v amdgpu.ids $OUT/ab_c4_$name.err | tail -2 | cut -c1-200
done
