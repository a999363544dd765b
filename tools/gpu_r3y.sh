#!/bin/bash
# round 3: the pyramid-pooling scales on side streams (ops.run_branches in PPM / UPerNet): parity + in-box A/B
TAG=${1:-r3y}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "golden or full_size" 2>&1 | tail -4 | cut -c1-300
for cfg in 1 2; do
  export SEMSEG_TUNE_CACHE=/tmp/plans_c$cfg.json
  for name in off on off2 on2; do
    case $name in off*) export SEMSEG_BRANCH_STREAMS=0;; *) export SEMSEG_BRANCH_STREAMS=1;; esac
    timeout 600 python bench.py --config $cfg --steps 30 --warmup 6 --no-cpu-baseline > $OUT/ab_c${cfg}_$name.json 2> $OUT/ab_c${cfg}_$name.err
    echo "cfg$cfg $name: $(python -c "import json;d=json.load(open('$OUT/ab_c${cfg}_$name.json'));print(d['ms_per_step'], d['value'], d['config']['final_loss'])")"; grep -v amdgpu.ids $OUT/ab_c${cfg}_$name.err | tail -2 | cut -c1-200
  done
done
