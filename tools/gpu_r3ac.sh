#!/bin/bash
# round 3: ReLU gate bitmask for BN + residual (backward reads 1 bit instead of y): tests + in-box A/B (SEMSEG_GATE_MASK 0/1)
TAG=${1:-r3ac}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_ddp.py -m gpu -q -x -k "bn or golden or conv_bn_act or two_ranks_one_gpu" 2>&1 | tail -6 | cut -c1-300
for cfg in 1 3; do
  export SEMSEG_TUNE_CACHE=/tmp/plans_c$cfg.json
  for name in off on off2 on2; do
    case $name in off*) export SEMSEG_GATE_MASK=0;; *) export SEMSEG_GATE_MASK=1;; esac
    timeout 600 python bench.py --config $cfg --steps 30 --warmup 6 --no-cpu-baseline > $OUT/ab_c${cfg}_$name.json 2> $OUT/ab_c${cfg}_$name.err
    echo "cfg$cfg $name: $(python -c "import json;d=json.load(open('$OUT/ab_c${cfg}_$name.json'));print(d['ms_per_step'], d['value'], d['config']['final_loss'])")"
  done
done
