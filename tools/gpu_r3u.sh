#!/bin/bash
# round 3, GPU call 22: bn_bwd_apply_h2 with per-thread channel ownership (per-channel terms loaded once per thread): tests + in-box A/B
TAG=${1:-r3u}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x -k "bn or golden or conv_bn_act" 2>&1 | tail -4 | cut -c1-300
BASE=$PWD/semantic-segmentation-pytorch_amd/mit_semseg/_native/variants/libsemseg_hip_base.so
for cfg in 1 4; do
  export SEMSEG_TUNE_CACHE=/tmp/plans_c$cfg.json
  for name in base new base2 new2; do
    case $name in base*) export SEMSEG_NATIVE_LIB=$BASE;; *) unset SEMSEG_NATIVE_LIB;; esac
    timeout 600 python bench.py --config $cfg --steps 30 --warmup 6 --no-cpu-baseline > $OUT/ab_c${cfg}_$name.json 2> $OUT/ab_c${cfg}_$name.err
    echo "cfg$cfg $name: $(python -c "import json;d=json.load(open('$OUT/ab_c${cfg}_$name.json'));print(d['ms_per_step'], d['value'])")"
  done
done
