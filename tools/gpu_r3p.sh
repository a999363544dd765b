#!/bin/bash
# round 3, GPU call 17: conv + BN statistics through one entry point (split-K reduce folded into the statistics sweep), tuner play-off
TAG=${1:-r3p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -k "bnstats or conv_bn_act or golden or bn_h2" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log | cut -c1-300
grep -a "FAILED\|Error" $OUT/pytest.log | head -10 | cut -c1-250
echo "== step A/B (SEMSEG_CONV_STATS 0 / 1)"
for cfg in 1 4; do
  export SEMSEG_TUNE_CACHE=/tmp/plans_c$cfg.json
  for name in off on off2 on2; do
    case $name in off*) export SEMSEG_CONV_STATS=0;; *) export SEMSEG_CONV_STATS=1;; esac
    timeout 600 python bench.py --config $cfg --steps 30 --warmup 6 --no-cpu-baseline > $OUT/ab_c${cfg}_$name.json 2> $OUT/ab_c${cfg}_$name.err
    echo "cfg$cfg $name: $(python -c "import json;d=json.load(open('$OUT/ab_c${cfg}_$name.json'));print(d['ms_per_step'], d['value'])")"
  done
done
