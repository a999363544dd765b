#!/bin/bash
# round 3, GPU call 4: BN last-block finish (tickets): parity tests + in-box A/B; drivers test
TAG=${1:-r3d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
echo "== A/B tickets (bench first, cool GPU)"
for v in base:SEMSEG_BN_TICKETS=0 tick:SEMSEG_BN_TICKETS=1 base2:SEMSEG_BN_TICKETS=0 tick2:SEMSEG_BN_TICKETS=1; do
  name=${v%%:*}; kv=${v#*:}; IFS=, read -ra kvs <<< "$kv"
  env "${kvs[@]}" timeout 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(python -c "import json;d=json.load(open('$OUT/ab_$name.json'));print(d['ms_per_step'], d['value'])")"
done
echo "== tests"
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_drivers.py tests/test_gpu_models.py -m gpu -q -k "bn_ or drivers or golden or conv_bn or full_size" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log | cut -c1-300
grep -a "FAILED\|Error" $OUT/pytest_gpu.log | cut -c1-300 | head -30
