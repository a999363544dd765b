#!/bin/bash
# round 3, GPU call 6: full suite after the revert + forks + fused head; bench of all configs (profiles of the round)
TAG=${1:-r3f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
echo "== bench config 1"
timeout 600 python bench.py --steps 40 --warmup 6 > $OUT/bench_cfg1.json 2> $OUT/bench_cfg1.err; echo "rc=$?"; cut -c1-400 $OUT/bench_cfg1.json; tail -2 $OUT/bench_cfg1.err
echo "== full gpu suite"
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-300
grep -a "per-shape graphs\|segmented\|2 ranks on 1 GPU\|FAILED\|Error" $OUT/pytest_gpu.log | cut -c1-300 | head -40
for c in 2 3 4; do
  echo "== bench config $c"
  timeout 900 python bench.py --config $c --steps 40 --warmup 8 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; echo "rc=$?"; cut -c1-300 $OUT/bench_cfg$c.json; tail -2 $OUT/bench_cfg$c.err
done
echo "== inference bench"
timeout 300 python tools/bench_infer.py 2>&1 | grep images_per_sec | cut -c1-300 | tee $OUT/bench_infer.jsonl
