#!/bin/bash
# A/B inside one box: k order (chunk-major vs tap-major), graph copies on own streams vs one stream; PMC of the dominant conv
TAG=${1:-r1j}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== pytest -m gpu (without the full-size oracle test)"
timeout 600 python -m pytest tests -m gpu -q -x -k "not test_config1_full_size" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.log | cut -c1-300
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
b() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print(d['ms_per_step'], d['value'], d['roofline']['achieved'])")"; }
b default X=1
b tapmajor SEMSEG_TAP_MAJOR=1
b onestream SEMSEG_GRAPH_STREAMS=0
b default2 X=1
b tapmajor2 SEMSEG_TAP_MAJOR=1
b onecopy SEMSEG_GRAPH_COPIES=1
cp /tmp/semseg_plans_h2.json $OUT/plans_h2.json
echo "== PMC on conv_last fwd (h2, tile 5 split 4)"
MODE=h2 TILE=5 SPLIT=4 bash tools/gpu_pmc.sh $TAG/pmc conv_last fwd 2>&1 | grep -A 22 "igemm_dma"
du -sh $OUT
