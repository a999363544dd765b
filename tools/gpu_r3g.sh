#!/bin/bash
# round 3, GPU call 7: eval-mode Winograd (tests + inference bench), PMC passes of the dominant kernel (conv_last dgrad, h2) and
# of a north_star-named kernel (512->512 3x3 dilation 4 forward), HW-queue experiment for the 2-ranks-on-1-GPU segmented replay
TAG=${1:-r3g}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
echo "== inference bench"
timeout 300 python tools/bench_infer.py 2>&1 | grep images_per_sec | cut -c1-300 | tee $OUT/bench_infer.jsonl
SEMSEG_WINOGRAD=0 timeout 300 python tools/bench_infer.py 2>&1 | grep images_per_sec | cut -c1-300 | tee $OUT/bench_infer_nowino.jsonl
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_eval_loop.py -m gpu -q -k "absmax or eval_mode or golden or inference or evaluate or upsample" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-300
grep -a "FAILED\|Error" $OUT/pytest_gpu.log | cut -c1-300 | head -20
echo "== 2 ranks on 1 GPU, segmented, HW queue variants"
for q in default:X=1 q2:GPU_MAX_HW_QUEUES=2 q8:GPU_MAX_HW_QUEUES=8; do
  name=${q%%:*}; kv=${q#*:}
  env $kv timeout 300 python -m pytest tests/test_gpu_ddp.py -m gpu -q -s -k two_ranks_segmented 2>&1 | grep -a "2 ranks on 1 GPU\|passed\|failed" | cut -c1-200 | sed "s/^/$name: /"
done
echo "== PMC: conv_last dgrad h2 (dominant launch of the step)"
MODE=h2 TILE=8 SPLIT=1 bash tools/gpu_pmc.sh $TAG/pmc_conv_last_dgrad conv_last dgrad > $OUT/pmc_conv_last_dgrad.txt 2>&1; tail -40 $OUT/pmc_conv_last_dgrad.txt | cut -c1-200
echo "== PMC: layer4 512->512 3x3 d4 fwd h2"
MODE=h2 TILE=10 SPLIT=2 bash tools/gpu_pmc.sh $TAG/pmc_l4_d4_fwd l4_conv2_d4 fwd > $OUT/pmc_l4_d4_fwd.txt 2>&1; tail -40 $OUT/pmc_l4_d4_fwd.txt | cut -c1-200
