#!/bin/bash
# round 3, GPU call 18: the N > 1 code path of bench.py end to end on the one GPU (2 ranks, gloo transport): segmented executor,
# JSON line; + default bench for reference
TAG=${1:-r3q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/plans.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "n1 rc=$?"; cut -c1-200 $OUT/bench_n1.json
for mode in segmented:SEMSEG_DDP_SEGMENTED=1 eager:SEMSEG_DDP_SEGMENTED=0; do
  name=${mode%%:*}; kv=${mode#*:}
  env $kv SEMSEG_DIST_BACKEND=gloo SEMSEG_BENCH_DEVICE=0 GPU_MAX_HW_QUEUES=2 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 4 > $OUT/bench_n2_$name.json 2> $OUT/bench_n2_$name.err; echo "n2 $name rc=$?"
  grep -a "^{" $OUT/bench_n2_$name.json | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['n_gpus'], d['config']['launch'], d['config']['collectives'], d['config']['final_loss'])"
  tail -3 $OUT/bench_n2_$name.err | cut -c1-200
done
