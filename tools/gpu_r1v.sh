#!/bin/bash
# functional check of bench.py's N>1 path on the one GPU: 2 ranks, gloo collectives, exactly the driver's launch line
TAG=${1:-r1v}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
echo "== 1 rank eager (reference point)"
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-graph --no-cpu-baseline > $OUT/bench_1.json 2> $OUT/bench_1.err; cut -c1-200 $OUT/bench_1.json
echo "== 2 ranks on one GPU (gloo)"
SEMSEG_DIST_BACKEND=gloo SEMSEG_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 3 > $OUT/bench_2.json 2> $OUT/bench_2.err; echo "rc=$?"; cat $OUT/bench_2.json | cut -c1-700; tail -5 $OUT/bench_2.err | cut -c1-300
echo "== world-1 nccl through torchrun (RCCL init + SyncBN-off path)"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline > $OUT/bench_1t.json 2> $OUT/bench_1t.err; echo "rc=$?"; cut -c1-200 $OUT/bench_1t.json
