#!/bin/bash
# round 3, GPU call 19: one-node peer exchange (csrc/peer.hip): ABI test, two ranks on the one GPU (IPC-mapped inboxes), N=2 bench path
TAG=${1:-r3r}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_ddp.py -x -q -s 2>&1 | tail -25 | cut -c1-300 > $OUT/pytest_ddp.txt; cat $OUT/pytest_ddp.txt
export SEMSEG_TUNE_CACHE=/tmp/plans.json
for mode in peer_segmented:SEMSEG_PEER=1 rccl_path_segmented:SEMSEG_PEER=0; do
  name=${mode%%:*}; kv=${mode#*:}
  env $kv SEMSEG_PEER_TIMEOUT_S=60 SEMSEG_DIST_BACKEND=gloo SEMSEG_BENCH_DEVICE=0 GPU_MAX_HW_QUEUES=2 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 4 > $OUT/bench_n2_$name.json 2> $OUT/bench_n2_$name.err; echo "n2 $name rc=$?"
  grep -a "^{" $OUT/bench_n2_$name.json | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['n_gpus'], d['config']['launch'], d['config']['collectives'], d['config']['final_loss'])"
  grep -v "amdgpu.ids\|socket.cpp\|capture_end\|CUDA Graph is empty" $OUT/bench_n2_$name.err | tail -5 | cut -c1-300
done
