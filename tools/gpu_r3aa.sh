#!/bin/bash
# round 3: the N>1 self-test script and the N=2 bench path on the one GPU (gloo transport, IPC peer exchange), after the last changes
TAG=${1:-r3aa}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_DIST_BACKEND=gloo SEMSEG_BENCH_DEVICE=0 GPU_MAX_HW_QUEUES=2 SEMSEG_PEER_TIMEOUT_S=60 SEMSEG_TUNE=0
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/probes/ddp_graph_selftest.py > $OUT/selftest.log 2>&1; echo "selftest rc=$? (the whole-step graph stage cannot pass on gloo: expected non-zero)"
grep -a "_OK\|_OFF\|Error\|error" $OUT/selftest.log | cut -c1-220 | head -12
unset SEMSEG_TUNE; export SEMSEG_TUNE_CACHE=/tmp/plans.json
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 4 > $OUT/bench_n2.json 2> $OUT/bench_n2.err; echo "n2 rc=$?"
grep -a "^{" $OUT/bench_n2.json | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['n_gpus'], d['config']['launch'], d['config']['collectives'], d['config']['final_loss'])"
grep -v "amdgpu.ids\|socket.cpp\|capture_end\|CUDA Graph is empty\|OMP_NUM\|\*\*\*\*" $OUT/bench_n2.err | tail -5 | cut -c1-300
