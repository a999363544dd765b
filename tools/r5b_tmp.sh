bash tools/gpu_run.sh r5v "test:many_problems or deferred_wgrad or DEFER_WGRAD or DEPTHWISE_DIRECT or hrnet_branch_streams or (golden and hrnetv2_c1_128_train)" ab:perlayer:SEMSEG_DEFER_WGRAD_LAUNCH=0 ab:batched:X=1 ab:perlayer:SEMSEG_DEFER_WGRAD_LAUNCH=0 ab:batched:X=1
for i in 1 2; do
SEMSEG_DEFER_WGRAD_LAUNCH=0 timeout 400 python bench.py --config 4 --steps 30 --warmup 6 --no-cpu-baseline --repeats 0 --no-box --no-scaling-model 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg4 per-layer wgrad launches', d['ms_per_step'], d['value'])"
timeout 400 python bench.py --config 4 --steps 30 --warmup 6 --no-cpu-baseline --repeats 0 --no-box --no-scaling-model 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg4 batched small wgrads', d['ms_per_step'], d['value'])"
done
