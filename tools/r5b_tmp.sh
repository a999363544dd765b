bash tools/gpu_run.sh r5t "test:deferred_wgrad or DEFER_WGRAD or (golden and r50d_ppmds_64_train) or hrnet_branch_streams" ab:per:SEMSEG_DEFER_WGRAD_REDUCE=0 ab:multi:X=1 ab:per:SEMSEG_DEFER_WGRAD_REDUCE=0 ab:multi:X=1
for i in 1 2; do
SEMSEG_DEFER_WGRAD_REDUCE=0 timeout 400 python bench.py --config 4 --steps 30 --warmup 6 --no-cpu-baseline --repeats 0 --no-box --no-scaling-model 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg4 per-tensor reduces', d['ms_per_step'], d['value'])"
timeout 400 python bench.py --config 4 --steps 30 --warmup 6 --no-cpu-baseline --repeats 0 --no-box --no-scaling-model 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg4 one multi-tensor reduce', d['ms_per_step'], d['value'])"
done
