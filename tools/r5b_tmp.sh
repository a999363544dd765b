bash tools/gpu_run.sh r5y "test:many_problems or deferred_wgrad or WGRAD_MULTI or DEFER_WGRAD_LAUNCH" ab:args:SEMSEG_WGRAD_MULTI_TABLE=0 ab:table:X=1 ab:args:SEMSEG_WGRAD_MULTI_TABLE=0 ab:table:X=1
for i in 1 2; do
SEMSEG_WGRAD_MULTI_TABLE=0 timeout 400 python bench.py --config 4 --steps 30 --warmup 6 --no-cpu-baseline --repeats 0 --no-box --no-scaling-model 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg4 24 per launch (args)', d['ms_per_step'], d['value'])"
timeout 400 python bench.py --config 4 --steps 30 --warmup 6 --no-cpu-baseline --repeats 0 --no-box --no-scaling-model 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg4 one launch (table)', d['ms_per_step'], d['value'])"
done
