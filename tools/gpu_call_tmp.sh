OUT=gpurun_out/r6X; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
timeout 400 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_probes.py tests/test_gpu_models.py -q -x -k "per_shape_train_graphs or timeline or probe or graph or replay" 2>&1 | grep -v amdgpu.ids | tail -8
for fs in 1; do
SEMSEG_CAPTURE_FIRST_SIGHT=$fs timeout 600 python bench.py --config 3 --shapes 0 --steps 300 --warmup 100 --no-cpu-baseline --no-other-configs --repeats 0 --no-box --no-scaling-model > $OUT/raw$fs.json 2> $OUT/raw$fs.err
python - $fs <<'PY'
import json, sys
fs = sys.argv[1]
d=json.loads([l for l in open('gpurun_out/r6X/raw%s.json' % fs) if l.startswith('{')][-1])
print('first_sight=%s' % fs, d['value'], d['ms_per_step'], d['config']['shape_events_timed'], d['config']['launch_plans'], d['config']['final_loss'])
PY
tail -2 $OUT/raw$fs.err
done
