OUT=gpurun_out/r6N; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
timeout 600 python bench.py --config 3 --shapes 0 --steps 300 --warmup 100 --no-cpu-baseline --no-other-configs --repeats 0 --no-box --no-scaling-model > $OUT/raw.json 2> $OUT/raw.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6N/raw.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['config']['shape_events_timed'], d['config']['launch_plans'])
PY
tail -3 $OUT/raw.err
