OUT=gpurun_out/r6Y; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6Y/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['cpu_baseline']['value'])
for k,v in d['config']['other_configs'].items():
    print(k, v.get('img_s'), v.get('ms_per_step'), {kk: vv for kk, vv in (v.get('raw_stream') or {}).items() if kk in ('img_s','fraction_of_steady_state','events_timed','host_ms_per_capture')})
PY
tail -2 $OUT/bench.err
