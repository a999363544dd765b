OUT=gpurun_out/r6f; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
b() { name=$1; cfg=$2; shift; shift; env "$@" timeout 600 python bench.py --config $cfg --steps 30 --warmup 6 --no-cpu-baseline --no-other-configs --no-box --no-scaling-model --repeats 1 > $OUT/bench_$name.json 2> $OUT/bench_$name.err; python - $OUT/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], d['ms_per_step'],'ms', d['config']['repeat_windows']['ms_per_step'], d['config']['launch_plans'])
except Exception as e: print(sys.argv[2],'no line',e)
PY
}
b tuneA 4 SEMSEG_TUNE_DB=0 SEMSEG_TUNE_CACHE=/tmp/a.json
b tuneB 4 SEMSEG_TUNE_DB=0 SEMSEG_TUNE_CACHE=/tmp/b.json SEMSEG_TUNE_MAX_SPLITK=1
b A1 4 SEMSEG_TUNE_DB=0 SEMSEG_TUNE_CACHE=/tmp/a.json; b B1 4 SEMSEG_TUNE_DB=0 SEMSEG_TUNE_CACHE=/tmp/b.json SEMSEG_TUNE_MAX_SPLITK=1
b A2 4 SEMSEG_TUNE_DB=0 SEMSEG_TUNE_CACHE=/tmp/a.json; b B2 4 SEMSEG_TUNE_DB=0 SEMSEG_TUNE_CACHE=/tmp/b.json SEMSEG_TUNE_MAX_SPLITK=1
b C1a 1 SEMSEG_TUNE_DB=0 SEMSEG_TUNE_CACHE=/tmp/c.json SEMSEG_TUNE_MAX_SPLITK=1; b C1b 1 SEMSEG_TUNE_DB=0 SEMSEG_TUNE_CACHE=/tmp/c.json SEMSEG_TUNE_MAX_SPLITK=1; b D1 1 X=1
cp /tmp/b.json $OUT/plans_nosplit_cfg4.json; cp /tmp/a.json $OUT/plans_default_cfg4.json
