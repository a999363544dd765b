#!/bin/bash
# Regenerate the shipped performance database (mit_semseg/perfdb/gfx950_h2.json) on an MI355X:
#   gpurun --timeout 1500 -- 'bash tools/make_perfdb.sh'     then   cp gpurun_out/perfdb/gfx950_h2.json semantic-segmentation-pytorch_amd/mit_semseg/perfdb/
# Every BASELINE configuration is run once with the database OFF and an empty read-write cache: every geometry is timed by
# mit_semseg/tuner.py on the box (tile x split sweep + play-off), the cache collects the winners.
OUT=gpurun_out/perfdb; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
export SEMSEG_TUNE_DB=0 SEMSEG_TUNE_CACHE=$PWD/$OUT/plans.json
rm -f $SEMSEG_TUNE_CACHE
for c in 1 2 4 3; do
  timeout 900 python bench.py --config $c --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs --no-box --no-scaling-model --repeats 0 \
      > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
  echo "cfg$c rc=$? $(python -c "import json,sys; d=json.loads([l for l in open('$OUT/bench_cfg$c.json') if l.startswith('{')][-1]); print(d['ms_per_step'],'ms')" 2>/dev/null)"
done
timeout 300 python tools/bench_infer.py > $OUT/infer.log 2>&1; echo "infer rc=$?"
# the geometries of the parity suite as well: with every plan of a test process coming from the database, a test launches the SAME
# plans on every box and in every run -- the summation order of a split reduction, and with it which way a knife-edge ReLU gate
# resolves, no longer depends on what the tuner happened to time in that process
timeout 2000 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_tuning.log 2>&1; echo "pytest (tuning pass) rc=$?"; tail -3 $OUT/pytest_tuning.log | cut -c1-200
python - <<'PY'
import json, os, subprocess
out = os.path.join('gpurun_out', 'perfdb')
d = json.load(open(os.path.join(out, 'plans.json')))
head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
db = {'_about': 'launch plans measured by mit_semseg/tuner.py on one MI355X (tools/make_perfdb.sh): key = scheme,pass,geometry; value = '
                '[tile, split, ms]; pass 0 forward, 1 data gradient, 2 weight gradient, 3 batched Winograd GEMM tile, 4 choice between '
                'launch forms (tuner.choose).  %d entries.' % len(d)}
db.update(dict(sorted(d.items())))
json.dump(db, open(os.path.join(out, 'gfx950_h2.json'), 'w'), indent=0)
print('entries', len(d))
PY
