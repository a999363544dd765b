#!/bin/bash
# Re-time the launch plans of the BENCH geometries (BASELINE configs[1..4] + the inference bench) after new tile forms were added,
# and merge them into the shipped performance database (the geometries of the parity suite keep their entries):
#   gpurun --timeout 1500 -- 'bash tools/retune_bench_geoms.sh'   then   cp gpurun_out/retune/gfx950_h2.json semantic-segmentation-pytorch_amd/mit_semseg/perfdb/
OUT=gpurun_out/retune; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
export SEMSEG_TUNE_DB=0 SEMSEG_TUNE_CACHE=$PWD/$OUT/plans.json
rm -f $SEMSEG_TUNE_CACHE
for c in ${RETUNE_CONFIGS:-1 2 4 3}; do      # RETUNE_CONFIGS="4": only that config's geometries are re-timed and merged
  timeout 900 python bench.py --config $c --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs --no-box --no-scaling-model --repeats 0 \
      > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
  echo "cfg$c rc=$? $(python -c "import json,sys; d=json.loads([l for l in open('$OUT/bench_cfg$c.json') if l.startswith('{')][-1]); print(d['ms_per_step'],'ms')" 2>/dev/null)"
done
[ -z "$RETUNE_CONFIGS" ] && { timeout 300 python tools/bench_infer.py > $OUT/infer.log 2>&1; echo "infer rc=$?"; }
python - <<'PY'
import json, os
out = os.path.join('gpurun_out', 'retune')
new = json.load(open(os.path.join(out, 'plans.json')))
path = os.path.join('semantic-segmentation-pytorch_amd', 'mit_semseg', 'perfdb', 'gfx950_h2.json')
db = json.load(open(path))
db.pop('_about', None)
changed = sum(1 for k, v in new.items() if db.get(k, [None, None])[:2] != v[:2])
db.update(new)
res = {'_about': 'launch plans measured by mit_semseg/tuner.py on one MI355X (tools/make_perfdb.sh; the bench geometries re-timed by '
                 'tools/retune_bench_geoms.sh): key = scheme,pass,geometry; value = [tile, split, ms]; pass 0 forward, 1 data gradient, '
                 '2 weight gradient, 3 batched Winograd GEMM tile, 4 choice between launch forms (tuner.choose).  %d entries.' % len(db)}
res.update(dict(sorted(db.items())))
json.dump(res, open(os.path.join(out, 'gfx950_h2.json'), 'w'), indent=0)
print('re-timed', len(new), 'entries;', changed, 'plans changed; database', len(db))
import collections
c = collections.Counter()
for k, v in new.items():
    p = k.split(',')[1]
    c[(p, v[0])] += 1
print(sorted(c.items()))
PY
