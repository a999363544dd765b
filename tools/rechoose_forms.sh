#!/bin/bash
# Re-time the launch-FORM choices (tuner.choose, plan pass 4: e.g. which form of the fused Winograd data gradient runs) of the shipped
# performance database after new forms were added to the library, without re-timing the ~1900 tile x split plans:
#   gpurun --timeout 900 -- 'bash tools/rechoose_forms.sh'   then   cp gpurun_out/rechoose/gfx950_h2.json semantic-segmentation-pytorch_amd/mit_semseg/perfdb/
OUT=gpurun_out/rechoose; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
export SEMSEG_TUNE_RECHOOSE=1 SEMSEG_TUNE_CACHE=$PWD/$OUT/plans.json
rm -f $SEMSEG_TUNE_CACHE
for c in 1 2 3; do
  timeout 600 python bench.py --config $c --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --no-box --no-scaling-model --repeats 0 \
      > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
  echo "cfg$c rc=$?"
done
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "full_size or winograd_dgrad" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
python - <<'PY'
import json, os
out = os.path.join('gpurun_out', 'rechoose')
new = json.load(open(os.path.join(out, 'plans.json')))
path = os.path.join('semantic-segmentation-pytorch_amd', 'mit_semseg', 'perfdb', 'gfx950_h2.json')
db = json.load(open(path))
n = 0
for k, v in new.items():
    if k.startswith('h2,4,'):
        if db.get(k) != v:
            print(k, db.get(k), '->', v)
        db[k] = v
        n += 1
about = db.pop('_about')
out_db = {'_about': about}
out_db.update(dict(sorted(db.items())))
json.dump(out_db, open(os.path.join(out, 'gfx950_h2.json'), 'w'), indent=0)
print('form choices re-timed:', n)
PY
