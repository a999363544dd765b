OUT=gpurun_out/r6d; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail $OUT/build.log; exit 1; }
b() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 40 --warmup 6 --no-cpu-baseline --no-other-configs --no-box --no-scaling-model --repeats 1 > $OUT/bench_$name.json 2> $OUT/bench_$name.err; python - $OUT/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], d['ms_per_step'],'ms', d['config']['repeat_windows']['ms_per_step'])
except Exception as e: print(sys.argv[2],'no line',e)
PY
}
timeout 900 python -m pytest tests -m gpu -q -x -k "deferred_fork_sums or native_gradients_vs_reference_anchor or (matches_reference_golden and (r50d_ppmds_64_train or hrnetv2_c1_128 or r18d_ppmds_64_train or r50_upernet))" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log | cut -c1-300
b f0a 1 SEMSEG_DEFER_FORK_SUMS=0; b f1a 1 SEMSEG_DEFER_FORK_SUMS=1; b f0b 1 SEMSEG_DEFER_FORK_SUMS=0; b f1b 1 SEMSEG_DEFER_FORK_SUMS=1
b w0 1 SEMSEG_WINO_WGRAD_FORM=0; b w1 1 SEMSEG_WINO_WGRAD_FORM=1; b w2 1 SEMSEG_WINO_WGRAD_FORM=2; b w0b 1 SEMSEG_WINO_WGRAD_FORM=0; b w1b 1 SEMSEG_WINO_WGRAD_FORM=1; b w2b 1 SEMSEG_WINO_WGRAD_FORM=2
b s0 1 SEMSEG_DMA64_SPREAD=0; b s1 1 SEMSEG_DMA64_SPREAD=1
b h0 4 SEMSEG_DEFER_FORK_SUMS=0; b h1 4 SEMSEG_DEFER_FORK_SUMS=1; b h0b 4 SEMSEG_DEFER_FORK_SUMS=0; b h1b 4 SEMSEG_DEFER_FORK_SUMS=1
