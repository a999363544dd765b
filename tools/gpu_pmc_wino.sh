#!/bin/bash
# PMC passes (separate runs, as MI355X_MICROARCH.md prescribes) on the Winograd-domain data gradient of conv_last (the dominant pass of
# the configs[1] step):   gpurun --timeout 900 -- 'bash tools/gpu_pmc_wino.sh <tag>'
TAG=${1:-pmcw}; EXTRA="${2:-}"; ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
cd /tmp
CMD="python $ROOT/tools/probes/winograd_dgrad_pass.py --iters 5 ${EXTRA//,/ }"     # e.g. --mode,fwd,--geom,2:64:64:512:512:4 (":" -> ",")
CMD=${CMD//:/,}
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o $n -- $CMD > $OUT/$n.log 2>&1
  echo "pmc pass $n rc=$?"; }
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $ROOT
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; grep -A14 "igemm_dma_kernel\|wino_" $OUT/summary.txt | head -80
find $OUT -name '*.csv' -size +8M -delete
