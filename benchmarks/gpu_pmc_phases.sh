#!/bin/bash
# rocprofv3 --pmc passes over the bench job (benchmarks/pmc_workload.py), one counter set per pass and no trace
# domain mixed in; merged into gpurun_out/${PMC_NAME:-r06_pmc}.json (copy to profiles/ to commit).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
NAME=${PMC_NAME:-r06_pmc}
STEPS=${DECODE_STEPS:-2}
SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY")
DIRS=""
i=0
for S in "${SETS[@]}"; do
  D=/tmp/pmc_set$i
  rm -rf $D
  (cd /tmp && timeout 420 rocprofv3 --pmc $S -f csv -d $D -o run -- python $REPO/benchmarks/pmc_workload.py --decode-steps $STEPS ${PMC_ARGS:-} > $REPO/gpurun_out/pmc_set$i.log 2>&1)
  echo "set $i ($S): exit $? $(tail -1 $REPO/gpurun_out/pmc_set$i.log | cut -c1-160)"
  DIRS="$DIRS $D"
  i=$((i+1))
done
python benchmarks/summarize_pmc_phases.py gpurun_out/$NAME.json llama-3-8b 64 $STEPS $DIRS > gpurun_out/${NAME}_summary.txt 2>&1
head -60 gpurun_out/${NAME}_summary.txt
