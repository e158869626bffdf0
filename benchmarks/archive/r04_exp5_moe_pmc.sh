#!/bin/bash
# PMC passes over fused_experts at a Mixtral prefill batch (benchmarks/r04_exp3_moe_kernels.py): what the 256 x 256 x 64 grouped
# GEMM waits for.  One counter set per pass, no trace domain mixed in.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
# (the counter sets of benchmarks/gpu_pmc_phases.sh: known to collect on this image; TCC_* / TA_* / TCP_* derived sets produced no
# counter file or aborted rocprofv3 here)
SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE")
i=0
for S in "${SETS[@]}"; do
  D=/tmp/moe_pmc$i; rm -rf $D
  (cd /tmp && timeout 90 rocprofv3 --pmc $S -f csv -d $D -o run -- python $REPO/benchmarks/r04_exp3_moe_kernels.py 7680 14336 > $REPO/gpurun_out/moe_pmc$i.log 2>&1)
  echo "set $i ($S): exit $?"
  python - "$D" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "moe_gemm256" if "moe_gemm256" in r["Kernel_Name"] else "moe_tiled_128" if "moe_tiled_gemm_kernel" in r["Kernel_Name"] else None
        if k is None:
            continue
        key = (k, r.get("Grid_Size"))
        a = acc[key][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for key, cs in acc.items():
    print("   ", key, {c: round(v[1] / v[0], 1) for c, v in cs.items()}, "dispatches", max(v[0] for v in cs.values()))
PY
  i=$((i+1))
done
