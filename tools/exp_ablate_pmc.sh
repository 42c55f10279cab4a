#!/bin/bash
# Per-phase LDS / VALU accounting of the wave kernel: PMC passes of the AUX (ablation) build with cumulative PHAZE_ABLATE masks.
# usage (GPU box): tools/exp_ablate_pmc.sh [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ablate; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ab in ${ABLATE_LIST:-128 129 131 135 143 159}; do
  PHAZE_ABLATE=$ab timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/ab$ab -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/ab$ab.log 2>&1
  python - <<PY
import csv,glob
acc={}
for f in glob.glob("$OUT/ab$ab/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "pv_" in r["Kernel_Name"]: acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print("ablate=$ab", {k: round(sum(v)/len(v)/1066867,1) for k,v in acc.items()}, "(per computed frame)")
PY
done
