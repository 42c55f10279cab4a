#!/bin/bash
# Counters of the dominant kernel of one bench.py workload for the product and for experiment builds, one box (separate --pmc passes, never with tracing
# domains other than --kernel-trace):   tools/pmc_ab.sh <outdir> "<bench args>" -- name ...      ("product" is always measured first)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$1; ARGS=$2; shift 2; [ "$1" == "--" ] && shift
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM"
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE")
for n in product "$@"; do
  lib=""; [ "$n" != "product" ] && lib=$ROOT/build/exp/libphaze_$n.so
  i=0
  for grp in "${GROUPS_[@]}"; do
    i=$((i+1))
    PHAZE_LIB=$lib timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${n}_$i -- python $ROOT/bench.py --no-extras --no-cpu-baseline --allow-lib-override --steps 4 --warmup 2 --repeats 1 $ARGS > $OUT/${n}_$i.log 2>&1
  done
done
python - "$OUT" product "$@" <<'PY' | tee $OUT/pmc_ab.txt
import csv, glob, sys, os
out, names = sys.argv[1], sys.argv[2:]
tab = {}
for n in names:
    per = {}
    for f in glob.glob(os.path.join(out, n + "_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "pv_" in r["Kernel_Name"]:
                per.setdefault(r["Kernel_Name"][:40], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if not per: continue
    k = max(per, key=lambda q: sum(sum(v) for v in per[q].values()))
    tab[n] = {c: sum(v) / len(v) for c, v in per[k].items()}
cs = sorted({c for t in tab.values() for c in t})
print("%-24s" % "counter" + "".join("%16s" % n for n in names) + "   ratio(last/first)")
for c in cs:
    vals = [tab.get(n, {}).get(c, float("nan")) for n in names]
    print("%-24s" % c + "".join("%16.4g" % v for v in vals) + ("   %.3f" % (vals[-1] / vals[0]) if vals[0] else ""))
PY
