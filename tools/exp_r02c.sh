#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02c; mkdir -p $OUT
cd $ROOT
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline"
run() { n=$1; lib=$2; shift 2
  PHAZE_LIB=$lib $B "$@" > $OUT/$n.json 2> $OUT/$n.err
  python - <<PY
import json
try:
    j=json.loads(open("$OUT/$n.json").read().strip().splitlines()[-1])
    print("$n", "ms=%.4f"%j["roofline"]["kernel_ms"], "frames/s=%.4g"%j["value"], "frac=%.4f"%j["roofline"]["frac"], "parity=%.3g"%(j["parity_rms_vs_oracle"] or -1), "fpc", j["config"]["frames_per_chunk"], "lds", j["config"]["lds_bytes_per_workgroup"])
except Exception as e:
    print("$n FAILED", e, open("$OUT/$n.err").read()[-400:])
PY
}
E=$ROOT/build/exp
run new ""
run old $E/libphaze_old.so
run w4 $E/libphaze_w4.so
run w8 $E/libphaze_w8.so
run new2 ""
cd /tmp && export TMPDIR=/tmp
for v in w4 w8; do
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH" ; do
  i=$((i+1))
  PHAZE_LIB=$E/libphaze_$v.so timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_${v}_$i -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/pmc_${v}_$i.log 2>&1
done
python - <<PY
import csv,glob
acc={}
for f in glob.glob("$OUT/pmc_${v}_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "pv_wave" in r["Kernel_Name"]: acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print("== $v")
for k,v in sorted(acc.items()): print(k, "%.5g"%(sum(v)/len(v)), "per frame %.1f"%(sum(v)/len(v)/1066867))
PY
done
