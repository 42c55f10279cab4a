#!/bin/bash
# round-2 experiment batch A (GPU box): A/B of kernel builds through PHAZE_LIB, occupancy sweep, PMC wait counters
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02a; mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
run() { # name lib args...
  n=$1; lib=$2; shift 2
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
run new2 ""
run w8 $E/libphaze_w8.so
run w15 $E/libphaze_w15.so
run tp64 $E/libphaze_tp64.so
run new_pf08 "" --pitch 0.8
run old_pf08 $E/libphaze_old.so --pitch 0.8
run new_2048 "" --fft 2048 --hop 512 --channels 2 --hops 262144 --pitch 0.8
run old_2048 $E/libphaze_old.so --fft 2048 --hop 512 --channels 2 --hops 262144 --pitch 0.8
run new_4096 "" --fft 4096 --hop 1024 --channels 1024 --hops 64 --pitch 1.25
run old_4096 $E/libphaze_old.so --fft 4096 --hop 1024 --channels 1024 --hops 64 --pitch 1.25
run new_8192 "" --fft 8192 --hop 2048 --channels 8 --hops 16384 --pitch 1.5
run old_8192 $E/libphaze_old.so --fft 8192 --hop 2048 --channels 8 --hops 16384 --pitch 1.5
run new_native "" --fft 2048 --hop 128 --channels 2 --hops 262144 --pitch 1.5
run old_native $E/libphaze_old.so --fft 2048 --hop 128 --channels 2 --hops 262144 --pitch 1.5
# clocks under load
( python bench.py --steps 3000 --warmup 5 --no-cpu-baseline > $OUT/long.json 2>/dev/null & ) ; sleep 14
for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -4; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -2; sleep 1; done > $OUT/clocks.txt 2>&1
wait; sleep 3; tail -12 $OUT/clocks.txt
# PMC: wait / activity counters of the new build
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$i -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/pmc_$i.log 2>&1
done
python - <<PY
import csv,glob
acc={}
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "pv_wave" in r["Kernel_Name"]: acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k, "%.5g"%(sum(v)/len(v)), "per frame %.1f"%(sum(v)/len(v)/1066867))
PY
