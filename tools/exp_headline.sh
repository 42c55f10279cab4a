#!/bin/bash
# Round-3 measurement session for pv_wave_kernel_1024 on ONE gpurun box: calibrated pipe microbenchmark, ablation builds (timing only,
# results wrong by construction), phase clocks, and the rocprofv3 --att attempt.  Everything lands in gpurun_out/$1.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r03a}; mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --allow-lib-override"
one() { n=$1; lib=$2; shift 2
  PHAZE_LIB=$lib $B "$@" > $OUT/$n.json 2> $OUT/$n.err
  python - <<PY
import json
try:
    j=json.loads(open("$OUT/$n.json").read().strip().splitlines()[-1])
    print("%-16s"%"$n", "ms=%.4f"%j["roofline"]["kernel_ms"], "frac=%.4f"%j["roofline"]["frac"], "parity=%.3g"%(j["parity_rms_vs_oracle"] or -1))
except Exception as e:
    print("$n FAILED", e, open("$OUT/$n.err").read()[-400:])
PY
}
E=$ROOT/build/exp
{
one base ""
for lib in $E/libphaze_${VARIANTS:-abl}*.so; do [ -f "$lib" ] && one $(basename $lib .so | sed s/libphaze_//) $lib; done
one base_again ""
one base_f08 "" --pitch 0.8
for lib in $E/libphaze_stamps*.so; do [ -f "$lib" ] && one $(basename $lib .so | sed s/libphaze_//) $lib; done
} 2>&1 | tee $OUT/summary.txt
for lib in $E/libphaze_stamps*.so; do
  n=$(basename $lib .so | sed s/libphaze_//)
  PHAZE_LIB=$lib python tools/read_stamps.py 1.5 > $OUT/$n.phases.json 2> $OUT/$n.phases.txt
  PHAZE_LIB=$lib python tools/read_stamps.py 0.8 > $OUT/$n.phases_f08.json 2> $OUT/$n.phases_f08.txt
  cat $OUT/$n.phases.txt
done
timeout 300 tools/r03_pipe_microbench > $OUT/pipe_microbench.txt 2>&1; tail -50 $OUT/pipe_microbench.txt
# thread trace: does this image decode it?
( cd /tmp && timeout 300 rocprofv3 --att --att-target-cu 1 --kernel-include-regex pv_wave_kernel -d $OUT/att -- python $ROOT/bench.py --steps 1 --warmup 1 --hops 65536 --no-cpu-baseline --no-extras ) > $OUT/att.log 2>&1
echo "att rc=$?"; tail -15 $OUT/att.log; find $OUT/att -type f | head -20; du -sh $OUT/att
