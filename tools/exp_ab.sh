#!/bin/bash
# A/B of library builds through PHAZE_LIB on one box: usage tools/exp_ab.sh <outdir> <name=lib|-> ... -- <bench args per line file>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift; mkdir -p $OUT
cd $ROOT
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras"
run() { n=$1; lib=$2; shift 2
  PHAZE_LIB=$lib $B --allow-lib-override "$@" > $OUT/$n.json 2> $OUT/$n.err
  python - <<PY
import json
try:
    j=json.loads(open("$OUT/$n.json").read().strip().splitlines()[-1])
    print("%-14s"%"$n", "ms=%.4f"%j["roofline"]["kernel_ms"], "frames/s=%.4g"%j["value"], "frac=%.4f"%j["roofline"]["frac"], "parity=%.3g"%(j["parity_rms_vs_oracle"] or -1), "fpc", j["config"]["frames_per_chunk"])
except Exception as e:
    print("$n FAILED", e, open("$OUT/$n.err").read()[-600:])
PY
}
# example body (edit per experiment): builds live under build/exp/ (git-ignored, shipped by gpurun)
E=$ROOT/build/exp
run new ""
for lib in $E/libphaze_*.so; do [ -f "$lib" ] && run $(basename $lib .so) $lib; done
run new_again ""
