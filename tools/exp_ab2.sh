#!/bin/bash
# A/B of experiment builds (build/exp/libphaze_<name>.so) against the product on ONE box: tools/exp_ab2.sh <outdir> <name> ... [-- bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift; mkdir -p $OUT
cd $ROOT
names=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done; [ "$1" == "--" ] && shift
B="python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-extras --allow-lib-override"
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
{
one base "" "$@"
for n in "${names[@]}"; do one $n $ROOT/build/exp/libphaze_$n.so "$@"; done
one base_again "" "$@"
} 2>&1 | tee -a $OUT/summary.txt
