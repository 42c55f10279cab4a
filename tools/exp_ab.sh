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
E=$ROOT/build/exp
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
run new ""
run old $E/libphaze_old.so
run new2 ""
run new_pf08 "" --pitch 0.8
run old_pf08 $E/libphaze_old.so --pitch 0.8
run new_h128 "" --hop 128 --hops 524288
run old_h128 $E/libphaze_old.so --hop 128 --hops 524288
run new_h512 "" --hop 512 --hops 524288
run old_h512 $E/libphaze_old.so --hop 512 --hops 524288
run new_8ch "" --channels 8 --hops 131072
python bench.py --steps 5 --warmup 2 > $OUT/full.json 2> $OUT/full.err; python -c "
import json; j=json.loads(open('$OUT/full.json').read().strip().splitlines()[-1])
print('FULL', j['value'], j['roofline']['frac'], j['dtype'])
for c in j.get('configs',[]): print('  ', c['workload'][:60], '%.4g'%c['value'], '%.4f'%c['roofline_frac'], c['parity_rms_vs_oracle'], c['kernel'])
print('  latency', j.get('latency_us')); print('  cpu', j.get('cpu_baseline'))
" || tail -5 $OUT/full.err
