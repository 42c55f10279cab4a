#!/bin/bash
# Samples rocm-smi (socket power, sclk) while the headline launch loops for a few seconds: is the kernel's clock set by the power budget?
# usage: tools/power_probe.sh <outfile> [lib]   (lib: an A/B build under build/exp, default = the product)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$1; LIB=${2:-}
( for i in $(seq 1 14); do rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import json,sys
try:
    j=json.load(sys.stdin); c=j[sorted(j)[0]]
    print({k:v for k,v in c.items() if 'ower' in k or 'sclk' in k or 'mclk' in k})
except Exception as e: print('smi?', e)
"; sleep 0.5; done ) > $OUT.smi 2>&1 &
PHAZE_LIB=$LIB python bench.py --steps 2500 --warmup 20 --no-cpu-baseline --no-extras --allow-lib-override > $OUT.json 2>$OUT.err
wait
python3 -c "
import json; j=json.loads(open('$OUT.json').read().strip().splitlines()[-1]); print('kernel_ms', j['roofline']['kernel_ms'])"
cat $OUT.smi
