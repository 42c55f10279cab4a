#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02b; mkdir -p $OUT
cd $ROOT
./tools/valu_microbench > $OUT/valu_microbench.txt 2>&1
cat $OUT/valu_microbench.txt
( python bench.py --steps 4000 --warmup 5 --no-cpu-baseline > $OUT/long.json 2>/dev/null & ) ; sleep 7
for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2; rocm-smi --showpower 2>/dev/null | grep -i "power (W)" | head -1; sleep 1; done > $OUT/clocks.txt 2>&1
wait; sleep 6; cat $OUT/clocks.txt; python -c "
import json; j=json.loads(open('$OUT/long.json').read().strip().splitlines()[-1]); print('long', j['ms_per_step'])"
