#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02e; mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
node tools/bench_latency_node.js 2000 > $OUT/latency_node.jsonl 2> $OUT/latency_node.err; python -c "
import json
for l in open('$OUT/latency_node.jsonl'): j=json.loads(l); print(j['config']['workload'][:50], 'p50 %.1f p99 %.1f max %.1f'%(j['p50'],j['p99'],j['max']))
" || tail -3 $OUT/latency_node.err
