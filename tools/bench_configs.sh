#!/bin/bash
# The other BASELINE shapes (profiles/*_other_configs.md): one bench.py line each, condensed.  usage (GPU box): tools/bench_configs.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras "$@" | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('| %-58s | %.3g | %.2f | %.2f %% | %s | parity %.2g' % ('$*', d['value'], d['ms_per_step'], 100*d['roofline']['frac'], d['roofline'].get('kernel','?'), d['parity_rms_vs_oracle']))"
}
run --fft 1024 --hop 256 --channels 1 --hops 1048576 --pitch 1.5
run --fft 1024 --hop 256 --channels 1 --hops 1048576 --pitch 0.8
run --fft 2048 --hop 512 --channels 2 --hops 262144 --pitch 0.8
run --fft 2048 --hop 512 --channels 2 --hops 262144 --pitch 1.5
run --fft 4096 --hop 1024 --channels 64 --hops 4096 --pitch 1.25
run --fft 8192 --hop 2048 --channels 8 --hops 16384 --pitch 1.25
run --fft 8192 --hop 2048 --channels 8 --hops 16384 --pitch 0.8
run --fft 2048 --hop 128 --channels 2 --hops 524288 --pitch 1.5
run --fft 2048 --hop 128 --channels 2 --hops 524288 --pitch 0.8
run --fft 1024 --hop 128 --channels 1 --hops 1048576 --pitch 1.5
run --fft 1024 --hop 512 --channels 1 --hops 1048576 --pitch 1.5
