#!/bin/bash
# One GPU-box call that checks everything the driver will run: pytest -m gpu, the default bench line (with the per-config lines, latency and
# cpu_baseline), the --gpus degradation and a simulated time-shard span.  usage: gpurun -- bash tools/gpu_check.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/check; mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; j=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('HEAD', '%.4g'%j['value'], 'frac %.4f'%j['roofline']['frac'], 'kernel_ms %.4f'%j['roofline']['kernel_ms'], j['dtype'], j['config']['workload'][:70])
for c in j.get('configs',[]): print('  ', c['workload'][:50], '%.4g'%c['value'], 'frac %.4f'%c['roofline_frac'], 'parity %.2g'%c['parity_rms_vs_oracle'], c['kernel'])
print('  latency', {k:j['latency_us'][k] for k in ('p50','p99','max')}); cb=j['cpu_baseline']; print('  cpu', cb['value'], cb.get('reference_ratio'), cb.get('reference_frames_per_s_est'), cb['cpu_model'])
" || tail -5 $OUT/bench.err
python bench.py --gpus 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GPUS2', j['n_gpus'], j.get('requested_gpus'), j.get('replicas_measured'))"
python bench.py --time-shard --simulate-shard 2/4 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TSHARD', j['span_starts_bit_exact_vs_processed_lead_in'], j['config']['rank0_span'], j['roofline']['kernel_ms'])"
# the RCCL branch of bench.py (process group, barrier, all_reduce(MAX)) at world size 1, launched the way the driver launches N ranks
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TORCHRUN1', j['n_gpus'], '%.4g'%j['value'], j['config']['parallelism'])"
