#!/bin/bash
# Round-6 evidence collection (run on the GPU box: gpurun -- bash profiles/run_profile_r06.sh <tag>):
#   1. rocprofv3 --kernel-trace --stats of the default bench command            -> gpurun_out/<tag>/trace
#   2. separate --pmc passes of the headline workload (never with sys/hip/hsa tracing) -> gpurun_out/<tag>/pmc_<i>
#   3. FETCH_SIZE / WRITE_SIZE passes for every BASELINE shape + a calibration copy     -> gpurun_out/<tag>/hbm_<shape>_<counter>
#   4. LDS / VALU / wait counters of the kernels of C3, C4, C5 and the native shape (pv_wave2k_kernel / pv_wg_kernel) -> gpurun_out/<tag>/wg_<shape>_<i>
# profiles/pmc_report.py turns the CSVs into the committed summaries.
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
HEAD="python $ROOT/bench.py --no-extras --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $HEAD --steps 40 --warmup 10 > "$OUT/trace.log" 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" \
           "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$i" -- $HEAD --steps 10 --warmup 3 > "$OUT/pmc_$i.log" 2>&1
done
# HBM traffic per shape (FETCH_SIZE and WRITE_SIZE need separate passes)
shape() { # name args...
  n=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/hbm_${n}_$c" -- $HEAD --steps 4 --warmup 2 "$@" > "$OUT/hbm_${n}_$c.log" 2>&1
  done
}
shape c2
shape c3 --fft 2048 --hop 512 --channels 2 --hops 262144 --pitch 0.8
shape c4 --fft 4096 --hop 1024 --channels 1024 --hops 64 --pitch 1.25
shape c5 --fft 8192 --hop 2048 --channels 8 --hops 16384 --pitch 1.5
shape native --fft 2048 --hop 128 --channels 2 --hops 262144 --pitch 1.0
# calibration: a device-to-device copy of 1 GiB (torch) under the same counters
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/hbm_calib_$c" -- python -c "
import torch
a=torch.empty(1<<28, dtype=torch.float32, device='cuda'); b=torch.empty_like(a)
for _ in range(4): b.copy_(a)
torch.cuda.synchronize()" > "$OUT/hbm_calib_$c.log" 2>&1
done
# workgroup kernel: LDS / VALU / wait counters per shape
wg() { n=$1; shift; j=0
  for grp in "SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
             "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" ; do
    j=$((j+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/wg_${n}_$j" -- $HEAD --steps 4 --warmup 2 "$@" > "$OUT/wg_${n}_$j.log" 2>&1
  done
}
wg c3 --fft 2048 --hop 512 --channels 2 --hops 262144 --pitch 0.8
wg c3f15 --fft 2048 --hop 512 --channels 2 --hops 262144 --pitch 1.5
wg c4 --fft 4096 --hop 1024 --channels 1024 --hops 64 --pitch 1.25
wg c5 --fft 8192 --hop 2048 --channels 8 --hops 16384 --pitch 1.5
wg c3f07 --fft 2048 --hop 512 --channels 2 --hops 262144 --pitch 0.7
wg native --fft 2048 --hop 128 --channels 2 --hops 262144 --pitch 1.0
wg c2f08 --pitch 0.8
wg c5f08 --fft 8192 --hop 2048 --channels 8 --hops 16384 --pitch 0.8
wg c5sweep --fft 8192 --hop 2048 --channels 8 --hops 16384 --pitch-sweep
# the calibrated pipe microbenchmark (tools/r03_pipe_microbench.hip): every row timed by s_memtime, s_memrealtime and HIP events
[ -x $ROOT/tools/_r04_issue_microbench ] && timeout 300 $ROOT/tools/_r04_issue_microbench > "$OUT/issue_microbench.txt" 2>&1
ls "$OUT" | wc -l
