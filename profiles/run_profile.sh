#!/bin/bash
# Collects the rocprofv3 evidence for one build of the kernel (run on the GPU box via gpurun):
#   1. --kernel-trace --stats  (per-kernel durations)            -> gpurun_out/<tag>_trace
#   2. separate --pmc passes   (HBM bytes, LDS, VALU, occupancy)  -> gpurun_out/<tag>_pmc_<group>
# PMC passes never combine with sys/hip/hsa trace domains (gpurun refuses that).
# usage: profiles/run_profile.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline $*"
# the trace pass runs 40 steps so that its mean is dominated by steady-state dispatches (the first 4-5 of a process run at ramping clocks)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline $* > "$OUT/${TAG}_trace.log" 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" \
           "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/${TAG}_pmc_$i" -- $BENCH > "$OUT/${TAG}_pmc_$i.log" 2>&1
  tail -2 "$OUT/${TAG}_pmc_$i.log"
done
ls "$OUT" | head -40
