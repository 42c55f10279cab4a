#!/bin/bash
# round 5, session 1 (one gpurun box): counters + phase clock of the fp32-first headline kernel, then a priority-table sweep.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=gpurun_out/${1:-r05b}; mkdir -p $OUT
bash tools/pmc_ab.sh ${1:-r05b}/pmc "" -- > /dev/null 2>&1
cat $OUT/pmc/pmc_ab.txt
for n in stamps1 stamps2; do
  PHAZE_LIB=$ROOT/build/exp/libphaze_$n.so python tools/read_stamps.py 1.5 > $OUT/$n.phases.json 2> $OUT/$n.phases.txt
  python - <<PY
import json
j=json.load(open("$OUT/$n.phases.json"))
print("$n", j["ticks_per_frame_per_wave_mean"], j["ticks_per_frame_per_wave_p5_p95"])
for p in j["phases"]: print("  %-70s %8.0f %5.1f %%" % (p["phase"], p["ticks"], 100*p["share"]))
PY
done 2>&1 | grep -v amdgpu.ids | tee $OUT/phases.txt
names=$(ls build/exp/libphaze_x*.so | sed 's/.*libphaze_//; s/\.so//')
bash tools/ab.sh ${1:-r05b} head 10 -- $names
