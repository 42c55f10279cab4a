#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py workload; prints the pv_* kernels: tools/trace.sh <outdir> [bench args...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 --repeats 1 "$@" > $OUT/log.txt 2>&1
python - "$OUT" <<'PY' | tee -a $OUT/trace.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pv_" in r["Name"]:
            print("%-90s calls %5s  avg %10.1f us  min %10.1f  max %10.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
