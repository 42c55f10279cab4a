#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes; FETCH_SIZE x2 = the gfx950 wide-read correction of MI355X_MICROARCH.md) of one
# bench.py workload, product and experiment builds:  tools/hbm_ab.sh <outdir> "<bench args>" -- name ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$1; ARGS=$2; shift 2; rm -rf $OUT/*_FETCH_SIZE $OUT/*_WRITE_SIZE; [ "$1" == "--" ] && shift
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for n in product "$@"; do
  lib=""; [ "$n" != "product" ] && lib=$ROOT/build/exp/libphaze_$n.so
  for c in FETCH_SIZE WRITE_SIZE; do
    PHAZE_LIB=$lib timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${n}_$c -- python $ROOT/bench.py --no-extras --no-cpu-baseline --allow-lib-override --steps 4 --warmup 2 --repeats 1 $ARGS > $OUT/${n}_$c.log 2>&1
  done
done
python - "$OUT" "$ARGS" product "$@" <<'PY' | tee -a $OUT/hbm_ab.txt
import csv, glob, sys, os, json
out, args, names = sys.argv[1], sys.argv[2], sys.argv[3:]
for n in names:
    v = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        per = {}
        for f in glob.glob(os.path.join(out, f"{n}_{c}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "pv_" in r["Kernel_Name"] and "classify" not in r["Kernel_Name"]:
                    per.setdefault(r["Kernel_Name"][:60], []).append(float(r["Counter_Value"]))
        k = max(per, key=lambda q: sum(per[q])) if per else None
        v[c] = (sum(per[k]) / len(per[k]) if k else float("nan"), k)
    try:
        j = json.loads([l for l in open(os.path.join(out, f"{n}_FETCH_SIZE.log")) if l.startswith("{")][-1])
        alg = j["roofline"]["algorithmic_bytes_per_launch"]
    except Exception:
        alg = float("nan")
    rd, wr = v["FETCH_SIZE"][0] * 1024 * 2, v["WRITE_SIZE"][0] * 1024          # counters are in KiB; reads x2 (gfx950)
    print(f"{n:12s} [{args}] kernel {v['FETCH_SIZE'][1]}: read {rd/1e9:.3f} GB  write {wr/1e9:.3f} GB  total {(rd+wr)/1e9:.3f} GB  algorithmic {alg/1e9:.3f} GB  ratio {(rd+wr)/alg:.3f}")
PY
