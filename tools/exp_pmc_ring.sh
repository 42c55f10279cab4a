cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_ring -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --fft 2048 --hop 128 --channels 2 --hops 262144 --pitch 1.5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d $R/gpurun_out/pmc_ring2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --fft 2048 --hop 128 --channels 2 --hops 262144 --pitch 1.5 > /dev/null 2>&1
python - <<PY
import csv,glob
for d in ("pmc_ring","pmc_ring2"):
    acc={}
    for f in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv" % d,recursive=True):
        for r in csv.DictReader(open(f)):
            if "pv_" in r["Kernel_Name"]: acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    print(d, {k: round(sum(v)/len(v)/524288,1) for k,v in acc.items()}, "per output frame (2 waves each)")
PY
