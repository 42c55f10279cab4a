cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pf in 1.5 0.8; do
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT --output-format csv -d $R/gpurun_out/pmc_pf$pf -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --pitch $pf > /dev/null 2>&1
python - <<PY
import csv,glob
acc={}
for f in glob.glob("$R/gpurun_out/pmc_pf$pf/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "pv_" in r["Kernel_Name"]: acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print("pf=$pf", {k: round(sum(v)/len(v)/1066867,1) for k,v in acc.items()})
PY
done
