# SQ counters of the split-f16 GEMM alone (development aid): tools/gemm_pmc.sh <tag> [M N K]
set -e
R=$PWD; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/gpmc_$TAG/p$i -o p$i --output-format csv -- python3 $R/tools/diag/gemm_pmc.py "$@" > $R/gpurun_out/gpmc_${TAG}_p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python - <<PY
import csv,glob,collections
tot=collections.defaultdict(list); dur=[]
for f in glob.glob("gpurun_out/gpmc_$TAG/**/*counter_collection.csv",recursive=True):
    per=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "k_gemm_nt_h" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]]+=float(r["Counter_Value"])
    for d in sorted(per)[-4:]:
        for k,v in per[d].items(): tot[k].append(v)
for f in glob.glob("gpurun_out/gpmc_$TAG/**/*kernel_trace.csv",recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if "k_gemm_nt_h" in r["Kernel_Name"]]
    dur += [(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows[-4:]]
m={k:sum(v)/len(v) for k,v in tot.items()}
print("$TAG  kernel us (under pmc):", ["%.1f"%d for d in dur])
for k in sorted(m): print("  %-28s %.5g" % (k, m[k]))
PY
rm -rf gpurun_out/gpmc_$TAG
