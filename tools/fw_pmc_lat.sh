# usage: tools/fw_pmc_lat.sh <tag>   (AOMARL_LIB selects the library build): latency-side counters of the frame kernel alone
set -e
R=$PWD; TAG=$1
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LEVEL_WAVES" "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmcl_$TAG/p$i -o p$i --output-format csv -- python3 $R/tools/fw_pmc.py 256 0 > $R/gpurun_out/pmcl_${TAG}_p$i.log 2>&1
done
cd $R
python - <<PY
import csv,glob,collections
tot=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcl_$TAG/**/*counter_collection.csv",recursive=True):
    per=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "k_frame_wave" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]]+=float(r["Counter_Value"])
    for d in sorted(per)[-3:]:
        for k,v in per[d].items(): tot[k].append(v)
m={k:sum(v)/len(v) for k,v in tot.items()}
print("$TAG")
for k in sorted(m): print("  %-28s %.4g" % (k, m[k]))
def r(a,b): return m[a]/m[b] if a in m and b in m and m[b] else float('nan')
print("  vmem level/inst %.0f  lds level/inst %.0f  smem level/inst %.0f  ifetch level/fetch %.1f" % (r("SQ_INST_LEVEL_VMEM","SQ_INSTS_VMEM_RD"), r("SQ_INST_LEVEL_LDS","SQ_INSTS_LDS"), r("SQ_INST_LEVEL_SMEM","SQ_INSTS_SMEM"), r("SQ_IFETCH_LEVEL","SQ_IFETCH")))
PY
rm -rf gpurun_out/pmcl_$TAG
