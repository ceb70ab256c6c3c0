# memory-path counters of the frame kernel alone (development aid): tools/fw_pmc_mem.sh <tag>
set -e
R=$PWD; TAG=$1
cd /tmp && export TMPDIR=/tmp
i=0
# (the TA_* / TCP_* / TCC_EA0_* groups tried here did not come back within 200 s per pass on this pool:
#  three passes, ten GPU-minutes, nothing collected -- only the SQ group is kept)
for C in "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmcm_$TAG/p$i -o p$i --output-format csv -- python3 $R/tools/fw_pmc.py 256 0 > $R/gpurun_out/pmcm_${TAG}_p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python - <<PY
import csv,glob,collections
tot=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcm_$TAG/**/*counter_collection.csv",recursive=True):
    per=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "k_frame_wave" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]]+=float(r["Counter_Value"])
    for d in sorted(per)[-3:]:
        for k,v in per[d].items(): tot[k].append(v)
m={k:sum(v)/len(v) for k,v in tot.items()}
for k in sorted(m): print("  %-40s %.5g" % (k, m[k]))
g=lambda k: m.get(k,0.0)
if g("TCP_TCC_READ_REQ_sum"): print("  L1->L2 read latency (cycles) %.0f" % (g("TCP_TCC_READ_REQ_LATENCY_sum")/g("TCP_TCC_READ_REQ_sum")))
if g("TCC_EA0_RDREQ_sum"): print("  L2->fabric read latency (cycles) %.0f ; 32B share %.2f" % (g("TCC_EA0_RDREQ_LEVEL_sum")/g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_RDREQ_32B_sum")/g("TCC_EA0_RDREQ_sum")))
if g("SQ_INSTS_VMEM_RD"): print("  VMEM read latency per instruction (cycles) %.0f" % (g("SQ_INST_LEVEL_VMEM")/max(1,g("SQ_INSTS_VMEM_RD"))))
PY
rm -rf gpurun_out/pmcm_$TAG
