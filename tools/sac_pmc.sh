# SQ / TCC counters of the SAC update's kernels at production size (tools/prof_sac.py: 25 updates of 14 agents, batch
# 256) -- separate --pmc passes, kernel trace only:   bash tools/sac_pmc.sh <tag>   -> gpurun_out/<tag>_pmc_sac_update.txt
set -e
R=$PWD; TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/spmc_$TAG/p$i -o p$i --output-format csv -- python3 $R/tools/prof_sac.py > $R/gpurun_out/spmc_${TAG}_p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 - > gpurun_out/${TAG}_pmc_sac_update.txt <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/spmc_$TAG/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    key = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_sac" in n or "k_gemm_g" in n or "k_gemm_batched" in n:
            d = int(r["Dispatch_Id"])
            per[d][r["Counter_Name"]] += float(r["Counter_Value"])
            key[d] = (n.split("(")[0].replace("void ", ""), r.get("Grid_Size", ""))
    for d in per:
        for k, v in per[d].items():
            tot[key[d]][k].append(v)
print("# aomarl_sac_update, 14 agents x batch 256 (tools/prof_sac.py under rocprofv3 --pmc, three passes): counters per launch")
print("# averaged over the launches of a (kernel, grid); a grid of k_gemm_g_multi = one line of the update's sequence")
for key in sorted(tot):
    m = {k: sum(v) / len(v) for k, v in tot[key].items()}
    wg = int(key[1]) // 256 if key[1] else 0
    print("%s, %d workgroups (%d launches averaged)" % (key[0], wg, len(next(iter(tot[key].values())))))
    if m.get("SQ_INSTS_MFMA") and m.get("SQ_WAVES"):
        pw = m["SQ_INSTS_MFMA"] / m["SQ_WAVES"]
        busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(m.get("SQ_BUSY_CYCLES", 1), 1)
        print("    per wave: %.0f matrix + %.0f other vector + %.0f LDS + %.0f global-load instructions; waiting %.0f %% of its cycles; "
              "matrix pipe busy %.0f %% of the SQ-busy cycles; L2 hit rate %.2f" %
              (pw, (m.get("SQ_INSTS_VALU", 0) - m["SQ_INSTS_MFMA"]) / m["SQ_WAVES"], m.get("SQ_INSTS_LDS", 0) / m["SQ_WAVES"],
               m.get("SQ_INSTS_VMEM_RD", 0) / m["SQ_WAVES"], 100.0 * m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
               100.0 * busy / 4.0, m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1)))
    for k in sorted(m):
        print("    %-28s %.5g" % (k, m[k]))
PY
rm -rf gpurun_out/spmc_$TAG
head -30 gpurun_out/${TAG}_pmc_sac_update.txt
