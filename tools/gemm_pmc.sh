# SQ counters of the balanced fp32 GEMM (k_gemm_p, csrc/aomarl_gemm_p.h) alone on the loop's shapes (development
# aid; separate --pmc passes, kernel trace only):   bash tools/gemm_pmc.sh <tag>      -> gpurun_out/<tag>_pmc_gemm_p.txt
# The program after `--` is the native micro-benchmark itself (tools/bin/gemmbench pick).
set -e
R=$PWD; TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/gpmc_$TAG/p$i -o p$i --output-format csv -- $R/tools/bin/gemmbench pick > $R/gpurun_out/gpmc_${TAG}_p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 - > gpurun_out/${TAG}_pmc_gemm_p.txt <<PY
import csv, glob, collections, re
# launches of one kernel instantiation with one grid size; per shape: the benchmark's own timing (pass 1's log)
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/gpmc_$TAG/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    key = {}
    for r in csv.DictReader(open(f)):
        if "k_gemm_p" in r["Kernel_Name"]:
            d = int(r["Dispatch_Id"])
            per[d][r["Counter_Name"]] += float(r["Counter_Value"])
            key[d] = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Grid_Size", ""))
    for d in per:
        for k, v in per[d].items():
            tot[key[d]][k].append(v)
print("# k_gemm_p alone (tools/bin/gemmbench pick under rocprofv3 --pmc, four passes), counters per launch averaged over the")
print("# launches of a (tile, grid); the benchmark's own timings of that run (under the counters: slower than free-running):")
for l in open("gpurun_out/gpmc_${TAG}_p1.log"):
    if l.startswith("==") or " pick " in l:
        print("#   " + l.rstrip())
for key in sorted(tot):
    m = {k: sum(v) / len(v) for k, v in tot[key].items()}
    wg = int(key[1]) // 256 if key[1] else 0
    print("%s, %d workgroups (%d launches averaged)" % (key[0], wg, len(next(iter(tot[key].values())))))
    if m.get("SQ_INSTS_MFMA") and m.get("SQ_WAVES"):
        per_wave = m["SQ_INSTS_MFMA"] / m["SQ_WAVES"]
        print("    per wave: %.0f matrix instructions (= %.1f us of one SIMD's matrix pipe at 32 cycles each, 2.4 GHz), %.0f other vector, "
              "%.0f LDS, %.0f global-load instructions; %.1f %% of a wave's cycles spent waiting" %
              (per_wave, per_wave * 32 / 2400.0, (m.get("SQ_INSTS_VALU", 0) - m["SQ_INSTS_MFMA"]) / m["SQ_WAVES"],
               m.get("SQ_INSTS_LDS", 0) / m["SQ_WAVES"], m.get("SQ_INSTS_VMEM_RD", 0) / m["SQ_WAVES"],
               100.0 * m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
        if m.get("SQ_LDS_IDX_ACTIVE"):
            print("    LDS: bank-conflict cycles %.0f %% of its active cycles; L2: hit rate %.2f" %
                  (100.0 * m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"],
                   m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1)))
    for k in sorted(m):
        print("    %-28s %.5g" % (k, m[k]))
PY
rm -rf gpurun_out/gpmc_$TAG
cat gpurun_out/${TAG}_pmc_gemm_p.txt | head -60
