#!/bin/bash
# usage (GPU box, repo root): bash tools/profile_config5.sh <tag>
# configs[4] (noisy 40x40 sensor + the shipped denoiser, 256 envs, all fp32) as the bench's main workload under
# rocprofv3 --kernel-trace --stats -> per-kernel stats, steady-state summary, one-step timeline; then a --pmc pass on the
# fp32 denoiser kernel.
set -e
R=$PWD; TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/prof_${TAG}_config5
rm -rf $D
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 $R/bench.py --config production_sh_40x40_8m_3layers_d0_noise --denoiser shipped --steps 30 --warmup 5 --settle 10 --no-side-configs --no-cpu-baseline --timed-only > $R/gpurun_out/${TAG}_bench_line_under_rocprofv3_config5.json 2> $R/gpurun_out/${TAG}_prof_config5.err
T=$(find $D -name '*kernel_trace.csv' | head -1)
S=$(find $D -name '*kernel_stats.csv' | head -1)
cp $S $R/gpurun_out/${TAG}_rocprofv3_kernel_stats_config5.csv
python3 $R/tools/summarize_trace.py $T --steps 29 --marker k_frame_wave --from-index -1 > $R/gpurun_out/${TAG}_steady_state_kernel_summary_config5.csv
python3 $R/tools/step_timeline.py $T --steps 20 --from-index -1 > $R/gpurun_out/${TAG}_step_timeline_config5.txt
grep '^{"metric"' $R/gpurun_out/${TAG}_bench_line_under_rocprofv3_config5.json | tail -1 > $R/gpurun_out/${TAG}_line.tmp && mv $R/gpurun_out/${TAG}_line.tmp $R/gpurun_out/${TAG}_bench_line_under_rocprofv3_config5.json
rm -rf $D
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/dpmc_$TAG/p$i -o p$i --output-format csv -- python3 $R/tools/time_denoise.py > $R/gpurun_out/dpmc_${TAG}_p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 - > gpurun_out/${TAG}_pmc_denoise_f32.txt <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/dpmc_$TAG/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); key = {}
    for r in csv.DictReader(open(f)):
        if "k_denoise" in r["Kernel_Name"]:
            d = int(r["Dispatch_Id"]); per[d][r["Counter_Name"]] += float(r["Counter_Value"]); key[d] = r["Kernel_Name"].split("(")[0]
    for d in per:
        for k, v in per[d].items(): tot[key[d]][k].append(v)
print("# denoiser kernels alone (tools/time_denoise.py under rocprofv3 --pmc), counters per launch")
for key in sorted(tot):
    m = {k: sum(v) / len(v) for k, v in tot[key].items()}
    print(key)
    if m.get("SQ_WAVES") and m.get("SQ_INSTS_MFMA"):
        print("    per wave: %.0f matrix, %.0f other vector, %.0f LDS instructions; waiting %.1f %% of its cycles" %
              (m["SQ_INSTS_MFMA"] / m["SQ_WAVES"], (m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / m["SQ_WAVES"], m.get("SQ_INSTS_LDS", 0) / m["SQ_WAVES"],
               100.0 * m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
    for k in sorted(m): print("    %-28s %.5g" % (k, m[k]))
PY
rm -rf gpurun_out/dpmc_$TAG
head -12 gpurun_out/${TAG}_steady_state_kernel_summary_config5.csv
