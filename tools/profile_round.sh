#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the headline bench pass (all-fp32, frame pipeline; --timed-only: warm-up and
#    timed region only) -> per-kernel stats, steady-state summary, one-step timeline, idle time on the frame kernel's
#    queue, what each frame kernel waited for;  2. the same for the fast mode and for the plain call order;
# 3. rocprofv3 --pmc passes on the frame kernel alone, both arithmetics (tools/fw_pmc.sh).
# Everything lands under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -e
R=$PWD; TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
for MODE in f32 split_f16 f32_plain_order; do
  D=$R/gpurun_out/prof_${TAG}_$MODE
  rm -rf $D
  EXTRA="--frame-pipeline-always"; PREC=$MODE
  if [ $MODE = f32_plain_order ]; then EXTRA="--no-frame-pipeline"; PREC=f32; fi
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 $R/bench.py --steps 60 --warmup 10 --no-side-configs --no-cpu-baseline --timed-only $EXTRA --precision $PREC > $R/gpurun_out/${TAG}_bench_line_under_rocprofv3_$MODE.json 2> $R/gpurun_out/${TAG}_prof_$MODE.err
  T=$(find $D -name '*kernel_trace.csv' | head -1)
  S=$(find $D -name '*kernel_stats.csv' | head -1)
  cp $S $R/gpurun_out/${TAG}_rocprofv3_kernel_stats_$MODE.csv
  python3 $R/tools/summarize_trace.py $T --steps 59 --marker k_frame_wave --from-index -1 > $R/gpurun_out/${TAG}_steady_state_kernel_summary_$MODE.csv
  python3 $R/tools/step_timeline.py $T --steps 40 --from-index -1 > $R/gpurun_out/${TAG}_step_timeline_$MODE.txt
  grep '^{"metric"' $R/gpurun_out/${TAG}_bench_line_under_rocprofv3_$MODE.json | tail -1 > $R/gpurun_out/${TAG}_line.tmp && mv $R/gpurun_out/${TAG}_line.tmp $R/gpurun_out/${TAG}_bench_line_under_rocprofv3_$MODE.json
  K=$(python3 -c "import json;print(json.load(open('$R/gpurun_out/${TAG}_bench_line_under_rocprofv3_$MODE.json'))['roofline']['kernel'])")
  python3 $R/tools/queue_gaps.py $D "$K" > $R/gpurun_out/${TAG}_main_queue_gaps_$MODE.txt
  python3 $R/tools/pipe_gaps.py $T > $R/gpurun_out/${TAG}_frame_pipeline_waits_$MODE.txt
  python3 $R/tools/queue_busy.py $T 40 > $R/gpurun_out/${TAG}_queue_busy_$MODE.txt
  rm -rf $D
done
cd $R
bash tools/fw_pmc.sh ${TAG}f32 > gpurun_out/${TAG}_pmc_frame_kernel_summary_f32.txt 2>&1
AOMARL_PRECISION=split_f16 bash tools/fw_pmc.sh ${TAG}split > gpurun_out/${TAG}_pmc_frame_kernel_summary_split_f16.txt 2>&1
python3 tools/merge_pmc.py gpurun_out/${TAG}_pmc_frame_kernel.json gpurun_out/${TAG}f32_pmc_frame_kernel.json gpurun_out/${TAG}split_pmc_frame_kernel.json
tail -12 gpurun_out/${TAG}_pmc_frame_kernel_summary_f32.txt
head -30 gpurun_out/${TAG}_step_timeline_f32.txt
