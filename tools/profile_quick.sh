set -e
R=$PWD; TAG=${1:-r04a}
cd /tmp && export TMPDIR=/tmp
for MODE in f32 f32_plain_order; do
  D=$R/gpurun_out/prof_${TAG}_$MODE
  rm -rf $D
  EXTRA="--frame-pipeline-always"; PREC=f32
  if [ $MODE = f32_plain_order ]; then EXTRA="--no-frame-pipeline"; fi
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 $R/bench.py --steps 60 --warmup 10 --no-side-configs --no-cpu-baseline --timed-only $EXTRA --precision $PREC > $R/gpurun_out/${TAG}_bench_line_under_rocprofv3_$MODE.json 2> $R/gpurun_out/${TAG}_prof_$MODE.err
  T=$(find $D -name '*kernel_trace.csv' | head -1)
  S=$(find $D -name '*kernel_stats.csv' | head -1)
  cp $S $R/gpurun_out/${TAG}_rocprofv3_kernel_stats_$MODE.csv
  python3 $R/tools/summarize_trace.py $T --steps 59 --marker k_frame_wave --from-index -1 > $R/gpurun_out/${TAG}_steady_state_kernel_summary_$MODE.csv
  python3 $R/tools/step_timeline.py $T --steps 40 --from-index -1 > $R/gpurun_out/${TAG}_step_timeline_$MODE.txt
  rm -rf $D
done
