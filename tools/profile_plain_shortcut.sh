# The control / agent chain with the residual shortcut ALONE behind the frame kernel (plain call order) under rocprofv3:
#   bash tools/profile_plain_shortcut.sh <tag>  -> gpurun_out/<tag>_step_timeline_f32_plain_order_shortcut.txt
set -e
R=$PWD; TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/prof_${TAG}_ps; rm -rf $D
timeout -k 10 400 rocprofv3 --kernel-trace -d $D -o t --output-format csv -- python3 $R/bench.py --steps 60 --warmup 10 --no-side-configs --no-cpu-baseline --timed-only --no-frame-pipeline --residual-shortcut > $R/gpurun_out/${TAG}_ps.out 2>&1
T=$(find $D -name '*kernel_trace.csv' | head -1)
python3 $R/tools/step_timeline.py $T --steps 40 --from-index -1 > $R/gpurun_out/${TAG}_step_timeline_f32_plain_order_shortcut.txt
rm -rf $D
