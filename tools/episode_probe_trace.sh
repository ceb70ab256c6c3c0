set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
for M in acc none; do
D=$R/gpurun_out/prof_ep_$M; rm -rf $D
timeout -k 10 300 rocprofv3 --kernel-trace -d $D -o t --output-format csv -- python3 $R/tools/episode_probe_trace.py $M > $R/gpurun_out/ep_trace_$M.out 2>&1
T=$(find $D -name '*kernel_trace.csv' | head -1)
python3 $R/tools/step_timeline.py $T --steps 30 > $R/gpurun_out/ep_timeline_$M.txt
rm -rf $D
done
