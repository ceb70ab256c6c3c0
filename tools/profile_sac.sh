# SAC update at production size under rocprofv3 (kernel trace): per-kernel mean over the update loop.
set -e
R=$PWD; TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/time_sac.py 200 > $R/gpurun_out/${TAG}_sac_time.txt 2>&1
D=$R/gpurun_out/prof_${TAG}_sac
rm -rf $D
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 $R/tools/prof_sac.py > $R/gpurun_out/${TAG}_prof_sac.out 2> $R/gpurun_out/${TAG}_prof_sac.err
S=$(find $D -name '*kernel_stats.csv' | head -1)
T=$(find $D -name '*kernel_trace.csv' | head -1)
cp $S $R/gpurun_out/${TAG}_sac_kernel_stats.csv
python3 $R/tools/sac_timeline.py $T > $R/gpurun_out/${TAG}_sac_update_timeline.txt
rm -rf $D
