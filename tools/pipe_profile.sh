#!/bin/bash
# usage (GPU box, repo root): bash tools/pipe_profile.sh <tag> [extra bench args]   -- kernel trace of the pipelined headline pass
set -e
R=$PWD; TAG=${1:-pipe}; shift || true
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/prof_$TAG
rm -rf $D
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 $R/bench.py --steps 60 --warmup 10 --no-side-configs --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_line.json 2> $R/gpurun_out/${TAG}_prof.err
T=$(find $D -name '*kernel_trace.csv' | head -1)
python3 $R/tools/step_timeline.py $T --steps 40 --from-index -1 > $R/gpurun_out/${TAG}_step_timeline.txt
python3 $R/tools/summarize_trace.py $T --steps 59 --marker k_frame_wave --from-index -1 > $R/gpurun_out/${TAG}_steady.csv
python3 $R/tools/pipe_gaps.py $T > $R/gpurun_out/${TAG}_pipe_gaps.txt
rm -rf $D
cat $R/gpurun_out/${TAG}_step_timeline.txt
