# configs[1] under rocprofv3 --kernel-trace: one-step timeline (development aid)
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/prof_c1; rm -rf $D
timeout -k 10 300 rocprofv3 --kernel-trace -d $D -o t --output-format csv -- python3 $R/bench.py --config production_sh_10x10_2m --envs 64 --steps 300 --warmup 50 --timed-only --no-cpu-baseline --no-side-configs > $R/gpurun_out/c1_trace.out 2>&1
T=$(find $D -name '*kernel_trace.csv' | head -1)
python3 $R/tools/step_timeline.py $T --steps 100 --from-index -1 > $R/gpurun_out/c1_timeline.txt
rm -rf $D
