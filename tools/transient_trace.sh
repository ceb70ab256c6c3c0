set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ttrace
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/ttrace -o t --output-format csv -- python3 $R/tools/transient_probe.py 300 > $R/gpurun_out/transient_2.txt 2>&1
T=$(find /tmp/ttrace -name '*kernel_trace.csv' | head -1)
python3 $R/tools/transient_trace.py $T > $R/gpurun_out/transient_trace.txt
