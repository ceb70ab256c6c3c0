#!/bin/bash
# usage (GPU box, repo root): bash tools/profile_config2.sh <tag>   -- configs[1] (10x10, 64 envs) three ways under rocprofv3:
# frame pipeline (multi-stream, host-bound), plain order, and ONE stream replayed as a linear HIP graph (GPU-chain-bound)
set -e
R=$PWD; TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
for MODE in pipelined plain linear_graph; do
  D=$R/gpurun_out/prof_c2_$MODE
  rm -rf $D
  EXTRA="--frame-pipeline-always"
  [ $MODE = plain ] && EXTRA="--no-frame-pipeline"
  [ $MODE = linear_graph ] && EXTRA="--no-prefetch --graph-step"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 $R/bench.py --config production_sh_10x10_2m --envs 64 --steps 300 --warmup 60 --no-side-configs --no-cpu-baseline --timed-only $EXTRA > $R/gpurun_out/${TAG}_config2_line_$MODE.json 2> $R/gpurun_out/${TAG}_config2_$MODE.err
  T=$(find $D -name '*kernel_trace.csv' | head -1)
  python3 $R/tools/step_timeline.py $T --steps 100 --from-index -1 > $R/gpurun_out/${TAG}_step_timeline_config2_$MODE.txt
  python3 - $T <<'PY' >> $R/gpurun_out/${TAG}_step_timeline_config2_$MODE.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fr = [i for i, r in enumerate(rows) if "k_frame_wave" in r["Kernel_Name"]]
ts = [int(rows[i]["Start_Timestamp"]) for i in fr]
k = min(range(len(fr) - 200), key=lambda i: ts[i + 200] - ts[i])      # the 200 consecutive steps that took the least time
a, b = fr[k], fr[k + 200]
span = ts[k + 200] - ts[k]
# union of the kernels' execution intervals in that window (any queue)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows[a:b])
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("# 200 consecutive steps under the profiler: %.1f us per step, a kernel executing during %.0f %% of that time (union over all queues)" % (span / 200 / 1e3, 100.0 * busy / span))
PY
  grep '^{"metric"' $R/gpurun_out/${TAG}_config2_line_$MODE.json | tail -1 | python3 -c "import json,sys; d=json.load(sys.stdin); print('# bench line of this run: %.0f steps/s, %.4f ms per step, host enqueue %.4f ms' % (d['value_no_reset'], d['ms_per_step_no_reset'], d['host_enqueue_ms_per_step']))" >> $R/gpurun_out/${TAG}_step_timeline_config2_$MODE.txt
  rm -rf $D
  tail -3 $R/gpurun_out/${TAG}_step_timeline_config2_$MODE.txt
done
