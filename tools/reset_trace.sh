# Kernel durations inside aomarl_reset (256 environments, production 40x40) under rocprofv3:
#   bash tools/reset_trace.sh <tag>  -> gpurun_out/<tag>_reset_kernel_stats.csv, <tag>_reset_timeline.txt
set -e
R=$PWD; TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/prof_${TAG}_reset; rm -rf $D
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 $R/tools/reset_probe.py 256 once > $R/gpurun_out/${TAG}_reset_probe.out 2>&1
S=$(find $D -name '*kernel_stats.csv' | head -1); T=$(find $D -name '*kernel_trace.csv' | head -1)
cp $S $R/gpurun_out/${TAG}_reset_kernel_stats.csv
python3 - $T > $R/gpurun_out/${TAG}_reset_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 4000 dispatches: deep inside the last reset
rows = rows[-6000:-2000]
t0 = int(rows[0]["Start_Timestamp"])
print("# 60 consecutive dispatches inside a reset: queue, start us, duration us, kernel")
for r in rows[1000:1060]:
    print("%3s %10.1f %8.1f  %s" % (r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) * 1e-3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3, r["Kernel_Name"][:60]))
span = (int(rows[-1]["End_Timestamp"]) - t0) * 1e-3
busy = {}
for r in rows:
    k = r["Kernel_Name"][:40]
    busy.setdefault(k, [0, 0.0]); busy[k][0] += 1; busy[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
print("# %d dispatches over %.1f us" % (len(rows), span))
for k, (n, t) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
    print("# %-42s %6d calls  %10.1f us total  %7.2f us avg" % (k, n, t, t / n))
PY
rm -rf $D
