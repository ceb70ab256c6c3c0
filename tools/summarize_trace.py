#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv over the steady-state tail of a bench run:
per-kernel mean duration and share, using only dispatches inside the last `--steps` steps
(located through the k_wfs_spot_fast launches, one per step)."""
import argparse, csv, collections, sys
ap = argparse.ArgumentParser(); ap.add_argument("trace"); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--marker", default="k_wfs_spot")
ap.add_argument("--skip-last", type=int, default=0, help="marker launches to drop at the end (bench.py's diagnostic pass)")
ap.add_argument("--from-index", type=int, default=None, help="window = marker launches [i, i + steps] counted from the first one (e.g. bench.py: 1 reset + 1 timed reset + W warm-up = 2 + W)")
ap.add_argument("--before", default=None, help="ignore everything from the first launch of this kernel on")
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if a.before:
    cut = [i for i, r in enumerate(rows) if a.before in r["Kernel_Name"]]
    if cut:
        rows = rows[:cut[0]]
marks = [int(r["Start_Timestamp"]) for r in rows if a.marker in r["Kernel_Name"]]
if a.skip_last:
    marks = marks[:-a.skip_last]
if a.from_index is not None:
    if a.from_index < 0:          # the `steps` consecutive marker launches that took the least time: the timed region
        best = min(range(len(marks) - a.steps), key=lambda i: marks[i + a.steps] - marks[i])
        marks = marks[best:best + a.steps + 1]
    else:
        marks = marks[a.from_index:a.from_index + a.steps + 1]
t0 = marks[-a.steps - 1]; t1 = marks[-1]
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
agg = collections.OrderedDict()
for r in sel:
    n = r["Kernel_Name"]; d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = agg.setdefault(n, [0, 0]); k[0] += 1; k[1] += d
tot = sum(v[1] for v in agg.values())
wall = t1 - t0
print("# steady-state window: %d steps, %.3f ms/step wall, %.3f ms/step summed kernel time" % (a.steps, wall / a.steps / 1e6, tot / a.steps / 1e6))
print("kernel,calls_per_step,avg_us,us_per_step,share")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%.2f,%.2f,%.2f,%.3f' % (n[:100], c / a.steps, d / c / 1e3, d / a.steps / 1e3, d / tot))
