#!/usr/bin/env python3
"""One steady-state step of a rocprofv3 kernel_trace.csv as a timeline: for every dispatch between
two consecutive launches of the marker kernel, start offset, duration, gap to the previous end on
the same queue and the queue id.  Averaged over `--steps` steps (dispatch k of each step with
dispatch k of the others; steps whose dispatch count differs from the most common one are skipped).

    python tools/step_timeline.py gpurun_out/prof/t_kernel_trace.csv --steps 20 --skip-last 22
"""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--marker", default="k_frame_wave")
ap.add_argument("--skip-last", type=int, default=0)
ap.add_argument("--from-index", type=int, default=None, help="window = marker launches [i, i + steps] counted from the first one")
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
if a.skip_last:
    marks = marks[:-a.skip_last]
if a.from_index is not None:
    if a.from_index < 0:          # the `steps` consecutive marker launches that took the least time
        ts = [int(rows[i]["Start_Timestamp"]) for i in marks]
        best = min(range(len(marks) - a.steps), key=lambda i: ts[i + a.steps] - ts[i])
        marks = marks[best:best + a.steps + 1]
    else:
        marks = marks[a.from_index:a.from_index + a.steps + 1]
marks = marks[-a.steps - 1:]
steps = [rows[marks[k]:marks[k + 1]] for k in range(len(marks) - 1)]
common = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
steps = [s for s in steps if len(s) == common]
print("# %d steps of %d dispatches" % (len(steps), common))
wall = sum(int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"]) for s in steps) / len(steps)
print("%-52s %5s %9s %8s %8s" % ("kernel", "queue", "start_us", "dur_us", "gap_us"))
for k in range(common):
    name = steps[0][k]["Kernel_Name"][:52]
    q = steps[0][k].get("Queue_Id", "?")
    st = sum(int(s[k]["Start_Timestamp"]) - int(s[0]["Start_Timestamp"]) for s in steps) / len(steps)
    du = sum(int(s[k]["End_Timestamp"]) - int(s[k]["Start_Timestamp"]) for s in steps) / len(steps)
    gap = 0.0
    if k:
        gap = sum(int(s[k]["Start_Timestamp"]) - max(int(x["End_Timestamp"]) for x in s[:k]) for s in steps) / len(steps)
    print("%-52s %5s %9.1f %8.1f %8.1f" % (name, q, st / 1e3, du / 1e3, gap / 1e3))
print("# last end - first start: %.1f us" % (wall / 1e3))
