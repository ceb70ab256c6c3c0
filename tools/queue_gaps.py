"""Idle time between consecutive kernels of the frame kernel's queue, from a rocprofv3 kernel trace (development aid).
usage: queue_gaps.py <dir with *kernel_trace.csv> [marker substring]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_frame_wave<3, 1, true, false, false, true>"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fr = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
fr = fr[len(fr) // 3: 2 * len(fr) // 3 + 1]
gaps, durs = collections.OrderedDict(), collections.defaultdict(list)
for a, b in zip(fr[:-1], fr[1:]):
    q = rows[a]["Queue_Id"]
    sq = [r for r in rows[a:b + 1] if r["Queue_Id"] == q]
    for x, y in zip(sq[:-1], sq[1:]):
        key = (x["Kernel_Name"][:26], y["Kernel_Name"][:26])
        gaps.setdefault(key, []).append((int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3)
        durs[key].append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3)
n = len(fr) - 1
tot = 0.0
for k, v in gaps.items():
    print("%-28s -> %-28s gap %6.1f us  (first runs %6.1f us)  x%.2f/step" % (k[0], k[1], sum(v) / len(v), sum(durs[k]) / len(durs[k]), len(v) / n))
    tot += sum(v) / n
print("steps %d: %.1f us per step between frame kernels, %.1f us of it idle on that queue" % (n, (int(rows[fr[-1]]["Start_Timestamp"]) - int(rows[fr[0]]["Start_Timestamp"])) / 1e3 / n, tot))
