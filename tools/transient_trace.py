"""Reader of a rocprofv3 kernel trace of tools/transient_probe.py: per kernel, mean duration and launches per step
in windows of frame-kernel periods behind the LAST reset of the trace.   python tools/transient_trace.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0][:40]
# last reset = last k_reset_env launch
last_reset = max(i for i, r in enumerate(rows) if "k_reset_env" in r["Kernel_Name"] or "k_transpose_ring" in r["Kernel_Name"])
rows = rows[last_reset + 1:]
frames = [i for i, r in enumerate(rows) if "k_frame_wave" in r["Kernel_Name"]]
print("frames behind the last reset:", len(frames))
wins = [(0, 10), (10, 20), (20, 40), (40, 80), (80, 160), (160, 280)]
for a, b in wins:
    if b >= len(frames):
        break
    seg = rows[frames[a]:frames[b]]
    t0, t1 = int(rows[frames[a]]["Start_Timestamp"]), int(rows[frames[b]]["Start_Timestamp"])
    per = collections.defaultdict(list)
    for r in seg:
        per[name(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    n = b - a
    print("steps %3d..%3d: %.1f us/step | " % (a, b, (t1 - t0) / 1e3 / n) +
          "  ".join("%s %.1fx%.1f" % (k.replace("void ", "")[:18], len(v) / n, sum(v) / len(v)) for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:9]))
