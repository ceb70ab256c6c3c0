#!/usr/bin/env python3
"""Frame pipeline: what does each frame kernel wait for?  From a rocprofv3 kernel_trace.csv: for every k_frame_wave
launch, the time from the end of (previous frame kernel | last k_post_delay | last copyBuffer [origin snapshot] |
last k_extrude_scatter) before its start to its start.   python tools/pipe_gaps.py trace.csv [first] [count]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else -40
count = int(sys.argv[3]) if len(sys.argv) > 3 else 30
fr = [i for i, r in enumerate(rows) if "k_frame_wave" in r["Kernel_Name"]]
sel = fr[first:][:count]
print("%10s %9s %9s | start minus end of: %9s %9s %9s %9s" % ("frame", "dur_us", "period", "prev frame", "post_delay", "snapshot", "scatter"))
prev_start = None
for i in sel:
    st, en = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    def last_end(pat, j=i):
        for k in range(j - 1, max(j - 80, -1), -1):
            if pat in rows[k]["Kernel_Name"] and int(rows[k]["End_Timestamp"]) <= st + 200000:
                return int(rows[k]["End_Timestamp"])
        return None
    vals = [last_end("k_frame_wave"), last_end("k_post_delay"), last_end("copyBuffer"), last_end("k_extrude_scatter")]
    print("%10d %9.1f %9.1f | %29.1f %9.1f %9.1f %9.1f" % (i, (en - st) / 1e3, (st - prev_start) / 1e3 if prev_start else 0.0,
          *[((st - v) / 1e3 if v else float("nan")) for v in vals]))
    prev_start = st
