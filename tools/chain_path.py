#!/usr/bin/env python3
"""Pipelined step: the control / agent chain's path between the end of frame kernel t and the start of frame kernel
t + 2 (the frame that needs the command computed from frame t's measurement).  From a rocprofv3 kernel_trace.csv:
per step, times relative to the END of a frame kernel of: start / end of each kernel of the chain's queue that
follows it (tail product, k_assemble_state, k_actor_fused, k_compose_rewards, head product, k_delay_ahead,
k_post_delay), and the start of the next two frame kernels; averaged over the steps.
    python tools/chain_path.py trace.csv [--first -45] [--count 40] [--raw N]   (--raw: N steps kernel by kernel)"""
import argparse
import csv
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--first", type=int, default=-45, help="index of the first frame kernel looked at (negative: from the end)")
ap.add_argument("--count", type=int, default=40)
ap.add_argument("--raw", type=int, default=0)
args = ap.parse_args()
rows = list(csv.DictReader(open(args.trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first, count = args.first, args.count
fr = [i for i, r in enumerate(rows) if "k_frame_wave" in r["Kernel_Name"]]
qcol = "Queue_Id" if "Queue_Id" in rows[0] else "Queue_ID"
chain_q = None
for r in rows[fr[first]:]:
    if "k_assemble_state" in r["Kernel_Name"]:
        chain_q = r[qcol]
        break
acc = defaultdict(list)
order = []
sel = fr[first:][:count]
for n, i in enumerate(sel):
    if fr.index(i) + 2 >= len(fr):
        break
    end = int(rows[i]["End_Timestamp"])
    f1, f2 = fr[fr.index(i) + 1], fr[fr.index(i) + 2]
    acc["next frame start"].append((int(rows[f1]["Start_Timestamp"]) - end) / 1e3)
    acc["next frame end"].append((int(rows[f1]["End_Timestamp"]) - end) / 1e3)
    acc["frame after next start"].append((int(rows[f2]["Start_Timestamp"]) - end) / 1e3)
    seen = defaultdict(int)
    for k in range(i + 1, f2 + 1):
        r = rows[k]
        if r[qcol] != chain_q or int(r["Start_Timestamp"]) < end:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
        seen[name] += 1
        key = "%s #%d" % (name, seen[name])
        if key not in order:
            order.append(key)
        acc[key + " start"].append((int(r["Start_Timestamp"]) - end) / 1e3)
        acc[key + " end"].append((int(r["End_Timestamp"]) - end) / 1e3)
mean = lambda v: sum(v) / len(v)      # noqa: E731
print("# %d steps; times in us after the END of frame kernel t (mean [min .. max]); chain queue %s" % (len(acc["next frame start"]), chain_q))
for key in ("next frame start", "next frame end", "frame after next start"):
    v = acc[key]
    print("%-44s %8.1f  [%7.1f .. %7.1f]" % (key, mean(v), min(v), max(v)))
order.sort(key=lambda k: mean(acc[k + " start"]))
for key in order:
    s, e = acc[key + " start"], acc[key + " end"]
    print("%-34s n=%3d  start %8.1f [%7.1f .. %7.1f]   end %8.1f   dur %6.1f" % (key, len(s), mean(s), min(s), max(s), mean(e), mean(e) - mean(s)))
if args.raw:
    nraw = args.raw
    i0 = sel[len(sel) // 2]
    t0 = int(rows[i0]["Start_Timestamp"])
    print("\n# raw: %d steps from a frame kernel's start (us): kernel, queue, start -> end (duration)" % nraw)
    for r in rows[i0:fr[fr.index(i0) + nraw] + 1]:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:26]
        print("%-28s q%s  %8.1f -> %8.1f  (%6.1f)" % (n, r[qcol], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                      (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
