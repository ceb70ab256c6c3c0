"""Per hardware queue of a rocprofv3 kernel trace: busy time, gaps and kernels per step over the steady part of a
bench run (development aid): python tools/queue_busy.py <kernel_trace.csv> [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fr = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_frame_wave<3, 1, true, false, false")]
# the longest run of consecutive frame kernels less than 1 ms apart (no reset in between): its last `steps` periods
best, cur = (0, 0), 0
for k in range(1, len(fr)):
    if int(rows[fr[k]]["Start_Timestamp"]) - int(rows[fr[k - 1]]["Start_Timestamp"]) < 1000000:
        cur += 1
        if cur > best[0]:
            best = (cur, k)
    else:
        cur = 0
end = best[1]
steps = min(steps, best[0] - 2)
lo, hi = fr[end - steps], fr[end]
t0, t1 = int(rows[lo]["Start_Timestamp"]), int(rows[hi]["Start_Timestamp"])
print("# %d steps, %.1f us per step (frame kernel start to frame kernel start)" % (steps, (t1 - t0) * 1e-3 / steps))
byq = collections.defaultdict(list)
for r in rows[lo:hi]:
    byq[r.get("Queue_Id", "?")].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) * 1e-3 / steps
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rs, rs[1:])]
    pos = sum(g for g in gaps if g > 0) * 1e-3 / steps
    print("queue %s: %.1f launches per step, busy %.1f us per step, idle between its kernels %.1f us per step" % (q, len(rs) / steps, busy, pos))
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in rs:
        k = r["Kernel_Name"][:44]
        per[k][0] += 1; per[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("      %-46s %5.2f per step  %7.1f us avg  %7.1f us per step" % (k, n / steps, t / n, t / steps))
