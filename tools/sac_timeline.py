#!/usr/bin/env python3
"""Timeline of ONE SAC update out of a rocprofv3 kernel_trace.csv of tools/prof_sac.py: the last complete
update (k_sac_gather .. k_sac_alpha), every dispatch with its start offset, duration and queue, plus the
per-kernel mean over all updates of the run."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_sac_gather" in r["Kernel_Name"]]
ends = [i for i, r in enumerate(rows) if "k_sac_alpha" in r["Kernel_Name"]]
if not ends:                      # round 5: the temperature update rides in the policy's Adam launch (the update's last)
    adam = [i for i, r in enumerate(rows) if "k_sac_adam" in r["Kernel_Name"]]
    ends = adam[1::2]
if len(starts) < 3:
    sys.exit("no updates in the trace")
# steady updates: drop the first 5
per = []
for a, b in zip(starts[5:-1], starts[6:]):
    per.append(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]))
print("# %d updates; gather-to-gather period: mean %.1f us, min %.1f us" % (len(per), sum(per) / len(per) / 1e3, min(per) / 1e3))
a = starts[-2]; t0 = int(rows[a]["Start_Timestamp"])
b = [e for e in ends if e > a][0]
print("# one update (second to last): %.1f us from the gather's start to the end of the update's last kernel" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3))
print("%10s %9s %6s  %s" % ("start_us", "dur_us", "queue", "kernel"))
for r in rows[a:b + 1]:
    print("%10.1f %9.1f %6s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                   r.get("Queue_Id", "?"), r["Kernel_Name"][:110]))
agg = collections.OrderedDict()
n_up = len(starts) - 5
for r in rows[starts[5]:]:
    k = agg.setdefault(r["Kernel_Name"][:110], [0, 0]); k[0] += 1; k[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print("\n# per update (mean over %d): summed kernel time %.1f us" % (n_up, tot / n_up / 1e3))
print("kernel,calls_per_update,avg_us,us_per_update")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%.2f,%.2f,%.2f' % (n, c / n_up, d / c / 1e3, d / n_up / 1e3))
