#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (csv): wall span, the union of the kernels' intervals (GPU busy with at least one kernel), the
sum of the durations per kernel name (> the union when kernels overlap) and the idle gaps.  usage: trace_busy.py <kernel_trace.csv> [last_ms]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if len(sys.argv) > 2:            # only the last W milliseconds of the trace (a registration at the end of the process)
    tend = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= tend - int(float(sys.argv[2]) * 1e6)]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
big = []
last_name = rows[0][2]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        big.append((s - cur_e, last_name.split("(")[0][-40:], n.split("(")[0][-40:], (cur_e - t0) / 1e6))
        cur_s, cur_e = s, e
        last_name = n
    else:
        if e > cur_e:
            last_name = n
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
per = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    k = n.split("(")[0][:70]
    per[k][0] += e - s
    per[k][1] += 1
print("span %.3f s, busy (union) %.3f s = %.1f %%, sum of durations %.3f s, %d kernels, %d gaps (median %.1f us, total %.3f s)" % (
    (t1 - t0) / 1e9, busy / 1e9, 100.0 * busy / (t1 - t0), sum(v[0] for v in per.values()) / 1e9, len(rows), len(gaps),
    sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0, sum(gaps) / 1e9))
for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])[:12]:
    print("  %-70s %8.3f s %7d calls  mean %8.1f us" % (k, v[0] / 1e9, v[1], v[0] / v[1] / 1e3))
hist = defaultdict(lambda: [0, 0])
for g, a, b, _ in big:
    hist[(a, b)][0] += g
    hist[(a, b)][1] += 1
print("idle time by (kernel before -> kernel after):")
for k, v in sorted(hist.items(), key=lambda kv: -kv[1][0])[:10]:
    print("  %-40s -> %-40s %8.1f us in %4d gaps (mean %.1f us)" % (k[0], k[1], v[0] / 1e3, v[1], v[0] / v[1] / 1e3))
print("largest gaps:", ", ".join("%.0f us at %.2f ms" % (g / 1e3, t) for g, _, _, t in sorted(big, reverse=True)[:10]))
