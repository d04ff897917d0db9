#!/usr/bin/env python3
"""Per-dispatch durations of the bound-evaluation kernels of ONE registration, in launch order, from a rocprofv3 kernel trace
(rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/trace_e2e.py bunny).  usage: python3 tools/e2e_dispatches.py DIR [kernel-name substring]"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
tot = {}
out = []
for r in rows:
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void goicp::", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[short] = tot.get(short, 0.0) + d
    if "bounds_queue_kernel" in name or "bounds_tile_kernel" in name:
        out.append((int(r["Start_Timestamp"]), short, d))
print("total us by kernel:")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
    print("  %-60s %10.1f" % (k[:60], v))
print("bound-evaluation dispatches in order (us):")
line = []
prev = None
for ts, short, d in out:
    if prev is not None and ts - prev > 3_000_000:      # a gap of > 3 ms: ICP ran in between
        print("  " + " ".join(line)); line = []
    line.append("%.0f" % d)
    prev = ts
print("  " + " ".join(line))
if len(sys.argv) > 2:                                       # any other kernel, by substring
    print("%s dispatches in order (us):" % sys.argv[2])
    print("  " + " ".join("%.0f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if sys.argv[2] in r["Kernel_Name"]))
