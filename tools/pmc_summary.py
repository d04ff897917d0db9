#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per dispatch, grouped by kernel.
usage: pmc_summary.py <dir with *_counter_collection.csv> [kernel-substring] [--largest]
--largest: only the dispatches with the kernel's largest grid (the full-size launches of a bench run; warm-up
and scoring launches of other sizes are left out).  Also reports the average dispatch duration (End - Start)."""
import collections
import csv
import glob
import json
import os
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
largest = "--largest" in sys.argv
root = args[0]
want = args[1] if len(args) > 1 else ""
rows = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if want in k:
            rows[k].append((f, r))
out = {}
for k, lst in rows.items():
    gmax = max(int(r["Grid_Size"]) for _, r in lst)
    acc, cnt, dur = collections.defaultdict(float), collections.defaultdict(set), {}
    for f, r in lst:
        if largest and int(r["Grid_Size"]) != gmax:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])          # every counter lives in one pass only
        cnt[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
        dur[(f, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out[k] = {"dispatches": max(len(v) for v in cnt.values()), "grid_size": gmax if largest else None,
              "avg_duration_ns_under_pmc": sum(dur.values()) / len(dur), **{c: v / len(cnt[c]) for c, v in acc.items()}}
print(json.dumps(out, indent=1))
