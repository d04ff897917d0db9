#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per dispatch, grouped by kernel.
usage: pmc_summary.py <dir with *_counter_collection.csv> [kernel-substring]"""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if want not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add((f, r["Dispatch_Id"]))
out = {k: {"dispatches": len(cnt[k]), **{c: v / len(cnt[k]) for c, v in acc[k].items()}} for k in acc}
print(json.dumps(out, indent=1))
