#!/usr/bin/env python3
"""Per-dispatch view of a tools/pmc_collect.sh run, joined with a kernel trace of the same command.
usage: tools/pmc_per_dispatch.py <pmc outdir> <kernel substring>      (expects <outdir>/trace/ = rocprofv3 --kernel-trace)
Prints one row per dispatch of the kernel with more than 2e5 vector-memory reads: duration, gather instructions, CU-cycles
per gather instruction, L1 tag accesses per instruction, L1 / L2 hit rates, bytes fetched beyond L2 (FETCH_SIZE x2 on
gfx950, KB -> MB), TA busy share.  The dispatch order of the two runs is the same (the search is deterministic)."""
import collections
import csv
import glob
import sys

base, kern = sys.argv[1].rstrip("/") + "/", sys.argv[2]


def load(p):
    f = glob.glob(base + p + "/*/*counter_collection.csv")[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [per[k] for k in sorted(per)]


p1, p2, p3, p4 = load("pass1"), load("pass2"), load("pass3"), load("pass4")
dur = []
for r in csv.DictReader(open(glob.glob(base + "trace/*/*kernel_trace.csv")[0])):
    if kern in r["Kernel_Name"]:
        dur.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
dur = [d for _, d in sorted(dur)]
n = min(len(p1), len(p2), len(p3), len(p4), len(dur))
print("dispatches: pmc %d / trace %d" % (len(p1), len(dur)))
print("idx   dur_us  gathers_M  cu_cycles/gather  tags/gather  L1hit  L2hit  beyondL2_MB  TB/s  TAbusy%")
for i in range(n):
    v = p3[i].get("SQ_INSTS_VMEM_RD", 0)
    if v < 2e5:
        continue
    tot, req, ta = p4[i]["TCP_TOTAL_CACHE_ACCESSES_sum"], p4[i]["TCP_TCC_READ_REQ_sum"], p4[i]["TA_TA_BUSY_sum"]
    hit, miss, fetch_mb = p2[i]["TCC_HIT_sum"], p2[i]["TCC_MISS_sum"], p1[i]["FETCH_SIZE"] / 1e3 * 2
    d = dur[i]
    print("%3d %8.1f %9.2f %12.1f %12.1f %8.3f %6.3f %10.1f %6.2f %7.1f" % (i, d / 1e3, v / 1e6, d * 1e-9 * 2.4e9 * 256 / v, tot / v, 1 - req / tot, hit / (hit + miss),
                                                                  fetch_mb, fetch_mb * 1e6 / (d * 1e-9) / 1e12, 100 * ta / 256 / (d * 2.4)))
