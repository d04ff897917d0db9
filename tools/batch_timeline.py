#!/usr/bin/env python3
"""Timeline of the LAST large rotation batch of a registration out of a `rocprofv3 --kernel-trace` csv (tools/trace_e2e.py bunny): every
kernel with start (us from the batch's first queue kernel), duration and the idle gap before it; then the totals per phase.
usage: batch_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        short = n.split("(")[0].replace("goicp::", "").replace("void ", "")[:40]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
rows.sort()
# the registration of interest = the last one in the trace; batches = runs of queue/bounds kernels between ICP stretches
reg = [r for r in rows if any(k in r[2] for k in ("bnb_queue_kernel", "bounds_queue_kernel", "task_", "bounds_tile", "icp_", "bnb_init", "copyBuffer", "bounds_kernel", "bounds_finalize"))]
# split into batches at bnb_init_kernel
batches, cur = [], []
for r in reg:
    if "bnb_init_kernel" in r[2]:
        if cur: batches.append(cur)
        cur = [r]
    elif cur:
        cur.append(r)
if cur: batches.append(cur)
big = [b for b in batches if sum(1 for r in b if "bounds_queue_kernel" in r[2]) >= 10]
print("batches:", len(batches), "with >= 10 rounds:", len(big))
tot_tail_us = 0.0
for bi, b in enumerate(batches):
    b = [r for r in b if "icp_" not in r[2] and "bounds_kernel" not in r[2] and "bounds_finalize" not in r[2]]
    rounds = []
    for r in b:
        if "bnb_queue_kernel" in r[2]: rounds.append([r])
        elif rounds: rounds[-1].append(r)
    # a tail round: its bound evaluation lasted < 8 us
    for rd in rounds:
        ev = [x for x in rd if "bounds_queue_kernel" in x[2]]
        if ev and (ev[0][1] - ev[0][0]) < 8000 and len(rd) >= 2:
            tot_tail_us += (rd[-1][1] - rd[0][0]) / 1e3
print("time inside tail rounds (queue kernel start -> last kernel end of the round, bound evaluation < 8 us): %.1f us" % tot_tail_us)
b = big[-1] if big else batches[-1]
t0, prev = b[0][0], b[0][0]
print("%9s %8s %7s  %s" % ("start", "dur", "gap", "kernel"))
for s, e, n in b:
    if "icp_" in n: break
    print("%9.1f %8.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n))
    prev = e
