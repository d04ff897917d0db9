#!/usr/bin/env python3
"""Per-stream timeline out of a `rocprofv3 --kernel-trace` of `tools/overlap_probe.py --short` (the measured rejection of the
overlapped ICP refinement).  Prints, for a window in which the ICP stream and the bounds stream are both busy, every kernel with its queue,
start and duration, and the gaps of the ICP chain: how long an ICP pass / finalize waits behind the bound evaluation's workgroups.
usage: overlap_timeline.py <kernel_trace.csv> [window_us]"""
import csv
import sys

path = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 1500.0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("Name") or ""
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = r.get("Queue_Id") or r.get("Stream_Id") or "?"
        kind = "icp" if "icp_" in name else ("bounds" if "bounds_" in name else None)
        if kind:
            rows.append((s, e, kind, q, name.split("(")[0].replace("goicp::", "").replace("void ", "")[:48]))
rows.sort()
icp = [r for r in rows if r[2] == "icp"]
bnd = [r for r in rows if r[2] == "bounds"]
# the side-by-side phase: ICP kernels that start after the LAST standalone bounds phase began and while bounds kernels are running
def busy_beside(t):
    return any(b[0] <= t <= b[1] for b in near(t))
import bisect
bstarts = [b[0] for b in bnd]
def near(t):
    i = bisect.bisect_right(bstarts, t)
    return bnd[max(0, i - 4):i + 1]
beside = [r for r in icp if busy_beside(r[0])]
alone = [r for r in icp if not busy_beside(r[0])]
def chain_stats(rs, tag):
    if len(rs) < 4:
        return
    durs = [(r[1] - r[0]) / 1e3 for r in rs]
    gaps = [(rs[i + 1][0] - rs[i][1]) / 1e3 for i in range(len(rs) - 1) if rs[i + 1][0] - rs[i][1] < 5e6]
    import statistics as st
    print("%-28s %5d kernels: duration median %.1f us (p90 %.1f), gap to the next ICP kernel median %.1f us (p90 %.1f, max %.1f)" %
          (tag, len(rs), st.median(durs), sorted(durs)[int(0.9 * len(durs))], st.median(gaps), sorted(gaps)[int(0.9 * len(gaps))], max(gaps)))
chain_stats(alone, "ICP kernels, GPU otherwise idle")
chain_stats(beside, "ICP kernels beside bounds")
if beside:
    t0 = beside[len(beside) // 2][0]
    print("\ntimeline, %.0f us window in the side-by-side phase (t in us from the window start):" % win)
    print("%10s %9s  %-6s %-6s %s" % ("start", "dur", "queue", "kind", "kernel"))
    for s, e, kind, q, name in rows:
        if t0 <= s <= t0 + win * 1e3:
            print("%10.1f %9.1f  %-6s %-6s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, kind, name))
