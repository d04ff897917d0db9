#!/usr/bin/env python3
"""HIP ICP against the CPU oracle, iteration by iteration, on the skull data set (the reference's icp_iter_skull.json pins 1 / 2 / 10 iterations)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import load_pkg, skull_problem, golden
import oracle as O
pkg = load_pkg(); pkg.load_library()
target, source, _, _ = skull_problem()
src = np.ascontiguousarray(source[::10])
g = golden("icp_iter_skull")
c0 = g["cases"][4 if "--pose-b" in sys.argv else 0]
kd = O.KdTree(target)
print("start pose", c0["R0"], c0["t0"])
for kw in ({}, {"morton_sort": 0}):
    reg = pkg.Registration(target, src, 1e-3, trans_batch=1, wide_children=0, **kw)
    print("engine options", kw)
    for k in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 20, 40):
        err, R, t = pkg.IterativeClosestPoint3D(reg, k, c0["err_diff"], c0["R0"], c0["t0"]).run()
        oerr, oR, ot, _ = kd.icp_run(src, c0["R0"], c0["t0"], k, c0["err_diff"])
        ref = [c for c in (g["cases"][4:] if "--pose-b" in sys.argv else g["cases"][:4]) if c["max_iter"] == k]
        extra = ""
        if ref:
            extra = "  | vs reference: HIP dR %.2e  oracle dR %.2e  err ref %.6g" % (np.abs(R.ravel() - np.array(ref[0]["R"])).max(), np.abs(oR.ravel() - np.array(ref[0]["R"])).max(), ref[0]["err"])
        print("  iters %2d: HIP err %.6g oracle err %.6g  max|dR| %.2e max|dt| %.2e%s" % (k, err, oerr, np.abs(R - oR).max(), np.abs(t - ot).max(), extra))
    reg.close()
