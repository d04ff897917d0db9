#!/usr/bin/env python3
"""Verbose trace of one registration (ICP runs, incumbent updates, phase times).  usage: python3 tools/trace_e2e.py [bunny|bunny10|skull|spanner|s1] [key=value ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import _pkg  # noqa: E402
from conftest import cloud, skull_problem  # noqa: E402

pkg = _pkg()
pkg.load_library()
which = sys.argv[1] if len(sys.argv) > 1 else "bunny"
kw = {}
for a in sys.argv[2:]:
    k, v = a.split("=")
    kw[k] = float(v) if ("." in v or "e" in v) else int(v)
mse = kw.pop("mse", None)
if which == "bunny":
    model, data, m0 = cloud("model_bunny"), cloud("data_bunny"), 1e-3
elif which == "bunny10":
    model, data, m0 = cloud("model_bunny"), cloud("data_bunny", 10), 1e-3
elif which == "skull":
    model, data, _, _ = skull_problem(); m0 = 1e-3
elif which == "spanner":
    model, data, m0 = cloud("spanner_target"), cloud("spanner_source"), 1e-4
else:
    from cuda_go_icp_amd import synth
    model, data, _, _ = synth.make_pair(**{k: synth.S1[k] for k in ("seed", "M", "N")}); m0 = 1e-4
reps = kw.pop("reps", 1)
for r in range(reps):
    eng = pkg.FastGoICP(model, data, mse or m0, verbose=2 if r == reps - 1 else 0, **kw)
    t0 = time.perf_counter()
    eng.run()
    wall = time.perf_counter() - t0
    c = eng.counters
    print("%s %s: %.2f ms sse %.6g cubes %d rot %d icp runs %d iters %d rounds %d" % (which, kw, wall * 1e3, eng.get_best_error(), c.cubes, c.rot_pops, c.icp_runs, c.icp_iters, c.bounds_launches) + " queue fallbacks %d" % c.queue_fallbacks, flush=True)
    eng.registration.close()
