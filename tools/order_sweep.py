#!/usr/bin/env python3
"""Outer-search order: the six workloads with ub_tiebreak 0 / 1 (best of three): wall time, cube bounds, rotation nodes, ICP iterations."""
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
from cuda_go_icp_amd import synth  # noqa: E402
work = []
work.append(("bunny", cloud("model_bunny"), cloud("data_bunny"), 1e-3, {}))
work.append(("bunny/10", cloud("model_bunny"), cloud("data_bunny", 10), 1e-3, {}))
t, s, _, _ = skull_problem()
work.append(("skull", t, s, 1e-3, {}))
work.append(("spanner", cloud("spanner_target"), cloud("spanner_source"), 1e-4, {}))
t, s, _, _ = synth.make_pair(**{k: synth.S1[k] for k in ("seed", "M", "N")})
work.append(("S1 1e-4", t, s, 1e-4, {}))
work.append(("bunny 1e-4", cloud("model_bunny"), cloud("data_bunny"), 1e-4, {}))
if "--s2" in sys.argv:
    t, s, Rgt, tgt = synth.make_pair(seed=synth.S2["seed"], M=synth.S2["M"], N=synth.S2["N"], amp=0.15)
    reg = pkg.Registration(t, s, 1e-3, dt_size=512)
    floor = float(reg.compute_sse_error(Rgt, tgt)) / len(s)
    reg.close()
    work.append(("S2 0.15", t, s, 1.2 * floor, {"dt_size": 512}))
extra = {}
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=")
        extra[k] = float(v) if "." in v else int(v)
for name, tg, sr, mse, kw in work:
    for tb in (0, 1):
        best = None
        for rep in range(3):
            eng = pkg.FastGoICP(tg, sr, mse, ub_tiebreak=tb, **kw, **extra)
            t0 = time.perf_counter()
            eng.run()
            wall = time.perf_counter() - t0
            c = eng.counters
            row = (wall, float(eng.get_best_error()), int(c.cubes), int(c.rot_pops), int(c.icp_iters), int(c.icp_runs), int(c.bounds_launches))
            eng.registration.close()
            if best is None or row[0] < best[0]:
                best = row
        print("%-10s ub_tiebreak %d: %8.2f ms  sse %.5f  cube bounds %9d  rot nodes %5d  icp iters %5d in %2d runs  rounds %4d" % ((name, tb, best[0] * 1e3) + best[1:]), flush=True)
