#!/usr/bin/env python3
"""One prove-the-optimum registration (bunny, mse 3e-5 unless given) -- the command tools/deep_trace.sh puts under rocprofv3.
usage: python3 tools/deep_run.py [mse] [key=value engine parameters ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _pkg  # noqa: E402

pkg = _pkg()
pkg.load_library()
g = os.path.join(ROOT, "tests", "golden")
model = np.fromfile(os.path.join(g, "model_bunny.f32"), dtype="<f4").reshape(-1, 3)
data = np.fromfile(os.path.join(g, "data_bunny.f32"), dtype="<f4").reshape(-1, 3)
mse = float(sys.argv[1]) if len(sys.argv) > 1 else 3e-5
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[2:]}
eng = pkg.FastGoICP(model, data, mse, **kw)
for _ in range(int(os.environ.get("DEEP_RUN_REPEATS", "1"))):
    t0 = time.perf_counter(); eng.run(); wall = time.perf_counter() - t0
c = eng.counters
print("mse %g %s: %.3f s  sse %.6f  cube bounds %d  from tiles %.1f %%  rot nodes %d  rounds %d  lane batches %d  queue fallbacks %d" % (
    mse, kw, wall, eng.get_best_error(), c.cubes, 800.0 * c.tile_expansions / c.cubes, c.rot_pops, c.bounds_launches, c.lane_batches, c.queue_fallbacks), flush=True)
