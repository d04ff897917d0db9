#!/usr/bin/env python3
"""LDS-staged DT tiles on the search path (lds_tiles 0 / 1 / 2): registrations that have to dig (thresholds below the optimum's
error: the search proves the optimum) and the default ones, wall time / cube bounds / share evaluated from tiles.
usage: python3 tools/tile_e2e.py [mse ...]      (default: 1e-3 1e-4; add 2e-5 for the 35 s run)"""
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
mses = [float(x) for x in sys.argv[1:]] or [1e-3, 1e-4]
extra = {}
for kv in os.environ.get("GOICP_TILE_PARAMS", "").split(","):
    if "=" in kv:
        k, v = kv.split("=")
        extra[k] = float(v) if "." in v else int(v)
for name, src in (("bunny", data), ("bunny/10", np.ascontiguousarray(data[::10]))):
    for mse in mses:
        if name == "bunny/10" and mse < 1e-4:
            continue
        for tiles in (0, 1, 2):
            best = None
            for rep in range(3 if mse >= 1e-4 else 1):
                eng = pkg.FastGoICP(model, src, mse, lds_tiles=tiles, **extra)
                t0 = time.perf_counter()
                eng.run()
                wall = time.perf_counter() - t0
                c = eng.counters
                row = (wall, float(eng.get_best_error()), int(c.cubes), int(c.tile_expansions) * 8, int(c.rot_pops), int(c.bounds_launches))
                eng.registration.close()
                if best is None or row[0] < best[0]:
                    best = row
            print("%-8s mse %-7g lds_tiles %d: %8.2f ms  sse %.5f  cube bounds %11d  from tiles %11d (%4.1f %%)  rot nodes %6d  rounds %5d" % (
                name, mse, tiles, best[0] * 1e3, best[1], best[2], best[3], 100.0 * best[3] / max(best[2], 1), best[4], best[5]), flush=True)
