#!/usr/bin/env python3
"""The prove-the-optimum bunny registration (mse below the optimum's error) with the tile list off / on at several spreads.
usage: python3 tools/tile_deep.py [mse=3e-5] [spread,min ...]"""
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
cfgs = [(0, 6.0, 12)] + [(1, float(a.split(",")[0]), int(a.split(",")[1])) for a in (sys.argv[2:] or ["6,12", "10,8", "16,8"])]
for tiles, spread, tmin in cfgs:
    eng = pkg.FastGoICP(model, data, mse, lds_tiles=tiles, tile_spread_vox=spread, tile_min=tmin)
    t0 = time.perf_counter(); eng.run(); wall = time.perf_counter() - t0
    c = eng.counters
    print("mse %g lds_tiles %d spread %4.1f min %2d: %6.2f s  sse %.5f  cube bounds %d  from tiles %4.1f %%  rot nodes %d  rounds %d" % (
        mse, tiles, spread, tmin, wall, eng.get_best_error(), c.cubes, 100.0 * c.tile_expansions * 8 / c.cubes, c.rot_pops, c.bounds_launches), flush=True)
    eng.registration.close()
