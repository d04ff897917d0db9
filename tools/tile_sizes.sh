#!/bin/bash
# LDS tile capacity experiment: the tile probe (depths 5, 6, 7) and the prove-the-optimum bunny run with the in-tree library
# (32 KB tiles) and with the variant builds of csrc/build/variants (48 KB, 64 KB).
for lib in "" cuda-go-icp_amd/csrc/build/variants/libgoicp_12288.so cuda-go-icp_amd/csrc/build/variants/libgoicp_16384.so; do
  echo "=== library: ${lib:-in-tree (8192 floats)}"
  GOICP_LIBRARY=${lib:+$PWD/$lib} python3 tools/tile_probe.py bunny 2>&1 | grep "chunks  32" | grep -E "depth  (5|6|7|8)"
  GOICP_LIBRARY=${lib:+$PWD/$lib} python3 - <<PY
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from __graft_entry__ import _pkg
pkg = _pkg(); pkg.load_library()
g = os.path.join("tests", "golden")
model = np.fromfile(os.path.join(g, "model_bunny.f32"), dtype="<f4").reshape(-1, 3)
data = np.fromfile(os.path.join(g, "data_bunny.f32"), dtype="<f4").reshape(-1, 3)
for spread, tmin in ((10.0, 8), (16.0, 8)):
    eng = pkg.FastGoICP(model, data, 3e-5, lds_tiles=1, tile_spread_vox=spread, tile_min=tmin)
    t0 = time.perf_counter(); eng.run(); wall = time.perf_counter() - t0
    c = eng.counters
    print("mse 3e-5 spread %.0f min %d: %.2f s sse %.5f cubes %d tiles %.1f %% rot %d" % (spread, tmin, wall, eng.get_best_error(), c.cubes, 100.0 * c.tile_expansions * 8 / c.cubes, c.rot_pops), flush=True)
    eng.registration.close()
PY
done
