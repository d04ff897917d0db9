"""Dev script (not a test): repeated full-bunny registrations, wall time per run (min / median)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import cloud, load_pkg  # noqa: E402

pkg = load_pkg()
model, data = cloud("model_bunny"), cloud("data_bunny")
ts = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 7):
    eng = pkg.FastGoICP(model, data, 1e-3)
    t0 = time.perf_counter()
    eng.run()
    ts.append(time.perf_counter() - t0)
    r = eng.registration.poll()
print("register_s min %.4f median %.4f  sse %.6f  cubes %d  launches %d  icp_iters %d" % (
    min(ts), float(np.median(ts)), r.best_sse, r.counters.cubes, r.counters.bounds_launches, r.counters.icp_iters))
