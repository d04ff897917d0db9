"""Dev script (not a test): full-bunny registration wall time for a sweep of the batching parameters."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import cloud, load_pkg  # noqa: E402

pkg = load_pkg()
model, data = cloud("model_bunny"), cloud("data_bunny")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for tb, rb in ((32, 8), (32, 64), (64, 64)):
    ts = []
    for _ in range(reps):
        eng = pkg.FastGoICP(model, data, 1e-3, trans_batch=tb, rot_batch=rb)
        t0 = time.perf_counter()
        eng.run()
        ts.append(time.perf_counter() - t0)
        r = eng.registration.poll()
    print("trans_batch %3d rot_batch %2d: register_s min %.4f median %.4f  sse %.5f  cubes %d  launches %d  icp_iters %d" % (
        tb, rb, min(ts), float(np.median(ts)), r.best_sse, r.counters.cubes, r.counters.bounds_launches, r.counters.icp_iters), flush=True)
