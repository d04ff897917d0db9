"""Dev script (not a test): full-bunny registration with trimming, wall time and the engine's split."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import cloud, load_pkg  # noqa: E402

pkg = load_pkg()
model, data = cloud("model_bunny"), cloud("data_bunny")
for tf in (0.0, 0.1, 0.1):
    eng = pkg.FastGoICP(model, data, 1e-3, trim_fraction=tf, verbose=1)
    t0 = time.perf_counter()
    eng.run()
    r = eng.registration.poll()
    print("trim %.2f: register_s %.4f sse %.5f cubes %d launches %d icp_iters %d" % (
        tf, time.perf_counter() - t0, r.best_sse, r.counters.cubes, r.counters.bounds_launches, r.counters.icp_iters), file=sys.stderr)
