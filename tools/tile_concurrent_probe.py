#!/usr/bin/env python3
"""A/B of the forked tile evaluation (env GOICP_TILE_CONCURRENT = 0 / 1; one process per setting): prove-the-optimum registrations."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time, numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from conftest import cloud, load_pkg
    pkg = load_pkg()
    which, mse = sys.argv[2], float(sys.argv[3])
    if which == "bunny": model, data = cloud("model_bunny"), cloud("data_bunny")
    elif which == "bunny10": model, data = cloud("model_bunny"), cloud("data_bunny", 10)
    elif which == "spanner": model, data = cloud("spanner_target"), cloud("spanner_source")
    else:
        from cuda_go_icp_amd import synth
        model, data, _, _ = synth.make_pair(**{k: synth.S1[k] for k in ("seed", "M", "N")})
    eng = pkg.FastGoICP(model, data, mse)
    ts = []
    for _ in range(2):
        t0 = time.perf_counter(); eng.run(); ts.append(time.perf_counter() - t0)
    c = eng.counters
    print("%-8s mse %g tile_concurrent %s: %.3f s  cube bounds %d  from tiles %.1f %%  sse %.6g  R00 %.7f" % (which, mse, os.environ.get("GOICP_TILE_CONCURRENT", "1(default)"), min(ts), c.cubes, 800.0 * c.tile_expansions / max(c.cubes, 1), eng.get_best_error(), eng.optR[0, 0]), flush=True)
else:
    for which, mse in (("bunny", "3e-5"), ("bunny", "1e-4"), ("bunny10", "3e-5"), ("s1", "3e-5"), ("spanner", "2e-5"), ("bunny", "1e-3")):
        for v in ("0", "1"):
            env = dict(os.environ); env["GOICP_TILE_CONCURRENT"] = v
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", which, mse], env=env)
