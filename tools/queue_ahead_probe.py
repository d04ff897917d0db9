#!/usr/bin/env python3
"""A/B of the one-chunk-ahead submission of the inner-BnB rounds (env GOICP_QUEUE_AHEAD = 0 / 1; one process per setting)."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time, numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from conftest import cloud, load_pkg, skull_problem
    pkg = load_pkg()
    which, mse = sys.argv[2], float(sys.argv[3])
    if which == "bunny": model, data = cloud("model_bunny"), cloud("data_bunny")
    elif which == "bunny10": model, data = cloud("model_bunny"), cloud("data_bunny", 10)
    elif which == "skull": model, data, _, _ = skull_problem()
    elif which == "spanner": model, data = cloud("spanner_target"), cloud("spanner_source")
    else:
        from cuda_go_icp_amd import synth
        model, data, _, _ = synth.make_pair(**{k: synth.S1[k] for k in ("seed", "M", "N")})
    eng = pkg.FastGoICP(model, data, mse)
    ts = []
    for _ in range(8 if mse >= 1e-4 else 2):
        t0 = time.perf_counter(); eng.run(); ts.append(time.perf_counter() - t0)
    c = eng.counters
    print("%-8s mse %g queue_ahead %s: median %.2f ms min %.2f  cube bounds %d  rounds %d  sse %.6g" % (which, mse, os.environ.get("GOICP_QUEUE_AHEAD", "default"), 1e3 * float(np.median(ts[1:])), 1e3 * min(ts), c.cubes, c.bounds_launches, eng.get_best_error()), flush=True)
else:
    for which, mse in (("bunny", "1e-3"), ("bunny10", "1e-3"), ("skull", "1e-3"), ("spanner", "1e-4"), ("s1", "1e-4"), ("bunny", "1e-4"), ("bunny", "3e-5")):
        for v in ("0", "1"):
            env = dict(os.environ); env["GOICP_QUEUE_AHEAD"] = v
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", which, mse], env=env)
