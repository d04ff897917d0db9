#!/usr/bin/env python3
"""What would an ICP refinement overlapped with the next rotation batches do to the SEARCH?  The engine folds a refinement's result in k
batches late (env GOICP_ICP_DELAY_BATCHES = k: the ICP still runs synchronously, so wall times here are serial -- the cube-bound counts are
the measurement): the batches in between run against the unrefined incumbent.  One process per k."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time, numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from conftest import cloud, load_pkg, skull_problem
    pkg = load_pkg()
    which = sys.argv[2]
    if which == "bunny": model, data, mse = cloud("model_bunny"), cloud("data_bunny"), 1e-3
    elif which == "bunny10": model, data, mse = cloud("model_bunny"), cloud("data_bunny", 10), 1e-3
    elif which == "skull": model, data, _, _ = skull_problem(); mse = 1e-3
    elif which == "spanner": model, data, mse = cloud("spanner_target"), cloud("spanner_source"), 1e-4
    else:
        from cuda_go_icp_amd import synth
        model, data, _, _ = synth.make_pair(**{k: synth.S1[k] for k in ("seed", "M", "N")}); mse = 1e-4
    eng = pkg.FastGoICP(model, data, mse)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.run(); ts.append(time.perf_counter() - t0)
    c = eng.counters
    print("%-8s delay %s batches: %.2f ms (serial ICP)  cube bounds %9d  rot nodes %5d  icp iters %4d  sse %.6g" % (which, os.environ.get("GOICP_ICP_DELAY_BATCHES", "0"), 1e3 * min(ts), c.cubes, c.rot_pops, c.icp_iters, eng.get_best_error()), flush=True)
else:
    for which in ("bunny", "bunny10", "skull", "spanner", "s1"):
        for k in ("0", "1", "2", "3"):
            env = dict(os.environ); env["GOICP_ICP_DELAY_BATCHES"] = k
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", which], env=env)
