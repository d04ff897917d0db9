#!/usr/bin/env python3
"""A/B of the two-lane rounds (env GOICP_LANES = 1 / 2, GOICP_LANE_MIN = smallest batch that is cut in two; one process per setting):
registration wall time (median of the repeats after the first), cube bounds, rotation nodes and the result, which must not depend on it."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time, numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from conftest import cloud, load_pkg, skull_problem
    pkg = load_pkg()
    which, mse, reps = sys.argv[2], float(sys.argv[3]), int(sys.argv[4])
    if which == "bunny": model, data = cloud("model_bunny"), cloud("data_bunny")
    elif which == "bunny10": model, data = cloud("model_bunny"), cloud("data_bunny", 10)
    elif which == "spanner": model, data = cloud("spanner_target"), cloud("spanner_source")
    elif which == "skull": model, data, _, _ = skull_problem()
    else:
        from cuda_go_icp_amd import synth
        model, data, _, _ = synth.make_pair(**{k: synth.S1[k] for k in ("seed", "M", "N")})
    kw = {k: int(v) for k, v in (a.split('=') for a in os.environ.get('GOICP_PROBE_KW', '').split(',') if a)}      # e.g. rot_batch=128
    eng = pkg.FastGoICP(model, data, mse, **kw)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); eng.run(); ts.append(time.perf_counter() - t0)
    c = eng.counters
    t = float(np.median(ts[1:])) if len(ts) > 2 else min(ts)
    print("%-8s %s mse %g lanes %s min %s work %s auto-lanes %s: %9.2f ms  cube bounds %d  rot nodes %d  rounds %d  tiles %.0f %%  icp %d  sse %.7g  R00 %.7f" % (
        which, os.environ.get('GOICP_PROBE_KW', ''), mse, os.environ.get("GOICP_LANES", "default"), os.environ.get("GOICP_LANE_MIN", "default"), os.environ.get("GOICP_LANE_MIN_WORK", "default"), os.environ.get("GOICP_AUTO_LANES", "default"), 1e3 * t, c.cubes, c.rot_pops, c.bounds_launches, 800.0 * c.tile_expansions / max(c.cubes, 1), c.icp_iters,
        eng.get_best_error(), eng.optR[0, 0]), flush=True)
else:
    sets = [a for a in sys.argv[1:] if ":" in a] or ["1:64", "2:64", "0:64"]
    only = [a for a in sys.argv[1:] if ":" not in a]          # workload names (with mse: bunny@1e-4) to restrict the sweep to
    for which, mse, reps in (("bunny", "1e-3", 8), ("skull", "1e-3", 8), ("spanner", "1e-4", 8), ("s1", "1e-4", 8), ("bunny10", "3e-5", 3), ("bunny", "1e-4", 4), ("s1", "3e-5", 3), ("bunny", "3e-5", 2)):
        if only and ("%s@%s" % (which, mse)) not in only:
            continue
        for st in sets:
            lanes, mn, *work = st.split(":")
            env = dict(os.environ); env["GOICP_LANES"] = lanes; env["GOICP_LANE_MIN"] = mn
            if work: env["GOICP_LANE_MIN_WORK"] = work[0]
            if len(work) > 1: env["GOICP_AUTO_LANES"] = work[1]
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", which, mse, str(reps)], env=env)
