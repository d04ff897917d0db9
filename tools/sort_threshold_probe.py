#!/usr/bin/env python3
"""VERDICT r3 #7, one measurement: the footprint sort of the work items for EVERY round of a large rotation batch (threshold 1 .. 2 048
expansions, env GOICP_SORT_MIN_GROUPS) -- does sorting round 1 (230 roots, L2 hit 0.57 in profiles/r03_pmc_e2e_rounds.txt) and the rounds
under 2 048 expansions pay?  Full-bunny registration, median of 7, per threshold; one process per setting (the threshold is read at
engine creation)."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time, numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from conftest import cloud, load_pkg
    pkg = load_pkg()
    model, data = cloud("model_bunny"), cloud("data_bunny")
    mse = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
    if len(sys.argv) > 3 and sys.argv[3] == "s1":
        from cuda_go_icp_amd import synth
        model, data, _, _ = synth.make_pair(**{k: synth.S1[k] for k in ("seed", "M", "N")})
    if len(sys.argv) > 3 and sys.argv[3] == "bunny2":          # every second source point: 15 190 points
        data = np.ascontiguousarray(data[::2])
    eng = pkg.FastGoICP(model, data, mse)
    ts = []
    for _ in range(8 if mse >= 1e-3 else (4 if mse >= 1e-4 else 2)):
        t0 = time.perf_counter(); eng.run(); ts.append(time.perf_counter() - t0)
    c = eng.counters
    print((sys.argv[3] if len(sys.argv) > 3 else "bunny") + " mse %g GOICP_SORT_CHUNK_PTS=%s GOICP_SORT_MIN_GROUPS=%s: register_s median %.4f min %.4f (first run %.4f)  cubes %d" % (mse, os.environ.get("GOICP_SORT_CHUNK_PTS", "default"), os.environ.get("GOICP_SORT_MIN_GROUPS", "default"), float(np.median(ts[1:])), min(ts[1:]), ts[0], c.cubes), flush=True)
else:
    if sys.argv[1:2] == ["chunks2"]:
        for which, mse, cps in (("bunny", "1e-3", ("2304", "2560", "2816")), ("s1", "1e-4", ("2560", "3072", "3584", "4096")), ("s1", "3e-5", ("2560", "3072", "3584", "4096")),
                                ("bunny2", "1e-3", ("1280", "1536", "2048", "2560", "4096"))):
            for cp in cps:
                env = dict(os.environ); env["GOICP_SORT_MIN_GROUPS"] = "256"; env["GOICP_SORT_CHUNK_PTS"] = cp
                subprocess.run([sys.executable, os.path.abspath(__file__), "child", mse, which], env=env)
        sys.exit(0)
    if sys.argv[1:2] == ["chunks"]:
        for mse in ("1e-3", "1e-4"):
            for cp in ("2048", "2560", "3072", "4096"):
                for v in ("128", "256"):
                    env = dict(os.environ); env["GOICP_SORT_MIN_GROUPS"] = v; env["GOICP_SORT_CHUNK_PTS"] = cp
                    subprocess.run([sys.executable, os.path.abspath(__file__), "child", mse], env=env)
        sys.exit(0)
    for mse in (sys.argv[1:] or ["1e-3"]):
        for v in (("", "1", "128", "512", "1024", "4096", "1000000") if mse == "1e-3" else ("128", "256", "2048")):
            env = dict(os.environ)
            if v: env["GOICP_SORT_MIN_GROUPS"] = v
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", mse], env=env)
