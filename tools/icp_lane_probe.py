#!/usr/bin/env python3
"""(Runs at commit 994fba6 only: the lane walker was measured slower and removed -- EXPERIMENTS.md R4.15.)
The one-query-per-lane ICP pass against the cooperative one (goicp_params.icp_lane_walk 0 / 1; one process per setting): ICP iterations/s over a forced
200-iteration trajectory from the identity pose, a converging run, the registration, and the bits of the results (they must not differ)."""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import ctypes as C
    import time

    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from conftest import cloud, load_pkg, skull_problem
    pkg = load_pkg()
    from cuda_go_icp_amd import binding as B
    which, mode = sys.argv[2], int(sys.argv[3])
    dt_size = 300
    if which == "bunny":
        model, data, mse = cloud("model_bunny"), cloud("data_bunny"), 1e-3
    elif which == "skull":
        model, data, _, _ = skull_problem()
        mse = 1e-3
    elif which == "spanner":
        model, data, mse = cloud("spanner_target"), cloud("spanner_source"), 1e-4
    elif which == "s2":
        from cuda_go_icp_amd import synth
        model, data, _, _ = synth.make_pair(seed=synth.S2["seed"], M=synth.S2["M"], N=synth.S2["N"], amp=0.15)
        mse, dt_size = 7.3e-6, 512
    else:
        from cuda_go_icp_amd import synth
        model, data, _, _ = synth.make_pair(**{k: synth.S1[k] for k in ("seed", "M", "N")})
        mse = 1e-4
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    reg = pkg.Registration(model, data, mse, icp_lane_walk=mode, dt_size=dt_size)
    lib, h = reg._lib, reg.handle
    R, t = np.eye(3, dtype=np.float32).reshape(9).copy(), np.zeros(3, np.float32)
    err, it = C.c_float(), C.c_int32()
    B.check(lib.goicp_icp_run(h, fp(R), fp(t), 5, -1e30, C.byref(err), C.byref(it)))
    t1 = time.perf_counter()
    B.check(lib.goicp_icp_run(h, fp(R), fp(t), 200, -1e30, C.byref(err), C.byref(it)))
    rate = it.value / (time.perf_counter() - t1)
    bits = R.tobytes().hex()[:16] + t.tobytes().hex()[:8]
    R2, t2 = np.eye(3, dtype=np.float32).reshape(9).copy(), np.zeros(3, np.float32)
    t1 = time.perf_counter()
    B.check(lib.goicp_icp_run(h, fp(R2), fp(t2), 10000, mse / 1e4, C.byref(err), C.byref(it)))
    conv_ms, conv_it = 1e3 * (time.perf_counter() - t1), it.value
    ms = C.c_float()
    B.check(lib.goicp_time_icp_pass(h, fp(R2), fp(t2), 50, C.byref(ms)))          # frozen at the converged pose: every query next to the surface
    frozen_us = 1e3 * ms.value
    reg.close()
    eng = pkg.FastGoICP(model, data, mse, icp_lane_walk=mode, dt_size=dt_size)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        eng.run()
        ts.append(time.perf_counter() - t0)
    c = eng.counters
    print("%-8s N %7d M %7d lane walk %2d: %9.1f iterations/s (forced 200; pose bits %s)  converging run %d iterations %.2f ms (err %.7g)  iteration at the converged pose %.1f us  registration median %.2f ms  icp iters %d  sse %.7g" % (
        which, len(data), len(model), mode, rate, bits, conv_it, conv_ms, err.value, frozen_us, 1e3 * float(np.median(ts[1:])), c.icp_iters, eng.get_best_error()), flush=True)
else:
    for which in (sys.argv[1:] or ["spanner", "skull", "s1", "bunny", "s2"]):
        for mode in (0, 1):
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", which, str(mode)])
