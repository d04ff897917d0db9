#!/usr/bin/env python3
"""(Runs at commit d43fdcd only: the variant was measured slower and removed -- EXPERIMENTS.md R4.17.)
Eight queries per wavefront in the ICP pass (8-lane half rows; env GOICP_ICP_ROWS8 = 1, the default) against four (16-lane rows; = 0), one
process per setting: ICP iterations/s over a forced 200-iteration trajectory from the identity pose, one iteration frozen at the converged pose,
the registration -- and the bits of the poses, which must not differ (the accumulators receive the same integers)."""
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
    which = sys.argv[2]
    dt_size = 300
    if which == "bunny":
        model, data, mse = cloud("model_bunny"), cloud("data_bunny"), 1e-3
    elif which == "bunny10":
        model, data, mse = cloud("model_bunny"), cloud("data_bunny", 10), 1e-3
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
    reg = pkg.Registration(model, data, mse, dt_size=dt_size)
    lib, h = reg._lib, reg.handle
    R, t = np.eye(3, dtype=np.float32).reshape(9).copy(), np.zeros(3, np.float32)
    err, it = C.c_float(), C.c_int32()
    B.check(lib.goicp_icp_run(h, fp(R), fp(t), 5, -1e30, C.byref(err), C.byref(it)))
    t1 = time.perf_counter()
    B.check(lib.goicp_icp_run(h, fp(R), fp(t), 200, -1e30, C.byref(err), C.byref(it)))
    rate = it.value / (time.perf_counter() - t1)
    bits = R.tobytes().hex() + t.tobytes().hex()
    R2, t2 = np.eye(3, dtype=np.float32).reshape(9).copy(), np.zeros(3, np.float32)
    B.check(lib.goicp_icp_run(h, fp(R2), fp(t2), 10000, mse / 1e4, C.byref(err), C.byref(it)))
    conv_it = it.value
    ms = C.c_float()
    B.check(lib.goicp_time_icp_pass(h, fp(R2), fp(t2), 50, C.byref(ms)))
    reg.close()
    eng = pkg.FastGoICP(model, data, mse, dt_size=dt_size)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        eng.run()
        ts.append(time.perf_counter() - t0)
    c = eng.counters
    import hashlib
    print("%-8s N %7d M %7d rows8 %s: %9.1f iterations/s (forced 200; pose sha %s)  converging run %d iterations (err %.7g)  iteration at the converged pose %.1f us  registration median %.2f ms  icp iters %d  sse %.7g R sha %s" % (
        which, len(data), len(model), os.environ.get("GOICP_ICP_ROWS8", "default"), rate, hashlib.sha1(bytes.fromhex(bits)).hexdigest()[:10], conv_it, err.value, 1e3 * ms.value,
        1e3 * float(np.median(ts[1:])), c.icp_iters, eng.get_best_error(), hashlib.sha1(eng.optR.tobytes() + eng.optT.tobytes()).hexdigest()[:10]), flush=True)
else:
    for which in (sys.argv[1:] or ["bunny", "bunny10", "skull", "s1", "spanner", "s2"]):
        for mode in ("0", "1"):
            env = dict(os.environ); env["GOICP_ICP_ROWS8"] = mode
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", which], env=env)
