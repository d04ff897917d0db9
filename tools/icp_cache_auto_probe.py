#!/usr/bin/env python3
"""The exact walk-skipping neighbour cache switched on in the TAIL of an ICP run only (icp_nn_cache = 2; env GOICP_ICP_CACHE_REL = the error's
relative decrease per 16-iteration chunk under which it is switched on): ICP iterations/s of the bench's trajectory, the registration, and
bit-identical results against icp_nn_cache = 0.  One process per setting."""
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
    if which == "bunny":
        model, data, mse = cloud("model_bunny"), cloud("data_bunny"), 1e-3
    elif which == "skull":
        model, data, _, _ = skull_problem()
        mse = 1e-3
    elif which == "spanner":
        model, data, mse = cloud("spanner_target"), cloud("spanner_source"), 1e-4
    else:
        from cuda_go_icp_amd import synth
        model, data, _, _ = synth.make_pair(**{k: synth.S1[k] for k in ("seed", "M", "N")})
        mse = 1e-4
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    reg = pkg.Registration(model, data, mse, icp_nn_cache=mode)
    lib, h = reg._lib, reg.handle
    R, t = np.eye(3, dtype=np.float32).reshape(9).copy(), np.zeros(3, np.float32)
    err, it = C.c_float(), C.c_int32()
    B.check(lib.goicp_icp_run(h, fp(R), fp(t), 5, -1e30, C.byref(err), C.byref(it)))
    t1 = time.perf_counter()
    B.check(lib.goicp_icp_run(h, fp(R), fp(t), 200, -1e30, C.byref(err), C.byref(it)))
    rate = it.value / (time.perf_counter() - t1)
    # a converging run as the registration makes it (the reference's stop rule)
    R2, t2 = np.eye(3, dtype=np.float32).reshape(9).copy(), np.zeros(3, np.float32)
    t1 = time.perf_counter()
    B.check(lib.goicp_icp_run(h, fp(R2), fp(t2), 10000, mse / 1e4, C.byref(err), C.byref(it)))
    conv_ms, conv_it = 1e3 * (time.perf_counter() - t1), it.value
    reg.close()
    eng = pkg.FastGoICP(model, data, mse, icp_nn_cache=mode)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        eng.run()
        ts.append(time.perf_counter() - t0)
    c = eng.counters
    print("%-8s icp_nn_cache %d rel %s: %8.1f iterations/s (forced 200)  converging run %d iterations %.2f ms  registration median %.2f ms  icp iters %d  sse %.7g  R %s t %s err %.7g" % (
        which, mode, os.environ.get("GOICP_ICP_CACHE_REL", "default"), rate, conv_it, conv_ms, 1e3 * float(np.median(ts[1:])), c.icp_iters, eng.get_best_error(),
        R2[:3].tobytes().hex()[:12], t2.tobytes().hex()[:12], err.value), flush=True)
else:
    for which in ("bunny", "s1", "skull", "spanner"):
        for mode, rel in ((0, ""), (2, "0.005"), (2, "0.02"), (2, "0.05"), (2, "0.2"), (1, "")):
            env = dict(os.environ)
            if rel:
                env["GOICP_ICP_CACHE_REL"] = rel
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", which, str(mode)], env=env)
