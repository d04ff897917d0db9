#!/usr/bin/env python3
"""[runs against commit 4447c64 only: the packet walker (Params::icp_packet) was removed after this measurement -- EXPERIMENTS R4.10]
ICP neighbour search: walks seeded from the distance-transform bound (icp_packet 0: a walk per query) against the PACKET walk (1: four neighbouring queries walk together; was: DT-bound vs point seeds
nearest-target-point table (1): engine creation, iterations/s of one forced 200-iteration trajectory, pass + finalize time at
the identity pose and at the ICP optimum (HIP events), and identical results."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import _pkg  # noqa: E402
from conftest import cloud, skull_problem  # noqa: E402

pkg = _pkg()
pkg.load_library()
from cuda_go_icp_amd import binding as B, synth  # noqa: E402
work = [("bunny 30k/36k", cloud("model_bunny"), cloud("data_bunny"), 300)]
t, s, _, _ = synth.make_pair(**{k: synth.S1[k] for k in ("seed", "M", "N")})
work.append(("S1 40k/40k", t, s, 300))
t, s, _, _ = skull_problem()
work.append(("skull 30k/98k", t, s, 300))
work.append(("spanner 150k/150k", cloud("spanner_target"), cloud("spanner_source"), 300))
if "--s2" in sys.argv:
    t, s, _, _ = synth.make_pair(seed=synth.S2["seed"], M=synth.S2["M"], N=synth.S2["N"])
    work.append(("S2 1M/1M", t, s, 512))
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
for name, tg, sr, V in work:
    res = {}
    for seed in (0, 1):     # icp_packet
        t0 = time.perf_counter()
        reg = pkg.Registration(tg, sr, 1e-3, dt_size=V, icp_packet=seed)
        create = time.perf_counter() - t0
        lib, h = reg._lib, reg.handle
        R, tt = np.eye(3, dtype=np.float32).reshape(9).copy(), np.zeros(3, np.float32)
        err, it, ms0, ms1 = C.c_float(), C.c_int32(), C.c_float(), C.c_float()
        B.check(lib.goicp_time_icp_pass(h, fp(R), fp(tt), 50, C.byref(ms0)))
        B.check(lib.goicp_icp_run(h, fp(R), fp(tt), 5, -1e30, C.byref(err), C.byref(it)))
        t1 = time.perf_counter()
        B.check(lib.goicp_icp_run(h, fp(R), fp(tt), 200, -1e30, C.byref(err), C.byref(it)))
        rate = it.value / (time.perf_counter() - t1)
        B.check(lib.goicp_time_icp_pass(h, fp(R), fp(tt), 50, C.byref(ms1)))
        res[seed] = (R.copy(), tt.copy(), err.value)
        print("%-18s icp_packet %d: create %.3f s  %8.1f iterations/s  pass+finalize %.1f us (identity) / %.1f us (optimum)  err %.6g" % (
            name, seed, create, rate, ms0.value * 1e3, ms1.value * 1e3, err.value), flush=True)
        reg.close()
    # neighbours instead of strangers in a workgroup: the workgroups' float sums are other sums -> last-bit differences of the trajectory
    assert np.abs(res[0][0] - res[1][0]).max() <= 1e-5 and np.abs(res[0][1] - res[1][1]).max() <= 1e-5 and abs(res[0][2] - res[1][2]) <= 1e-5 * res[0][2], "results differ"
    # the NN operator through the packet walker against the per-query walker: bit-equal indices and distances, neighbouring and random queries
    rng = np.random.default_rng(5)
    q = np.concatenate([sr[:4096], rng.uniform(-1.2, 1.2, (4096, 3)).astype(np.float32), tg[:1024]]).astype(np.float32)
    out = {}
    for pk in (0, 1):
        reg = pkg.Registration(tg, sr, 1e-3, dt_size=V, icp_packet=pk)
        out[pk] = reg.nn_query(q)
        reg.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]), "NN operator: packet walk differs"
    print("%-18s NN operator: packet walk == per-query walk on %d queries (indices and squared distances bit-equal)" % (name, len(q)), flush=True)
