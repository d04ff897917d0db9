"""Exploration on the GPU box (not a test): spanner and S2 registrations at several thresholds."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_pkg, cloud, rot_angle
pkg = load_pkg(); pkg.load_library()
which = sys.argv[1] if len(sys.argv) > 1 else "spanner"
if which == "spanner":
    target, source = cloud("spanner_target"), cloud("spanner_source")
    s, t = source.astype(np.float64), target.astype(np.float64)
    ms, mt = s.mean(0), t.mean(0)
    U, _, Vt = np.linalg.svd((s - ms).T @ (t - mt))
    Rgt = Vt.T @ np.diag([1, 1, np.linalg.det(Vt.T @ U.T)]) @ U.T
    tgt = mt - Rgt @ ms
    for mse in (1e-4,):
        t0 = time.time(); eng = pkg.FastGoICP(target, source, mse, verbose=1); t1 = time.time()
        print("sse at GT pose:", eng.registration.compute_sse_error(Rgt, tgt), "thr", eng.sse_threshold, flush=True)
        eng.run(); t2 = time.time()
        c = eng.counters
        print("spanner mse %g: create %.2fs run %.3fs sse %.4f rot_err %.5f t_err %.5f rot_pops %d cubes %d icp %d launches %d" % (
            mse, t1 - t0, t2 - t1, eng.get_best_error(), rot_angle(eng.optR, Rgt), np.linalg.norm(eng.optT - tgt), c.rot_pops, c.cubes, c.icp_iters, c.bounds_launches), flush=True)
        eng.registration.close()
elif which == "ksweep":
    for K in (4, 8, 16, 32):
        for rb in (64,):
            best = None
            for rep in range(3):
                eng = pkg.FastGoICP(cloud("model_bunny"), cloud("data_bunny"), 1e-3, trans_batch=K, rot_batch=rb)
                t1 = time.time(); eng.run(); dt = time.time() - t1; c = eng.counters
                best = min(best, dt) if best else dt
                eng.registration.close()
            print("K=%d rot_batch=%d: best %.4fs sse %.4f trans_pops %d cubes %d launches %d" % (K, rb, best, eng.get_best_error() if False else 0, c.trans_pops, c.cubes, c.bounds_launches), flush=True)
elif which == "ramp":
    for ramp in (4, 8, 16, 32, 64):
        for rb in (32, 64, 128, 256):
            os.environ["GOICP_RAMP"] = str(ramp)
            best = None
            for rep in range(3):
                eng = pkg.FastGoICP(cloud("model_bunny"), cloud("data_bunny"), 1e-3, rot_batch=rb)
                t1 = time.time(); eng.run(); dt = time.time() - t1; c = eng.counters
                best = min(best, dt) if best else dt
                eng.registration.close()
            print("ramp=%d rot_batch=%d: best %.4fs rot_pops %d cubes %d launches %d icp %d" % (ramp, rb, best, c.rot_pops, c.cubes, c.bounds_launches, c.icp_iters), flush=True)
elif which == "bunny":
    for dq in (1, 0, 1, 0):
        eng = pkg.FastGoICP(cloud("model_bunny"), cloud("data_bunny"), 1e-3, verbose=1, device_queues=dq)
        t1 = time.time(); eng.run(); c = eng.counters
        print("bunny device_queues=%d run %.4fs sse %.4f rot_pops %d trans_pops %d cubes %d launches %d icp %d" % (
            dq, time.time() - t1, eng.get_best_error(), c.rot_pops, c.trans_pops, c.cubes, c.bounds_launches, c.icp_iters), flush=True)
        eng.registration.close()
else:
    from cuda_go_icp_amd import synth
    amp = float(os.environ.get("AMP", "0.35"))
    nn = int(os.environ.get("NPTS", "1000000"))
    target, source, Rgt, tgt = synth.make_pair(seed=synth.S2["seed"], M=nn, N=nn, amp=amp)
    print("amp", amp, "N", nn, flush=True)
    probe = pkg.Registration(target, source, 1e-3, dt_size=512)
    floor = float(probe.compute_sse_error(Rgt, tgt)) / len(source)
    probe.close()
    print("floor mse at the GT pose", floor, flush=True)
    for mse in [float(x) * (floor if float(x) >= 1 else 1) for x in sys.argv[2:]] or [1e-4, 3e-5, 2e-5, 1.5e-5]:
        t0 = time.time(); eng = pkg.FastGoICP(target, source, mse, dt_size=512, verbose=1); t1 = time.time()
        print("sse at GT pose:", eng.registration.compute_sse_error(Rgt, tgt), "thr", eng.sse_threshold, flush=True)
        eng.run(); t2 = time.time()
        c = eng.counters
        print("S2 mse %g: create %.2fs run %.3fs sse %.4f rot_err %.5f t_err %.5f rot_pops %d cubes %d icp %d launches %d" % (
            mse, t1 - t0, t2 - t1, eng.get_best_error(), rot_angle(eng.optR, Rgt), np.linalg.norm(eng.optT - tgt), c.rot_pops, c.cubes, c.icp_iters, c.bounds_launches), flush=True)
        eng.registration.close()
