"""Exploration on the GPU box (not a test): registrations of the test workloads under the driver options
(`flowsweep`: lock-step / continuous flow x round width on six workloads -- the tables of DESIGN section 4 --, `spanner`, `s2`,
`bunny`, `ksweep`, `ramp`, `flow`, `create`).  usage: python3 tools/tune_r2.py <mode>"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
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
elif which == "create":
    for name in ("bunny", "bunny", "bunny", "s2"):
        if name == "s2":
            from cuda_go_icp_amd import synth
            tg, sr, _, _ = synth.make_pair(seed=synth.S2["seed"], M=1000000, N=1000000, amp=0.15)
            V = 512
        else:
            tg, sr, V = cloud("model_bunny"), cloud("data_bunny"), 300
        t0 = time.time(); reg = pkg.Registration(tg, sr, 1e-3, dt_size=V, verbose=1); t1 = time.time()
        print("%s create %.4fs" % (name, t1 - t0), flush=True)
        t0 = time.time(); reg.close(); print("close %.4fs" % (time.time() - t0), flush=True)
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
elif which == "flowsweep":
    from cuda_go_icp_amd import synth
    cases = {"bunny": (cloud("model_bunny"), cloud("data_bunny"), 1e-3, 300), "bunny10": (cloud("model_bunny"), cloud("data_bunny", 10), 1e-3, 300),
             "spanner": (cloud("spanner_target"), cloud("spanner_source"), 1e-4, 300), "s1": synth.make_pair(**{k: synth.S1[k] for k in ("seed", "M", "N")})[:2] + (1e-4, 300)}
    sk = cloud("skull_scan"); rng = np.random.default_rng(1234); sub = sk[rng.random(len(sk)) < 0.3].astype(np.float64)
    cx, sx, cy, sy, cz, sz = np.cos(1.3), np.sin(1.3), np.cos(-0.7), np.sin(-0.7), np.cos(2.1), np.sin(2.1)
    Rg = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    cases["skull"] = (sk, ((sub - np.array([0.15, -0.10, 0.05])) @ Rg + rng.normal(scale=1e-3, size=sub.shape)).astype(np.float32), 1e-3, 300)
    t2, s2, _, _ = synth.make_pair(seed=synth.S2["seed"], M=1000000, N=1000000, amp=0.15)
    cases["s2amp.15"] = (t2, s2, 1.2 * 6.05e-6, 512)
    for name, (tg, sr, mse, V) in cases.items():
        for fl, ak in ((0, 0), (0, 1), (4, 1), (48, 1)):
            best = None
            for rep in range(3):
                eng = pkg.FastGoICP(tg, sr, mse, flow=fl, dt_size=V, adaptive_k=ak)
                t1 = time.time(); eng.run(); dt = time.time() - t1; c = eng.counters
                best = min(best, dt) if best else dt
                sse = eng.get_best_error(); eng.registration.close()
            print("%-9s flow=%-3d ak=%d best %.4fs sse %.4f rot_pops %d cubes %d rounds %d icp %d fallbacks %d" % (name, fl, ak, best, sse, c.rot_pops, c.cubes, c.bounds_launches, c.icp_iters, c.queue_fallbacks), flush=True)
elif which == "flow":
    for fl in (1, 0, 1, 0):
        for K in (32, 16):
            eng = pkg.FastGoICP(cloud("model_bunny"), cloud("data_bunny"), 1e-3, verbose=1 if K == 32 else 0, flow=fl, trans_batch=K)
            t1 = time.time(); eng.run(); c = eng.counters
            print("bunny flow=%d K=%d run %.4fs sse %.4f rot_pops %d trans_pops %d cubes %d launches %d icp %d" % (
                fl, K, time.time() - t1, eng.get_best_error(), c.rot_pops, c.trans_pops, c.cubes, c.bounds_launches, c.icp_iters), flush=True)
            eng.registration.close()
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
