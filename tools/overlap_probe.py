#!/usr/bin/env python3
"""Can the ICP chain of a registration be hidden behind bound evaluation on a second stream?  (VERDICT r3 #2.)

Before building the asynchronous refinement, the thing it rests on is measured directly: how the two kernel families co-schedule on
one MI355X.  Two engines on the same GPU, each with its own HIP stream, driven by two host threads:

  A  the ICP loop of the bunny registration (goicp_icp_run, 200 forced iterations from the identity pose: pass + finalize per iteration),
     stream priority default and highest;
  B  the bound evaluation as the search issues it: launches of the sibling-structured batch of E expansions (E = 256 / 2 048 / 8 192:
     the small, typical and large rounds of a rotation batch), back to back on its stream.

Reported: ICP iterations/s and cube bounds/s alone and side by side, and what an overlapped schedule of the bunny registration's
13.5 ms ICP + 15.8 ms bound evaluation could therefore gain at best:  T_serial = T_icp + T_bnb  against  T_overlap = max over the two
of (their time at the side-by-side rates).  usage (GPU box): python tools/overlap_probe.py > gpurun_out/overlap_probe.json
"""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from __graft_entry__ import _pkg
    pkg = _pkg()
    pkg.load_library()
    from cuda_go_icp_amd import binding as B
    import bench
    g = os.path.join(ROOT, "tests", "golden")
    ld = lambda n: np.fromfile(os.path.join(g, n + ".f32"), dtype="<f4").reshape(-1, 3)
    model, data = ld("model_bunny"), ld("data_bunny")
    dev = torch.device("cuda", 0)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    out = {"tool": "tools/overlap_probe.py", "workload": "bunny N=30379 M=35947 DT 300^3, two engines on one GPU", "runs": []}
    short = "--short" in sys.argv          # under rocprofv3 --kernel-trace: one configuration, 50 ms legs (tools/overlap_timeline.py reads the trace)
    for prio in ((1,) if short else (0, 1)):
        A = pkg.Registration(model, data, 1e-3, stream_priority=prio)
        Bn = pkg.Registration(model, data, 1e-3)
        lib = A._lib
        for E in ((2048,) if short else (256, 2048, 8192)):
            rots, recs, _ = bench.make_batch(pkg, Bn, E, 8, seed=7)
            Bc = len(recs)
            d_rots = torch.from_numpy(rots.reshape(-1)).to(dev)
            d_cubes = torch.from_numpy(recs.view(np.uint8).reshape(-1)).to(dev)
            d_ub = torch.empty(Bc, dtype=torch.float32, device=dev)
            d_lb = torch.empty(Bc, dtype=torch.float32, device=dev)
            stop = threading.Event()
            res = {}

            def icp_loop(tag, seconds):
                n, t0 = 0, time.perf_counter()
                while time.perf_counter() - t0 < seconds:
                    Ri, ti = np.eye(3, dtype=np.float32).reshape(9).copy(), np.zeros(3, np.float32)
                    err, it = C.c_float(), C.c_int32()
                    B.check(lib.goicp_icp_run(A.handle, fp(Ri), fp(ti), 200, -1e30, C.byref(err), C.byref(it)))
                    n += it.value
                res[tag] = n / (time.perf_counter() - t0)

            def bounds_loop(tag, seconds=None):
                ms = C.c_float()
                n, t0 = 0, time.perf_counter()
                while (time.perf_counter() - t0 < seconds) if seconds else not stop.is_set():
                    B.check(lib.goicp_time_bounds_device(Bn.handle, d_rots.data_ptr(), d_cubes.data_ptr(), Bc, d_ub.data_ptr(), d_lb.data_ptr(), 20, C.byref(ms)))
                    n += 20 * Bc
                res[tag] = n / (time.perf_counter() - t0)

            icp_loop("icp_alone", 0.02 if short else 0.5)
            bounds_loop("bounds_alone", 0.02 if short else 0.5)
            tb = threading.Thread(target=bounds_loop, args=("bounds_beside",))
            tb.start()
            time.sleep(0.05)
            icp_loop("icp_beside", 0.05 if short else 1.0)
            stop.set()
            tb.join()
            # the bunny registration's two stretches: 470 ICP iterations, 466.7 k cube bounds of bound evaluation at 34 ns each
            t_icp, t_bnb = 470 / res["icp_alone"], 15.8e-3
            t_icp_b = 470 / res["icp_beside"]
            slow_b = res["bounds_alone"] / res["bounds_beside"]
            # overlapped: the ICP runs for t_icp_b; the bound evaluation that runs beside it proceeds at 1 / slow_b of its rate, the rest alone
            done_beside = min(t_bnb, t_icp_b / slow_b)
            t_overlap = t_icp_b + (t_bnb - done_beside)
            r = {"icp_stream_priority": "highest" if prio else "default", "expansions_per_launch": E,
                 "icp_iters_per_s_alone": round(res["icp_alone"], 1), "icp_iters_per_s_beside_bounds": round(res["icp_beside"], 1),
                 "cube_bounds_per_s_alone": round(res["bounds_alone"], 1), "cube_bounds_per_s_beside_icp": round(res["bounds_beside"], 1),
                 "icp_slowdown": round(res["icp_alone"] / res["icp_beside"], 3), "bounds_slowdown": round(slow_b, 3),
                 "registration_model_ms": {"serial": round(1e3 * (t_icp + t_bnb), 2), "overlapped_at_measured_rates": round(1e3 * t_overlap, 2),
                                           "is": "470 ICP iterations + 15.8 ms of bound evaluation (full bunny); overlapped = every ICP iteration beside bound evaluation that is useful "
                                                 "(an upper bound on the gain: the evaluation beside a refinement runs against the unrefined incumbent and expands more)"}}
            out["runs"].append(r)
            print(json.dumps(r), file=sys.stderr)
        A.close(); Bn.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
