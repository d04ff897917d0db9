"""Developer tuning script (not a test): time one ICP correspondence pass at two poses."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from __graft_entry__ import _pkg
    pkg = _pkg()
    from cuda_go_icp_amd import binding as B
    g = os.path.join(ROOT, "tests", "golden")
    model = np.fromfile(os.path.join(g, "model_bunny.f32"), dtype="<f4").reshape(-1, 3)
    data = np.fromfile(os.path.join(g, "data_bunny.f32"), dtype="<f4").reshape(-1, 3)
    reg = pkg.Registration(model, data, 1e-3, dt_size=64, kd_gpu_build=int(os.environ.get("GOICP_TUNE_GPU_BUILD", "0")))
    print("kd_gpu_build =", reg.params.kd_gpu_build, flush=True)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    for name, R, t in (("identity", np.eye(3, dtype=np.float32).reshape(9), np.zeros(3, np.float32)),
                       ("converged", None, None)):
        if R is None:
            err, R, t = pkg.IterativeClosestPoint3D(reg, 300, 1e-7).run()
            R, t = np.ascontiguousarray(R.reshape(9)), np.ascontiguousarray(t)
        ms = C.c_float()
        B.check(reg._lib.goicp_time_icp_pass(reg.handle, fp(R), fp(t), 20, C.byref(ms)))
        print("pose=%s: %.1f us per pass (kernel + finalize)" % (name, 1e3 * ms.value), flush=True)
    import time
    icp = pkg.IterativeClosestPoint3D(reg, 400, -1e30)
    icp.run()
    t0 = time.perf_counter(); icp = pkg.IterativeClosestPoint3D(reg, 400, -1e30); icp.run(); el = time.perf_counter() - t0
    print("full loop: %.1f us per iteration (400 forced iterations from identity)" % (1e6 * el / 400), flush=True)
else:
    for gb in ("0", "1"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, GOICP_TUNE_GPU_BUILD=gb))
