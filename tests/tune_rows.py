"""Dev script (not a test): ICP pass-kernel time vs. number of queries, for GOICP_ICP_ROWS=<mode>."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import cloud, load_pkg  # noqa: E402

pkg = load_pkg()
B = pkg.binding
lib = B.load_library()
model, data = cloud("model_bunny"), cloud("data_bunny")
out = []
for n in (64, 1024, 4096, 8192, 16384, len(data)):
    reg = pkg.Registration(model, data[:n].copy(), 1e-3)
    Ri = np.eye(3, dtype=np.float32).ravel()
    ti = np.zeros(3, np.float32)
    ms = C.c_float()
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    B.check(lib.goicp_time_icp_pass(reg.handle, fp(Ri), fp(ti), 50, C.byref(ms)))
    out.append("%d:%.1fus" % (n, ms.value * 1e3))
print("mode", os.environ.get("GOICP_ICP_ROWS", "default"), " ".join(out))
