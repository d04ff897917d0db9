import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
from conftest import load_pkg, cloud
pkg = load_pkg(); pkg.load_library()
B = pkg.binding
model, data = cloud("model_bunny"), cloud("data_bunny")
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
R = pkg.fgoicp.rodrigues([0.3, -0.2, 0.9]).astype(np.float32)
for n in (1, 4, 48):
  for tw in (1, 0):
    reg = pkg.Registration(model, data, 1e-3, twin_fusion=tw)
    rng = np.random.default_rng(1)
    par = np.array([[-0.5, -0.5, -0.5, 1.0]] + [[-0.5 + 0.25 * int(rng.integers(0, 4)), -0.5 + 0.25 * int(rng.integers(0, 4)), -0.5, 0.25] for _ in range(n - 1)], np.float32)
    out = [np.full(8 * n, -1, np.float32) for _ in range(4)]
    info = (C.c_int32 * 2)()
    B.check(reg._lib.goicp_debug_queue_expand(reg.handle, fp(np.ascontiguousarray(R.reshape(-1))), 5, fp(np.ascontiguousarray(par.reshape(-1))), n, fp(out[0]), fp(out[1]), fp(out[2]), fp(out[3]), info))
    kids = []
    for p in par:
        w = p[3] / 2
        for j in range(8):
            kids.append([p[0] + (j & 1) * w + w / 2, p[1] + (j >> 1 & 1) * w + w / 2, p[2] + (j >> 2 & 1) * w + w / 2, w])
    kids = np.array(kids, np.float32)
    ub0, lb0 = reg.eval_bounds(R, kids, -1)
    ub1, lb1 = reg.eval_bounds(R, kids, 5)
    print("n", n, "twin", tw, "chunks", info[0], "max rel dev ub0 %.2e lb0 %.2e ub1 %.2e lb1 %.2e" % tuple(np.max(np.abs(a - b) / np.maximum(b, 1e-3)) for a, b in ((out[0], ub0), (out[1], lb0), (out[2], ub1), (out[3], lb1))))
    if n <= 4:
        print(np.c_[out[0][:16], ub0[:16], out[2][:16], ub1[:16]])
    reg.close()
