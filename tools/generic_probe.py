#!/usr/bin/env python3
"""SURVEY 8(d)'s microbench batch (65 536 unrelated cubes, 8 rotations): as given (a random rotation per cube), grouped by rotation,
and grouped by rotation + sorted by translation (Morton) -- what ordering alone is worth to the generic path of bounds_kernel."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from __graft_entry__ import _pkg  # noqa: E402

pkg = _pkg()
pkg.load_library()
from cuda_go_icp_amd import binding as B  # noqa: E402
g = os.path.join(ROOT, "tests", "golden")
model = np.fromfile(os.path.join(g, "model_bunny.f32"), dtype="<f4").reshape(-1, 3)
data = np.fromfile(os.path.join(g, "data_bunny.f32"), dtype="<f4").reshape(-1, 3)
reg = pkg.Registration(model, data, 1e-3)
lib, h = reg._lib, reg.handle
dev = torch.device("cuda", 0)
rots, recs, _ = bench.make_generic_batch(pkg, reg, 65536, 8, seed=99)
d_rots = torch.from_numpy(rots.reshape(-1)).to(dev)
d_ub = torch.empty(len(recs), dtype=torch.float32, device=dev)
d_lb = torch.empty(len(recs), dtype=torch.float32, device=dev)


def morton(c):
    q = np.clip(((c + 0.5) * 1024).astype(np.int64), 0, 1023)
    out = np.zeros(len(c), np.int64)
    for b in range(10):
        for k in range(3):
            out |= ((q[:, k] >> b) & 1) << (3 * b + k)
    return out


cen = np.stack([recs["tx"], recs["ty"], recs["tz"]], 1)
orders = {"as given": np.arange(len(recs)), "grouped by rotation": np.argsort(recs["rot"], kind="stable"),
          "grouped by rotation and pass": np.lexsort((recs["coeff"] > 0, recs["rot"])),
          "rotation, pass, Morton order of the translation": np.lexsort((morton(cen), recs["coeff"] > 0, recs["rot"]))}
ref = None
for name, o in orders.items():
    r = np.ascontiguousarray(recs[o])
    d_cubes = torch.from_numpy(r.view(np.uint8).reshape(-1)).to(dev)
    ms = C.c_float()
    B.check(lib.goicp_time_bounds_device(h, d_rots.data_ptr(), d_cubes.data_ptr(), len(r), d_ub.data_ptr(), d_lb.data_ptr(), 10, C.byref(ms)))
    ub = np.empty(len(r), np.float32)
    ub[o] = d_ub.cpu().numpy()
    if ref is None:
        ref = ub
    print("%-48s %.3f ms per launch = %.1f M cube bounds/s; same bounds: %s" % (name, ms.value, len(r) / ms.value / 1e3, bool(np.array_equal(ub, ref))), flush=True)
reg.close()
