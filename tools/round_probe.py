#!/usr/bin/env python3
"""The early rounds of a rotation batch as a stand-alone batch: R rotations x ALL translation cubes of one level (as the 8 children of
every cube of the level above), both passes.  What the bound evaluation costs on such a batch through the operator API (search-order items,
bounds_shape's launch shape).  The footprint-ordered form of DESIGN 3.8 was first measured with this batch (an experimental hook in
launch_bounds, since removed: 230 rotations, children of level 2 1.22 -> 1.04 ms, of level 3 8.87 -> 6.81 ms); it now lives in the
device-queue search only (launch_queue_sort).  usage: python3 tools/round_probe.py [nrot=64] [levels=1,2,3] [iters=5]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _pkg  # noqa: E402

pkg = _pkg()
pkg.load_library()
from cuda_go_icp_amd import binding as B  # noqa: E402

nrot = int(sys.argv[1]) if len(sys.argv) > 1 else 64
levels = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3").split(",")]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
g = os.path.join(ROOT, "tests", "golden")
model = np.fromfile(os.path.join(g, "model_bunny.f32"), dtype="<f4").reshape(-1, 3)
data = np.fromfile(os.path.join(g, "data_bunny.f32"), dtype="<f4").reshape(-1, 3)
reg = pkg.Registration(model, data, 1e-3)
lib, h = reg._lib, reg.handle
dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
rv = []
while len(rv) < nrot:
    v = rng.uniform(-np.pi, np.pi, 3)
    if np.linalg.norm(v) <= np.pi:
        rv.append(v)
rots = np.stack([pkg.fgoicp.rodrigues(v) for v in rv]).astype(np.float32)
d_rots = torch.from_numpy(rots.reshape(-1)).to(dev)
coeff = float(reg.rot_coeff(3))
dtype = [("tx", "<f4"), ("ty", "<f4"), ("tz", "<f4"), ("delta", "<f4"), ("coeff", "<f4"), ("rot", "<i4")]
for L in levels:
    # parents: all cubes of level L-1 (width 1/2^(L-1)); children: width w = 1/2^L, centre = corner + bit*w + w/2
    P = 1 << (L - 1)
    wp = np.float32(1.0 / P)
    w = np.float32(wp / 2)
    idx = np.stack(np.meshgrid(np.arange(P), np.arange(P), np.arange(P), indexing="ij"), -1).reshape(-1, 3)
    corner = (np.float32(-0.5) + idx.astype(np.float32) * wp).astype(np.float32)
    recs = np.zeros((nrot, 2, len(corner), 8), dtype=dtype)
    for c in range(8):
        bit = np.array([c & 1, (c >> 1) & 1, (c >> 2) & 1], np.float32)
        cen = corner + bit * w + w / np.float32(2)
        recs["tx"][:, :, :, c], recs["ty"][:, :, :, c], recs["tz"][:, :, :, c] = cen[:, 0], cen[:, 1], cen[:, 2]
    recs["delta"] = np.float32(lib.goicp_trans_delta(float(w)))
    recs["coeff"][:, 1] = coeff
    recs["rot"] = np.arange(nrot, dtype=np.int32)[:, None, None, None]
    flat = np.ascontiguousarray(recs.reshape(-1))
    d_cubes = torch.from_numpy(flat.view(np.uint8).reshape(-1)).to(dev)
    d_ub = torch.empty(len(flat), dtype=torch.float32, device=dev)
    d_lb = torch.empty(len(flat), dtype=torch.float32, device=dev)
    ms = C.c_float()
    B.check(lib.goicp_time_bounds_device(h, d_rots.data_ptr(), d_cubes.data_ptr(), len(flat), d_ub.data_ptr(), d_lb.data_ptr(), iters, C.byref(ms)))
    ub, lb = d_ub.cpu().numpy().astype(np.float64), d_lb.cpu().numpy().astype(np.float64)
    print("children of level %d: %d rotations x 2 passes x %d expansions = %d cube bounds: %.3f ms per launch = %.1f M cube bounds/s  (sum ub %.9g, sum lb %.9g, ub[7] %.8g)"
          % (L, nrot, len(corner), len(flat), ms.value, len(flat) / ms.value / 1e3, ub.sum(), lb.sum(), ub[7]), flush=True)
reg.close()
