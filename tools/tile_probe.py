#!/usr/bin/env python3
"""LDS-staged DT tiles against direct gathers on batches shaped like the deep rounds of inner searches (GPU box).
A segment = one search: one rotation, n = 64 translation nodes of depth d forming a 4x4x4 block of neighbours around a
random translation.  Prints per depth: the two kernels' time per 65 536 cube bounds, the largest relative difference of
their (ub, lb), and how many 64-point patches could be staged.  usage: python3 tools/tile_probe.py [bunny|s2] [depth,depth,...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _pkg  # noqa: E402

pkg = _pkg()
pkg.load_library()
from cuda_go_icp_amd import binding as B  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "bunny"
DEPTHS = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (4, 5, 6, 7, 8, 10)
if which == "bunny":
    g = os.path.join(ROOT, "tests", "golden")
    tg = np.fromfile(os.path.join(g, "model_bunny.f32"), dtype="<f4").reshape(-1, 3)
    sr = np.fromfile(os.path.join(g, "data_bunny.f32"), dtype="<f4").reshape(-1, 3)
    V = 300
else:
    from cuda_go_icp_amd import synth
    tg, sr, _, _ = synth.make_pair(seed=synth.S2["seed"], M=1000000, N=1000000)
    V = 512
reg = pkg.Registration(tg, sr, 1e-3, dt_size=V)
lib, h = reg._lib, reg.handle
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
rng = np.random.default_rng(7)
nseg, n = 128, 64
for chunks in ((16, 32) if which == "bunny" else (64,)):
    for depth in DEPTHS:
        w = np.float32(1.0 / (1 << depth))
        rots = np.stack([pkg.fgoicp.rodrigues(rng.uniform(-2.0, 2.0, 3)) for _ in range(nseg)]).astype(np.float32).reshape(-1)
        par = np.zeros((nseg, n, 4), np.float32)
        for i in range(nseg):
            c0 = (np.floor(rng.uniform(-0.3, 0.3, 3) / w) * w).astype(np.float32)
            k = 0
            for a in range(4):
                for b in range(4):
                    for c in range(4):
                        par[i, k] = (c0[0] + a * w, c0[1] + b * w, c0[2] + c * w, w); k += 1
        Bc = nseg * n * 8
        out = [np.zeros(Bc, np.float32) for _ in range(4)]
        ms = (C.c_float * 2)(); st = (C.c_uint32 * 2)()
        B.check(lib.goicp_debug_bounds_tile(h, fp(rots), fp(par.reshape(-1)), nseg, n, 6, chunks, fp(out[0]), fp(out[1]), fp(out[2]), fp(out[3]), ms, st))
        rel = max(np.max(np.abs(out[0] - out[2]) / np.maximum(np.abs(out[2]), 1e-6)), np.max(np.abs(out[1] - out[3]) / np.maximum(np.abs(out[3]), 1e-6)))
        print("%s chunks %3d depth %2d (child width %.5f = %.2f voxels): tile %.3f ms, direct %.3f ms -> %.2fx; max rel diff %.2e; patches staged %d / too large %d" % (
            which, chunks, depth, w / 2, float(w) / 2 * V / (2 * float(np.abs(tg).max()) * 1.0 + 1e-9) if False else float(w) / 2, ms[0], ms[1], ms[1] / ms[0], rel, st[0], st[1]), flush=True)
reg.close()
