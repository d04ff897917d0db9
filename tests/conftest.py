"""pytest plumbing: markers, paths, shared fixtures.

`-m "not gpu"` tests run in the build container (no GPU): oracle vs golden vectors, host logic, C-ABI
symbol table.  `-m gpu` tests are the parity tests proper and call the HIP path through the C-ABI.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


def load_pkg():
    """Import cuda-go-icp_amd/ (the hyphen keeps it from being a normal import name)."""
    name = "cuda_go_icp_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "cuda-go-icp_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def cloud(name, stride=1):
    a = np.fromfile(os.path.join(GOLDEN, name + ".f32"), dtype="<f4").reshape(-1, 3)
    return np.ascontiguousarray(a[::stride])


def skull_problem():
    """BASELINE configs[2] as SURVEY 8d builds it: target = the reference's data_skull.ply x 0.01 (98 359 points, committed
    blob), source = a seeded 30 % subsample of it under the rigid motion Rz(2.1) Ry(-0.7) Rx(1.3), t = (0.15, -0.10, 0.05),
    + N(0, 1e-3) noise.  Returns (target, source, Rgt, tgt) with target ~= Rgt source + tgt.  Shared by the GPU test and by
    oracle/gen_golden.py (which feeds every 10th source point to the reference's own GoICP::Register)."""
    target = cloud("skull_scan")
    rng = np.random.default_rng(1234)
    sub = target[rng.random(len(target)) < 0.3].astype(np.float64)
    cx, sx, cy, sy, cz, sz = np.cos(1.3), np.sin(1.3), np.cos(-0.7), np.sin(-0.7), np.cos(2.1), np.sin(2.1)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Rgt, tgt = Rz @ Ry @ Rx, np.array([0.15, -0.10, 0.05])
    source = ((sub - tgt) @ Rgt + rng.normal(scale=1e-3, size=sub.shape)).astype(np.float32)   # target ~= Rgt source + tgt
    return target, source, Rgt, tgt


def small_problem(seed):
    """Seeded small registration problem (tests/golden/e2e_small<seed>.json; oracle/gen_golden.py --small-e2e feeds the same clouds to the
    reference's own GoICP::Register): target = 400 points, source = 150 points of a seeded star-shaped surface (cuda-go-icp_amd/synth.py),
    the source under a seeded rigid motion (rotation anywhere in the pi-ball, translation within 0.25 per axis) + N(0, 0.004^2) noise.
    The sparse target leaves the optimum's SSE (0.5-0.7) ABOVE SSEThresh (mse 1e-3 x 150 = 0.15): no early exit -- the reference has to
    prove the optimum (7-18 k rotation nodes, 9-19 M translation nodes, 10-21 minutes of CPU).  Returns (target, source, R_gt, t_gt)."""
    load_pkg()
    from cuda_go_icp_amd import synth
    amp = (0.35, 0.25, 0.30, 0.20, 0.35, 0.15, 0.28, 0.22)[seed % 8]
    tgt, src, R0, t0 = synth.make_pair(seed=7000 + seed, M=400, N=150, noise=0.004, amp=amp)
    rng = np.random.default_rng(9000 + seed)
    while True:
        v = rng.uniform(-np.pi, np.pi, 3)
        if np.linalg.norm(v) <= np.pi:
            break
    Rx = synth._rodrigues(v)
    tx = rng.uniform(-0.25, 0.25, 3)
    s2 = ((src.astype(np.float64) - tx) @ Rx).astype(np.float32)          # src = Rx s2 + tx  =>  target ~= R0 Rx s2 + R0 tx + t0
    return tgt, s2, R0 @ Rx, R0 @ tx + t0


def tiny_problem(seed):
    """The same construction at 200 target / 60 source points, registered at mse 5e-3 (SSEThresh 0.3 under the optimum's 0.35-0.55: the reference
    proves the optimum in 1.2-2.4 k rotation / 0.1-0.25 M translation nodes, 3-6 s of CPU): small enough for the engine's REFERENCE-ORDER mode
    (one expansion per launch) and for the CPU oracle -- tests/golden/e2e_tiny<seed>.json.  Returns (target, source)."""
    load_pkg()
    from cuda_go_icp_amd import synth
    amp = (0.35, 0.25, 0.30, 0.20)[seed % 4]
    tgt, src, _, _ = synth.make_pair(seed=7100 + seed, M=200, N=60, noise=0.004, amp=amp)
    rng = np.random.default_rng(9100 + seed)
    while True:
        v = rng.uniform(-np.pi, np.pi, 3)
        if np.linalg.norm(v) <= np.pi:
            break
    Rx = synth._rodrigues(v)
    tx = rng.uniform(-0.25, 0.25, 3)
    return tgt, ((src.astype(np.float64) - tx) @ Rx).astype(np.float32)


def rot_angle(Ra, Rb):
    """Geodesic angle between two rotations, from the chord ||Ra-Rb||_F = 2*sqrt(2)*sin(theta/2)
    (well conditioned near 0, unlike arccos of the trace)."""
    d = np.linalg.norm(np.asarray(Ra, dtype=np.float64).reshape(3, 3) - np.asarray(Rb, dtype=np.float64).reshape(3, 3))
    return 2 * np.arcsin(min(1.0, d / (2 * np.sqrt(2))))


def golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def bunny_model():
    return cloud("model_bunny")


@pytest.fixture(scope="session")
def bunny_data10():
    return cloud("data_bunny", 10)


@pytest.fixture(scope="session")
def bunny_data():
    return cloud("data_bunny")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def oracle_dt_bunny(oracle_mod, bunny_model):
    return oracle_mod.DistanceTransform(bunny_model, 300, 2.0)
