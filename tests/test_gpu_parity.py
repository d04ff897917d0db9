"""GPU parity tests: the HIP path, called through the C ABI (libgoicp_mi355.so), against
 (1) the committed golden vectors produced by the real reference CPU Go-ICP, and
 (2) the CPU oracle (oracle/goicp_oracle.c) on the same seeded inputs.
Tolerances are SURVEY.md 8(c)'s and are written next to each assertion.
"""
import os

import numpy as np
import pytest

from conftest import cloud, golden, load_pkg, rot_angle

pytestmark = pytest.mark.gpu

# Pose bar of every test below that compares with the reference's optimum outside _e2e (sharded, device- vs host-queue, flow, search-range
# and fp16 runs; it was 3e-2 rad / 1e-2 up to round 3).  Measured on MI355X (round 4, printed by _pose_close): 5.1e-6 rad / 5.8e-7 for
# every visit order that ends in the reference's ICP optimum, 4.3e-4 / 7.8e-5 for the boxed search -- so the bar is the strict mode's
# 2e-3 / 2e-3.  Three orders (flow = 4, flow = 16 without adaptive_k, RCCL world 1 at 8 parents per step) end in a NEIGHBOURING ICP optimum
# of the same basin, 2.075e-2 rad / 4.99e-3 away, whose SSE is LOWER than the reference's: Go-ICP guarantees the error, not the pose, and
# the early exit (jly_goicp.cpp:527) takes the first optimum below SSEThresh.  That one deviation is accepted only when the run's SSE beats
# the reference's, and held to the measured distance + 20 %.
POSE_TOL = (2e-3, 2e-3)
POSE_TOL_BETTER_OPTIMUM = (2.5e-2, 6e-3)


@pytest.fixture(scope="module")
def pkg():
    m = load_pkg()
    m.load_library()          # fails loudly if the HIP extension is missing
    return m


@pytest.fixture(scope="module")
def reg10(pkg, bunny_model, bunny_data10):
    r = pkg.Registration(bunny_model, bunny_data10, 1e-3, trans_batch=1, wide_children=0)
    yield r
    r.close()


@pytest.fixture(scope="module")
def rho10(oracle_mod, bunny_data10):
    return oracle_mod.rot_radii(bunny_data10)[1]


# ----------------------------------------------------------------------------------------------
# distance transform
# ----------------------------------------------------------------------------------------------
def test_dt_geometry_matches_reference(reg10):
    g = golden("dt_lookup")
    V, scale, origin = reg10.dt_info()
    assert V == 300 and scale == g["scale"]                         # bit-exact doubles (jly_3ddt.cpp:891-923)
    assert origin == (g["xmin"], g["ymin"], g["zmin"])


def test_dt_grid_bit_exact_vs_oracle(reg10, oracle_dt_bunny):
    """GPU exact-EDT build vs the oracle's (Meijster, CPU): integer squared distances -> bit-identical floats."""
    assert np.array_equal(reg10.dt_download(), oracle_dt_bunny.grid())


def test_dt_grid_vs_reference_voxels(reg10):
    g = golden("dt_lookup")
    v = np.array(g["voxel"]).reshape(-1, 3)
    ref = np.array(g["voxel_distance"], dtype=np.float32)
    mine = reg10.dt_download()[v[:, 2], v[:, 1], v[:, 0]]
    vox = 1.0 / g["scale"]
    assert np.all(mine <= ref + 1e-7) and np.max(ref - mine) <= 0.35 * vox


@pytest.mark.parametrize("layout,V", [(0, 96), (1, 96), (1, 50), (0, 33)])
def test_dt_layouts_agree(pkg, bunny_model, bunny_data10, oracle_mod, layout, V):
    """both layouts, including grid sides that are not a multiple of the 4-voxel brick"""
    r = pkg.Registration(bunny_model, bunny_data10, 1e-3, dt_layout=layout, dt_size=V)
    dt = oracle_mod.DistanceTransform(bunny_model, V, 2.0)
    assert np.array_equal(r.dt_download(), dt.grid())
    ub, lb = r.eval_bounds(np.eye(3), np.array([[0.1, -0.2, 0.05, 0.25]], np.float32), -1)
    oub, olb = oracle_mod.cube_bound(dt, bunny_data10, None, [0.1, -0.2, 0.05], 0.25)
    assert abs(ub[0] - oub) <= 1e-4 * oub and abs(lb[0] - olb) <= 1e-4 * max(olb, 1e-3)
    r.close()


def test_dt_maximum_size(pkg, bunny_model, bunny_data10, oracle_mod):
    """The largest grid the engine takes: dt_size = 640 (1.05 GB of fp32 + the same again for the nearest-point table; the kernels address the
    grid with 32-bit BYTE offsets, device.hip dt_fetch, so 640^3 x 4 B < 2^32 is the bound -- 641 is refused, not wrapped).  The whole grid is
    bit-identical to the oracle's exact EDT at that size, the highest offsets are reached through the lookup itself (a one-point source at
    the origin makes goicp_eval_bounds(I, q, w = 0) return Distance(q)^2: queries in the last voxels, on the faces and beyond them), a cube
    bound agrees with the oracle's on that grid, and an ICP run lands where the 300^3 engine's does."""
    V = 640
    with pytest.raises(Exception):
        pkg.Registration(bunny_model, bunny_data10, 1e-3, dt_size=V + 1)
    dt = oracle_mod.DistanceTransform(bunny_model, V, 2.0)
    reg = pkg.Registration(bunny_model, bunny_data10, 1e-3, dt_size=V)
    Vg, scale, origin = reg.dt_info()
    assert Vg == V and scale == dt.scale and tuple(origin) == tuple(dt.origin)
    grid = reg.dt_download()
    assert grid.shape == (V, V, V) and np.array_equal(grid, dt.grid())
    del grid
    ub, lb = reg.eval_bounds(np.eye(3), np.array([[0.1, -0.2, 0.05, 0.25]], np.float32), -1)
    oub, olb = oracle_mod.cube_bound(dt, bunny_data10, None, [0.1, -0.2, 0.05], 0.25)
    assert abs(ub[0] - oub) <= 1e-4 * oub and abs(lb[0] - olb) <= 1e-4 * max(olb, 1e-3)
    small = pkg.Registration(bunny_model, bunny_data10, 1e-3)
    e640, R640, t640 = pkg.IterativeClosestPoint3D(reg, 200, 1e-9).run()
    e300, R300, t300 = pkg.IterativeClosestPoint3D(small, 200, 1e-9).run()
    assert rot_angle(R640, R300) <= 1e-6 and np.linalg.norm(t640 - t300) <= 1e-6      # the neighbour search does not depend on the grid (only its start does)
    small.close(); reg.close()
    one = pkg.Registration(bunny_model, np.zeros((1, 3), np.float32), 1e-3, dt_size=V)
    x0 = np.array(dt.origin)
    hi = x0 + (V - 1) / dt.scale                                      # centre of the last voxel per axis
    rng = np.random.default_rng(5)
    q = np.concatenate([hi + rng.uniform(-1.5, 1.5, (256, 3)) / dt.scale,           # the last voxels and just beyond the far faces: the largest offsets
                        x0 + rng.uniform(-1.5, 1.5, (64, 3)) / dt.scale,            # the first voxels and just before the near faces
                        x0 + rng.uniform(0, V - 1, (256, 3)) / dt.scale]).astype(np.float32)
    ub, lb = one.eval_bounds(np.eye(3), np.concatenate([q, np.zeros((len(q), 1), np.float32)], 1), -1)
    d = dt.distance(q.astype(np.float64)).astype(np.float32)
    assert np.array_equal(ub, d * d) and np.array_equal(lb, ub)
    one.close()


# ----------------------------------------------------------------------------------------------
# (a) cube bounds
# ----------------------------------------------------------------------------------------------
def _cubes(rng, n):
    lev = rng.integers(0, 7, n)
    w = (1.0 / (1 << lev)).astype(np.float32)
    c = (rng.uniform(-0.5, 0.5, (n, 3)) * (1 - w[:, None])).astype(np.float32)
    return np.concatenate([c, w[:, None]], 1).astype(np.float32)


@pytest.mark.parametrize("layout,morton", [(1, 2), (0, 2), (1, 1), (1, 0)])
def test_eval_bounds_vs_oracle(pkg, oracle_mod, oracle_dt_bunny, bunny_model, bunny_data10, rho10, layout, morton):
    """ub/lb of 3 rotations x 64 cubes x {no radius, level 3..7}: rel 1e-4 (summation order only;
    every per-point term is bit-identical)."""
    reg = pkg.Registration(bunny_model, bunny_data10, 1e-3, dt_layout=layout, morton_sort=morton)
    rng = np.random.default_rng(7)
    for v in ([1.5707963, -1.5707963, 1.5707963], [0.3, -0.2, 0.9], [-2.1, 0.4, 1.1]):
        R = pkg.fgoicp.rodrigues(v)
        assert np.array_equal(R, oracle_mod.rodrigues(v))
        prot = oracle_mod.rotate(R, bunny_data10)
        cubes = _cubes(rng, 64)
        for level in (-1, 3, 5, 7):
            ub, lb = reg.eval_bounds(R, cubes, level)
            rho = rho10[level] if level >= 0 else None
            for i, c in enumerate(cubes):
                oub, olb = oracle_mod.cube_bound(oracle_dt_bunny, prot, rho, c[:3], c[3])
                assert abs(ub[i] - oub) <= 1e-4 * max(oub, 1e-3)
                assert abs(lb[i] - olb) <= 1e-4 * max(olb, 1e-3)
                assert lb[i] <= ub[i]
    reg.close()


def test_eval_bounds_far_outside_grid(reg10, oracle_mod, oracle_dt_bunny, bunny_data10):
    """Translations that push the whole cloud out of the DT grid exercise the clamp+overshoot
    extension of DT3D::Distance (jly_3ddt.cpp:991-1025)."""
    R = np.eye(3, dtype=np.float32)
    cubes = np.array([[3.0, 0, 0, 0.5], [-4.0, 2.5, 0.1, 0.25], [0.2, -9.0, 7.0, 1.0], [1.2, 1.2, -1.2, 0.5]], np.float32)
    ub, lb = reg10.eval_bounds(R, cubes, -1)
    for i, c in enumerate(cubes):
        oub, olb = oracle_mod.cube_bound(oracle_dt_bunny, bunny_data10, None, c[:3], c[3])
        assert abs(ub[i] - oub) <= 1e-4 * oub and abs(lb[i] - olb) <= 1e-4 * max(olb, 1e-3)


def test_eval_bounds_ragged_and_empty(reg10):
    R = np.eye(3, dtype=np.float32)
    ub, lb = reg10.eval_bounds(R, np.zeros((0, 4), np.float32), -1)
    assert ub.size == 0 and lb.size == 0
    rng = np.random.default_rng(3)
    cubes = _cubes(rng, 77)                                   # not a multiple of the 8-cube group
    ub_all, lb_all = reg10.eval_bounds(R, cubes, 4)
    for n in (1, 7, 8, 9, 63):
        ub, lb = reg10.eval_bounds(R, cubes[:n], 4)
        assert np.array_equal(ub, ub_all[:n]) and np.array_equal(lb, lb_all[:n])   # batch-size independent, bitwise


def test_eval_bounds_batch_mixed_rotations(pkg, reg10):
    """Generic batches (cubes of different rotations inside one 8-group) take the non-uniform path."""
    rng = np.random.default_rng(5)
    rots = np.stack([pkg.fgoicp.rodrigues(rng.uniform(-1.5, 1.5, 3)) for _ in range(5)])
    cubes = _cubes(rng, 37)
    rot_of = rng.integers(0, 5, 37)
    delta = np.array([reg10._lib.goicp_trans_delta(float(w)) for w in cubes[:, 3]], np.float32)
    coeff = reg10.rot_coeff(5)
    recs = [(c[0], c[1], c[2], delta[i], coeff, int(rot_of[i])) for i, c in enumerate(cubes)]
    ub, lb = reg10.eval_bounds_batch(rots, recs)
    for k in range(5):
        sel = np.nonzero(rot_of == k)[0]
        u2, l2 = reg10.eval_bounds(rots[k], cubes[sel], 5)
        assert np.allclose(ub[sel], u2, rtol=2e-6, atol=0) and np.allclose(lb[sel], l2, rtol=2e-6, atol=1e-9)


def test_eval_bounds_grouped_same_bits(pkg, reg10):
    """goicp_eval_bounds_device_grouped: an unrelated batch (random rotation, pass and translation per cube, some translations outside
    the root cube, a ragged size) bucketed on the device by (rotation, pass, translation cell), evaluated and written back in the
    caller's order -- every bound bit-equal to the plain entry point's."""
    import ctypes as C
    B = pkg.binding
    lib, h = reg10._lib, reg10.handle
    hip = C.CDLL("libamdhip64.so")                     # the runtime the library itself is linked against (device buffers for the device-pointer entry points)
    hip.hipMalloc.argtypes, hip.hipFree.argtypes = [C.POINTER(C.c_void_p), C.c_size_t], [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def to_dev(a):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), a.nbytes) == 0 and hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
        return p

    def to_host(p, n):
        out = np.empty(n, np.float32)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, 2) == 0       # blocking: orders after the engine's stream work below
        return out

    rng = np.random.default_rng(17)
    for n, nrot in ((4099, 5), (65536, 8), (7, 1)):
        rots = np.stack([pkg.fgoicp.rodrigues(rng.uniform(-2.0, 2.0, 3)) for _ in range(nrot)]).astype(np.float32)
        recs = np.zeros(n, dtype=[("tx", "<f4"), ("ty", "<f4"), ("tz", "<f4"), ("delta", "<f4"), ("coeff", "<f4"), ("rot", "<i4")])
        c = rng.uniform(-0.7, 0.7, (n, 3)).astype(np.float32)
        recs["tx"], recs["ty"], recs["tz"] = c[:, 0], c[:, 1], c[:, 2]
        recs["delta"] = rng.choice(np.array([lib.goicp_trans_delta(1.0 / (1 << k)) for k in range(1, 7)], np.float32), n)
        recs["coeff"] = np.where(rng.random(n) < 0.5, np.float32(reg10.rot_coeff(4)), np.float32(0))
        recs["rot"] = rng.integers(0, nrot, n)
        d_rots, d_cubes = to_dev(np.ascontiguousarray(rots.reshape(-1))), to_dev(np.ascontiguousarray(recs).view(np.uint8))
        outs = [to_dev(np.full(n, -1.0, np.float32)) for _ in range(4)]
        ms = C.c_float()
        # the timing entries run the evaluation on the engine's own stream and wait for it
        B.check(lib.goicp_time_bounds_device(h, d_rots, d_cubes, n, outs[0], outs[1], 1, C.byref(ms)))
        B.check(lib.goicp_time_bounds_device_grouped(h, d_rots, nrot, d_cubes, n, outs[2], outs[3], 1, C.byref(ms)))
        ub, lb, gub, glb = (to_host(o, n) for o in outs)
        assert np.array_equal(ub, gub) and np.array_equal(lb, glb), n
        assert ub.min() >= 0.0 and np.all(lb <= ub)
        for p in [d_rots, d_cubes] + outs:
            hip.hipFree(p)


def test_golden_single_expansions(pkg, reg10):
    """The reference's own InnerBnB, one expansion: min ub over the 8 children and the arg-min child
    (tests/golden/inner_bnb.json 'single').  rel 1e-4."""
    g = golden("inner_bnb")
    for case in g["cases"]:
        R = np.array(case["R"], np.float32)
        for s in case["single"]:
            px, py, pz, pw = map(np.float32, s["parent"])
            w = pw / np.float32(2)
            kids = []
            for j in range(8):
                cx = px + np.float32(j & 1) * w; cy = py + np.float32(j >> 1 & 1) * w; cz = pz + np.float32(j >> 2 & 1) * w
                kids.append([cx + w / np.float32(2), cy + w / np.float32(2), cz + w / np.float32(2), w, cx, cy, cz])
            kids = np.array(kids, np.float32)
            ub, _ = reg10.eval_bounds(R, kids[:, :4], s["level"])
            j = int(np.argmin(ub))
            assert abs(ub[j] - s["min_ub"]) <= 1e-4 * max(s["min_ub"], 1e-3)
            if np.sum(np.abs(ub - ub[j]) <= 2e-5 * max(ub[j], 1e-3)) == 1:   # unambiguous arg-min
                assert np.array_equal(kids[j, 4:7], np.array(s["best"][:3], np.float32))


def test_golden_inner_bnb_full(pkg, reg10):
    """Whole InnerBnB calls in the reference visit order (trans_batch=1): value rel 1e-3, node pops
    within 1 %, best node identical."""
    g = golden("inner_bnb")
    for case in g["cases"]:
        R = np.array(case["R"], np.float32)
        for s in case["full"]:
            v, best, cnt = reg10.inner_bnb(R, s["level"], s["incumbent"])
            assert abs(v - s["value"]) <= 1e-3 * max(s["value"], 1e-3)
            assert abs(cnt.trans_pops - s["pops"]) <= max(2, 0.01 * s["pops"])
            if s["value"] < s["incumbent"]:
                assert best is not None
                if s["level"] < 0:
                    assert np.array_equal(best, np.array(s["best"], np.float32))
            else:
                assert best is None


def test_inner_bnb_batched_same_optimum(pkg, bunny_model, bunny_data10):
    """Expanding 16 nodes per launch visits more nodes but must return the same optimum (within
    SSEThresh, by construction of the stop rule)."""
    g = golden("inner_bnb")
    reg = pkg.Registration(bunny_model, bunny_data10, 1e-3, trans_batch=16)
    for case in g["cases"]:
        R = np.array(case["R"], np.float32)
        for s in case["full"]:
            v, _, _ = reg.inner_bnb(R, s["level"], s["incumbent"])
            assert abs(v - s["value"]) <= g["sse_threshold"] + 1e-3 * s["value"]
    reg.close()


def test_eval_sse_golden(reg10):
    g = golden("icp_dt_score")
    sse = reg10.compute_sse_error(np.array(g["R"]), np.array(g["t"]))
    assert abs(sse - g["dt_sse"]) <= 1e-4 * g["dt_sse"]               # jly_goicp.cpp:93-132


# ----------------------------------------------------------------------------------------------
# (b) ICP
# ----------------------------------------------------------------------------------------------
def test_nn_exact_vs_reference_kdtree(reg10):
    g = golden("nn")
    q = np.array(g["query"], np.float32).reshape(-1, 3)
    idx, d2 = reg10.nn_query(q)
    assert np.array_equal(d2, np.array(g["dist_sq"], np.float32))      # bit-exact squared distances
    assert np.mean(idx == np.array(g["index"])) > 0.999                # ties only


def test_nn_vs_bruteforce(reg10, oracle_mod, bunny_model):
    rng = np.random.default_rng(11)
    q = rng.uniform(-1.5, 1.5, (2000, 3)).astype(np.float32)
    q[:500] = bunny_model[rng.integers(0, len(bunny_model), 500)]      # exact hits: distance 0, lowest index wins
    idx, d2 = reg10.nn_query(q)
    bi, bd = oracle_mod.nn_brute(bunny_model, q)
    assert np.array_equal(d2, bd) and np.array_equal(idx, bi)


@pytest.mark.parametrize("side", [6, 14])
def test_nn_ties_on_a_lattice(pkg, oracle_mod, side):
    """Lattice target (side^3 points: one- and two-level hierarchies) with duplicated points, queries at
    cell centres / face centres / lattice points: 8-, 4-, 2-way and duplicate ties must all resolve to the
    lowest original index, exactly as a brute-force scan does."""
    g = (np.arange(side, dtype=np.float32) - (side - 1) / 2) * np.float32(0.125)       # exactly representable
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(side)
    target = np.concatenate([lat, lat[rng.integers(0, len(lat), len(lat) // 4)]])       # duplicates
    target = target[rng.permutation(len(target))].astype(np.float32)
    src = lat[:64].copy()
    reg = pkg.Registration(target, src, 1e-3, dt_size=64)
    h = np.float32(0.0625)
    q = np.concatenate([lat[:400] + h, lat[:400] + np.array([h, h, 0], np.float32),
                        lat[:400] + np.array([h, 0, 0], np.float32), lat[:400]]).astype(np.float32)
    idx, d2 = reg.nn_query(q)
    bi, bd = oracle_mod.nn_brute(target, q)
    assert np.array_equal(d2, bd) and np.array_equal(idx, bi)
    reg.close()


@pytest.mark.parametrize("M", [17, 33, 1024, 1025, 2049, 65536, 65537])
def test_nn_hierarchy_shapes(pkg, oracle_mod, M):
    """The hierarchy's binary depth follows the cloud (full leaves, sparse root group): depths 1, 2, 6, 7, 8,
    12, 13 = one, two and three box levels with 2 .. 64 real root children.  Exact vs brute force."""
    rng = np.random.default_rng(M)
    target = rng.uniform(-0.6, 0.6, (M, 3)).astype(np.float32)
    reg = pkg.Registration(target, target[:8].copy(), 1e-3, dt_size=48)
    q = np.concatenate([rng.uniform(-0.8, 0.8, (600, 3)), target[rng.integers(0, M, 200)]]).astype(np.float32)
    idx, d2 = reg.nn_query(q)
    bi, bd = oracle_mod.nn_brute(target, q)
    assert np.array_equal(d2, bd) and np.array_equal(idx, bi)
    reg.close()


def test_nn_three_level_hierarchy(pkg, oracle_mod):
    """M = 120 000 > 64*64*16 exercises the K = 3 box hierarchy (BASELINE configs[2]/[4] sizes): exact vs
    brute force, including queries far outside the cloud and outside the DT grid."""
    from cuda_go_icp_amd import synth
    target, source, _, _ = synth.make_pair(seed=77, M=120000, N=2000)
    reg = pkg.Registration(target, source, 1e-3, dt_size=128)
    rng = np.random.default_rng(5)
    q = np.concatenate([source, rng.uniform(-3, 3, (1000, 3)).astype(np.float32), target[:500]])
    idx, d2 = reg.nn_query(q)
    bi, bd = oracle_mod.nn_brute(target, q)
    assert np.array_equal(d2, bd) and np.array_equal(idx, bi)
    err, R, t = pkg.IterativeClosestPoint3D(reg, 5, 1e-9).run()            # and the fused pass agrees with the oracle ICP
    kd = oracle_mod.KdTree(target)
    oerr, oR, ot, _ = kd.icp_run(source, np.eye(3), np.zeros(3), 5, 1e-9)
    assert abs(err - oerr) <= 1e-3 * oerr and np.abs(R - oR).max() <= 1e-4 and np.abs(t - ot).max() <= 1e-4
    reg.close()


def test_icp_run_golden(pkg, reg10):
    """IterativeClosestPoint3D::run / ICP3D::Run with forced iteration counts: 1e-4 abs on R,t for
    <= 10 iterations, 1e-3 for the converged runs; err rel 1e-3."""
    g = golden("icp_iter")
    for c in g["cases"]:
        icp = pkg.IterativeClosestPoint3D(reg10, c["max_iter"], c["err_diff"], c["R0"], c["t0"])
        err, R, t = icp.run()
        tol = 1e-4 if c["max_iter"] <= 10 else 1e-3
        assert np.abs(R.ravel() - np.array(c["R"])).max() <= tol
        assert np.abs(t - np.array(c["t"])).max() <= tol
        assert abs(err - c["err"]) <= 1e-3 * c["err"]
        if c["max_iter"] <= 10:
            assert icp.iters == c["max_iter"]


def test_icp_first_pass_error_is_bruteforce_nn_sum(pkg, oracle_mod, bunny_model, bunny_data10):
    """The correspondence pass is exact: its error equals the brute-force nearest-neighbour sum."""
    reg = pkg.Registration(bunny_model, bunny_data10, 1e-3)
    _, d2 = oracle_mod.nn_brute(bunny_model, bunny_data10)
    err = reg.icp_step().best_sse
    assert abs(err - float(np.sum(d2.astype(np.float64)))) <= 5e-6 * float(np.sum(d2))
    reg.close()


def test_icp_fixed_point_sums_scale_and_repeat(pkg, bunny_model, bunny_data10):
    """The pass adds its workgroup sums as 64-bit fixed point (integer atomics; the scale is derived from the clouds'
    extents and the start pose).  (1) The same clouds in other units -- x500 and pushed 300 units off the origin, and
    x1e-3 -- take the same ICP trajectory: same iteration count, same rotation, translation scaled, error scaled by the
    square (no overflow, no loss of resolution).  (2) Integer sums do not depend on the order the workgroups arrive in:
    two runs of the same ICP give bit-identical poses.  (3) A start pose far outside the clouds is still summed right
    (the scale follows the start pose): first-pass error against brute force."""
    base = pkg.Registration(bunny_model, bunny_data10, 1e-3)
    icp0 = pkg.IterativeClosestPoint3D(base, 40, 1e-7, np.eye(3), np.zeros(3))
    e0, R0, t0 = icp0.run()
    icp1 = pkg.IterativeClosestPoint3D(base, 40, 1e-7, np.eye(3), np.zeros(3))
    e1, R1, t1 = icp1.run()
    assert e0 == e1 and np.array_equal(R0, R1) and np.array_equal(t0, t1) and icp0.iters == icp1.iters
    for s, off in ((500.0, np.array([300.0, -120.0, 40.0], np.float32)), (1e-3, np.zeros(3, np.float32))):
        reg = pkg.Registration((bunny_model * np.float32(s) + off).astype(np.float32), (bunny_data10 * np.float32(s) + off).astype(np.float32), 1e-3)
        icp = pkg.IterativeClosestPoint3D(reg, 40, 1e-7 * s * s, np.eye(3), np.zeros(3))
        e, R, t = icp.run()
        # x' = s x + off on both clouds: R' = R, t' = s t + off - R off
        assert np.abs(R - R0).max() <= 2e-3, (s, np.abs(R - R0).max())
        assert np.abs(t - (s * t0 + off - R0 @ off)).max() <= 5e-3 * s * max(1.0, np.abs(off).max() / s / 10), (s, t, t0)
        assert abs(e - e0 * s * s) <= 2e-2 * e0 * s * s, (s, e, e0)
        reg.close()
    far = np.array([40.0, -25.0, 10.0], np.float32)
    one = pkg.IterativeClosestPoint3D(base, 1, 1e-7, np.eye(3), far)
    err_far, _, _ = one.run()
    q = (bunny_data10 + far).astype(np.float32)
    from scipy.spatial import cKDTree
    dd, _ = cKDTree(bunny_model.astype(np.float64)).query(q.astype(np.float64))
    assert abs(err_far - float((dd ** 2).sum())) <= 1e-4 * float((dd ** 2).sum())
    base.close()


def test_icp_point_seed_is_exact(pkg, oracle_mod, bunny_model, bunny_data10):
    """Params::icp_point_seed: the neighbour walks start from a real candidate read from the per-voxel nearest-target-point
    table (the EDT passes carrying their arg-min) instead of from the distance-transform bound.  Any target point is a valid
    candidate, so nothing may change: NN indices / distances bit-equal to brute force (queries near, far and outside the
    grid), and ICP trajectories bit-identical with the table on and off."""
    rng = np.random.default_rng(21)
    q = np.concatenate([bunny_data10[:1500], rng.uniform(-1.5, 1.5, (1500, 3)).astype(np.float32), rng.uniform(-6.0, 6.0, (300, 3)).astype(np.float32)])
    out = {}
    for seed in (0, 1):
        reg = pkg.Registration(bunny_model, bunny_data10, 1e-3, icp_point_seed=seed)
        idx, d2 = reg.nn_query(q)
        e, R, t = pkg.IterativeClosestPoint3D(reg, 30, 1e-7, np.eye(3), np.zeros(3)).run()
        out[seed] = (idx, d2, e, R, t)
        reg.close()
    bi, bd = oracle_mod.nn_brute(bunny_model, q)
    for seed in (0, 1):
        assert np.array_equal(out[seed][0], bi) and np.array_equal(out[seed][1], bd)
    assert out[0][2] == out[1][2] and np.array_equal(out[0][3], out[1][3]) and np.array_equal(out[0][4], out[1][4])


def test_icp_step_decreases_error(pkg, bunny_model, bunny_data10):
    reg = pkg.Registration(bunny_model, bunny_data10, 1e-3)
    errs = [reg.icp_step().best_sse for _ in range(6)]
    assert all(b <= a * (1 + 1e-5) for a, b in zip(errs, errs[1:]))
    g = golden("icp_iter")["cases"][0]                                 # first pass error from identity
    assert abs(errs[0] - g["err"]) <= 1e-3 * g["err"]
    reg.close()


# ----------------------------------------------------------------------------------------------
# end to end (FastGoICP::run == GoICP::Register)
# ----------------------------------------------------------------------------------------------
def _e2e(pkg, tag, model, data, strict=True, wide_tol=(2e-3, 2e-3), **params):
    """strict (reference visit order): R within 2e-3 rad, t within 2e-3, SSE within 2 %.
    Widened search (the product default: speculative batches, device queues): Go-ICP's guarantee is the error, not the
    pose -- the early exit (jly_goicp.cpp:527) accepts the first ICP optimum below SSEThresh, and a different visit order
    can hand ICP a different start inside the same basin.  So the error bar is one-sided (SSE <= reference + 2 %, below
    SSEThresh); the pose is held to the SAME 2e-3 rad / 2e-3 as the strict mode, which is what the default mode
    measures on MI355X (round 3, printed by every run of this helper): rand-100 7e-7 rad / 1e-7, bunny/10 5e-6 / 6e-7,
    full bunny 3.4e-4 / 1.4e-4, skull/10 4e-6 / 7e-6.  A caller whose landscape is flat around the optimum passes its own
    wide_tol and says why."""
    g = golden("e2e_" + tag)
    eng = pkg.FastGoICP(model, data, g["mse_threshold"], **params)
    eng.run()
    assert eng.finished
    sse = eng.get_best_error()
    ang, dt = rot_angle(eng.optR, np.array(g["R"])), np.linalg.norm(eng.optT - np.array(g["t"]))
    slack = 1e-2 * g["sse_threshold"]          # the strided skull / spanner optima score an SSE of (nearly) 0: an absolute term, 1 % of what Go-ICP guarantees
    if strict:
        assert ang <= 2e-3 and dt <= 2e-3, (ang, dt)
        assert abs(sse - g["sse"]) <= 0.02 * g["sse"] + slack
    else:
        assert ang <= wide_tol[0] and dt <= wide_tol[1], (ang, dt)
        assert sse <= 1.02 * g["sse"] + slack
    assert sse < g["sse_threshold"]
    print("e2e %s strict=%d: rot_error_rad %.3e trans_error %.3e sse %.6g (reference %.6g)" % (tag, strict, ang, dt, sse, g["sse"]))
    return eng, g



def _pose_close(tag, R, t, g, tol=(2e-3, 2e-3), sse=None):
    """Pose against the reference's optimum (golden e2e file g), printed so that the tolerance can be held to what is measured.
    sse given: a pose outside `tol` is accepted when the run's SSE is strictly below the reference's (a better optimum than the
    reference found) and the pose lies within POSE_TOL_BETTER_OPTIMUM."""
    ang, dt = rot_angle(R, np.array(g["R"])), float(np.linalg.norm(np.asarray(t, np.float64).reshape(3) - np.array(g["t"])))
    print("pose %s: rot_error_rad %.3e trans_error %.3e sse %s (reference %.6g; tolerance %.1e / %.1e)" % (tag, ang, dt, "%.6g" % sse if sse is not None else "-", g["sse"], tol[0], tol[1]))
    if ang <= tol[0] and dt <= tol[1]:
        return
    assert sse is not None and sse < g["sse"] and ang <= POSE_TOL_BETTER_OPTIMUM[0] and dt <= POSE_TOL_BETTER_OPTIMUM[1], (tag, ang, dt, sse)


def test_e2e_rand100_reference_order(pkg):
    eng, g = _e2e(pkg, "rand100", cloud("model_rand"), cloud("data_rand"), trans_batch=1, wide_children=0)
    c = eng.counters
    assert c.rot_pops == g["rNodeCount"]
    assert abs(c.trans_pops - g["tNodeCount"]) <= 0.01 * g["tNodeCount"]


def test_e2e_rand100_wide(pkg):
    _e2e(pkg, "rand100", cloud("model_rand"), cloud("data_rand"), strict=False)


def test_e2e_bunny10_reference_order(pkg, bunny_model, bunny_data10):
    eng, g = _e2e(pkg, "bunny10", bunny_model, bunny_data10, trans_batch=1, wide_children=0)
    c = eng.counters
    assert abs(c.rot_pops - g["rNodeCount"]) <= 0.02 * g["rNodeCount"]
    assert abs(c.trans_pops - g["tNodeCount"]) <= 0.02 * g["tNodeCount"]


def test_e2e_bunny10_wide(pkg, bunny_model, bunny_data10):
    _e2e(pkg, "bunny10", bunny_model, bunny_data10, strict=False)


def test_e2e_bunny_full_wide(pkg, bunny_model, bunny_data, tmp_path):
    """BASELINE configs[1]: bunny_goicp, N = 30379, V = 300 (reference CPU: 502.7 s)."""
    eng, g = _e2e(pkg, "bunny_full", bunny_model, bunny_data, strict=False)
    out = tmp_path / "output.toml"
    eng.write_output(out)
    txt = out.read_text()
    assert "rotation" in txt and "translation" in txt and "sse" in txt
    viz = tmp_path / "viz.ply"
    eng.write_visualization(viz)
    back = pkg.load_cloud(viz)                        # our own loader reads it back: target + moved source
    assert back.shape == (len(bunny_model) + len(bunny_data), 3)
    assert np.array_equal(back[:len(bunny_model)], bunny_model)
    moved = eng.registration.transform_source(eng.optR, eng.optT)
    assert np.array_equal(back[len(bunny_model):], moved)


def test_e2e_bunny_full_reference_order(pkg, bunny_model, bunny_data):
    """Full bunny in the reference visit order (8 cubes per launch): strict pose/SSE parity and node
    counts against the reference's own run (378 rotation / 46 862 translation nodes)."""
    eng, g = _e2e(pkg, "bunny_full", bunny_model, bunny_data, trans_batch=1, wide_children=0)
    c = eng.counters
    assert abs(c.rot_pops - g["rNodeCount"]) <= 0.02 * g["rNodeCount"]
    assert abs(c.trans_pops - g["tNodeCount"]) <= 0.02 * g["tNodeCount"]


def _e2e_counts(eng, g):
    c = eng.counters
    assert abs(c.rot_pops - g["rNodeCount"]) <= max(1, 0.02 * g["rNodeCount"]), (c.rot_pops, g["rNodeCount"])
    assert abs(c.trans_pops - g["tNodeCount"]) <= 0.02 * g["tNodeCount"], (c.trans_pops, g["tNodeCount"])


@pytest.mark.parametrize("lanes", [1, 2, 4])
def test_lanes_prove_the_reference_optimum(pkg, lanes):
    """goicp_params::lanes: a batch of inner searches (GoICP::InnerBnB calls, jly_goicp.cpp:227-340 -- independent of each other) cut into lanes
    by rotation child, the lanes run their lock-step rounds side by side on their own streams (engine.cpp run_inner_device).  Per search nothing
    changes (same bounds, same stop and prune rules), so with the cut forced on every batch (lanes = 2 / 4, lane_min_searches = 2) the converged
    search of a small seeded problem meets the same bars against the reference's own GoICP::Register (tests/golden/e2e_small4.json) as with
    one lane, and the counters say which path ran."""
    from conftest import small_problem
    tgt, src, _, _ = small_problem(4)
    g = golden("e2e_small4")
    eng = pkg.FastGoICP(tgt, src, g["mse_threshold"], lanes=lanes, lane_min_searches=2)
    eng.run()
    c = eng.counters
    sse = float(eng.get_best_error())
    ang, dt = rot_angle(eng.optR, np.array(g["R"])), float(np.linalg.norm(eng.optT - np.array(g["t"])))
    print("lanes %d: sse %.7g (reference %.7g) rot_error %.2e trans_error %.2e rotation nodes %d (reference %d) cube bounds %d two-lane batches %d" % (
        lanes, sse, g["sse"], ang, dt, c.rot_pops, g["rNodeCount"], c.cubes, c.lane_batches))
    assert eng.finished and abs(sse - g["sse"]) <= 1e-5 * g["sse"]
    assert ang <= 1e-5 and dt <= 1e-5, (ang, dt)
    assert abs(c.rot_pops - g["rNodeCount"]) <= 0.01 * g["rNodeCount"]
    assert abs(c.cubes - 8 * g["tNodeCount"]) <= 0.05 * 8 * g["tNodeCount"]
    assert c.queue_fallbacks == 0
    assert (c.lane_batches > 0) == (lanes >= 2)
    eng.registration.close()


def test_auto_lanes_are_deterministic_and_change_no_result(pkg, bunny_model, bunny_data):
    """lanes = 0 (the default) cuts a batch in two when the PREVIOUS batch's mean round was throughput-bound -- a count of point-expansions, not
    a time -- so two runs of one engine take the same path: identical counters, identical result.  Bunny at mse 1e-4 (the search proves the
    optimum: 994 rotation nodes, 8.8 M cube bounds) has such batches; its result equals the single-lane run's (same SSE to 1e-6 relative,
    same pose, same rotation nodes; cube bounds within 0.1 %: the adaptive round width follows each lane's own count of running searches).
    The default registration (mse 1e-3: early exit after 384 rotation nodes) never qualifies."""
    auto = pkg.FastGoICP(bunny_model, bunny_data, 1e-4)
    one = pkg.FastGoICP(bunny_model, bunny_data, 1e-4, lanes=1)
    seen = []
    for _ in range(2):
        auto.run()
        c = auto.counters
        seen.append((float(auto.get_best_error()), auto.optR.tobytes(), auto.optT.tobytes(), c.cubes, c.rot_pops, c.trans_pops, c.icp_iters, c.lane_batches, c.bounds_launches))
    assert seen[0] == seen[1], (seen[0][3:], seen[1][3:])
    one.run()
    ca, c1 = auto.counters, one.counters
    print("auto lanes: %d two-lane batches, cube bounds %d vs %d (one lane), rotation nodes %d vs %d, sse %.7g vs %.7g" % (
        ca.lane_batches, ca.cubes, c1.cubes, ca.rot_pops, c1.rot_pops, auto.get_best_error(), one.get_best_error()))
    assert ca.lane_batches > 0 and c1.lane_batches == 0
    assert abs(auto.get_best_error() - one.get_best_error()) <= 1e-6 * one.get_best_error()
    assert rot_angle(auto.optR, one.optR) <= 1e-6 and float(np.linalg.norm(auto.optT - one.optT)) <= 1e-6
    assert ca.rot_pops == c1.rot_pops and abs(ca.cubes - c1.cubes) <= 1e-3 * c1.cubes
    shallow = pkg.FastGoICP(bunny_model, bunny_data, 1e-3)
    shallow.run()
    assert shallow.counters.lane_batches == 0
    for e in (auto, one, shallow):
        e.registration.close()


@pytest.mark.parametrize("seed", [1, 4, 6, 8])
def test_e2e_small_proven_optimum(pkg, seed):
    """The converged-search counterpart of the early-exit fixtures: small seeded problems (conftest.small_problem: 400-point target, 150-point
    source, mse 1e-3) whose optimum scores ABOVE SSEThresh, so the reference's own GoICP::Register (tests/golden/e2e_small<seed>.json) runs its
    outer BnB to convergence -- 7-18 k rotation nodes, 9-19 M translation nodes, 10-21 minutes of CPU -- and PROVES the optimum to within
    SSEThresh.  The engine's default (widened, device-queue) search proves it too: its SSE lies within SSEThresh of the reference's (both are
    within SSEThresh of the global optimum from above).  Measured (round 4): the SAME optimum -- SSE equal to 7 digits, pose to 5e-7 rad / 1e-7
    -- in 0.24-0.53 s against 638-1 288 s, and, although the visit order differs, nearly the same search: 12 457 / 9 965 / 7 407 / 18 256
    rotation nodes against the reference's 12 455 / 9 965 / 7 403 / 18 248, cube bounds within 2 % of 8 x its tNodeCount (a converged BnB has
    to expand every node whose lower bound stays SSEThresh under the optimum, whatever the order).  Bars: pose 1e-5, rotation nodes 1 %,
    cube bounds 5 %."""
    from conftest import small_problem
    tgt, src, _, _ = small_problem(seed)
    g = golden("e2e_small%d" % seed)
    assert g["sse"] > g["sse_threshold"] and g["rNodeCount"] > 5000          # the point of these fixtures: no early exit
    eng = pkg.FastGoICP(tgt, src, g["mse_threshold"])
    import time
    t0 = time.perf_counter()
    eng.run()
    wall = time.perf_counter() - t0
    sse = float(eng.get_best_error())
    ang, dt = rot_angle(eng.optR, np.array(g["R"])), float(np.linalg.norm(eng.optT - np.array(g["t"])))
    c = eng.counters
    print("e2e small%d: %.3f s (reference CPU %.0f s), sse %.6g (reference %.6g), rot_error %.3e trans_error %.3e, rotation nodes %d (reference %d), cube bounds %d (reference <= %d)"
          % (seed, wall, g["register_s"], sse, g["sse"], ang, dt, c.rot_pops, g["rNodeCount"], c.cubes, 8 * g["tNodeCount"]))
    assert eng.finished and abs(sse - g["sse"]) <= 1e-5 * g["sse"]
    assert ang <= 1e-5 and dt <= 1e-5, (ang, dt)
    assert abs(c.rot_pops - g["rNodeCount"]) <= 0.01 * g["rNodeCount"]
    assert abs(c.cubes - 8 * g["tNodeCount"]) <= 0.05 * 8 * g["tNodeCount"]
    eng.registration.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 6])
def test_e2e_tiny_proven_optimum_reference_order(pkg, seed):
    """A CONVERGED search in the reference's own visit order (trans_batch = 1, wide_children = 0: one expansion per launch, host queues): tiny
    seeded problems (conftest.tiny_problem, 200 x 60 points, mse 5e-3) on which the reference's GoICP::Register proves the optimum
    (tests/golden/e2e_tiny<seed>.json: 1 201-2 277 rotation nodes, 0.11-0.25 M translation nodes).  Measured (round 4): the same optimum to 7
    digits, the rotation-node counts IDENTICAL (1 609 / 1 857 / 1 201 / 2 277), the translation-node counts identical on three of the four
    (202 740, 107 515, 248 359) and 177 133 against 177 135 on the fourth (float sums in another order flip a prune decision now and then).
    Bars: rotation nodes equal, translation nodes within 0.1 %."""
    from conftest import tiny_problem
    tgt, src = tiny_problem(seed)
    g = golden("e2e_tiny%d" % seed)
    assert g["sse"] > g["sse_threshold"] and g["rNodeCount"] > 1000
    eng = pkg.FastGoICP(tgt, src, g["mse_threshold"], trans_batch=1, wide_children=0)
    eng.run()
    sse, c = float(eng.get_best_error()), eng.counters
    ang, dt = rot_angle(eng.optR, np.array(g["R"])), float(np.linalg.norm(eng.optT - np.array(g["t"])))
    print("e2e tiny%d reference order: sse %.7g (reference %.7g) rot_error %.2e trans_error %.2e rotation nodes %d (reference %d) translation nodes %d (reference %d)"
          % (seed, sse, g["sse"], ang, dt, c.rot_pops, g["rNodeCount"], c.trans_pops, g["tNodeCount"]))
    assert eng.finished and abs(sse - g["sse"]) <= 1e-5 * g["sse"] and ang <= 1e-5 and dt <= 1e-5
    assert c.rot_pops == g["rNodeCount"]
    assert abs(c.trans_pops - g["tNodeCount"]) <= 1e-3 * g["tNodeCount"]
    eng.registration.close()
    # the default (widened) search proves the same optimum
    eng = pkg.FastGoICP(tgt, src, g["mse_threshold"])
    eng.run()
    assert abs(float(eng.get_best_error()) - g["sse"]) <= 1e-5 * g["sse"] and rot_angle(eng.optR, np.array(g["R"])) <= 1e-5
    # (batches of up to 64 parents expand a few cubes a one-at-a-time order would have pruned first: + 8 ... 12 % on these 1.2-2.3 k nodes, + 0.1 % on the 7-18 k of e2e_small*)
    print("e2e tiny%d default mode: rotation nodes %d (reference %d)" % (seed, eng.counters.rot_pops, g["rNodeCount"]))
    assert 0.98 * g["rNodeCount"] <= eng.counters.rot_pops <= 1.25 * g["rNodeCount"], (eng.counters.rot_pops, g["rNodeCount"])
    eng.registration.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_ranks_prove_the_reference_optimum(pkg, world):
    """SURVEY 8(e)'s invariant on a CONVERGED search: conftest.small_problem(6) (the reference's Register proves the optimum in 7 403 rotation
    nodes, tests/golden/e2e_small6.json) sharded over 2 / 4 / 8 ranks (host threads on this GPU, the library's protocol with its default step
    ramp): every rank ends with the reference's optimum (SSE 1e-5, pose 1e-5), the ranks' rotation nodes add up to the reference's count within
    10 % (a rank's share of the frontier is expanded in its own order; measured 7 431 / 7 477 / 7 559 against 7 403), no rank fails."""
    from conftest import small_problem
    from cuda_go_icp_amd import sharded
    tgt, src, _, _ = small_problem(6)
    g = golden("e2e_small6")
    engines = [pkg.FastGoICP(tgt, src, g["mse_threshold"]) for _ in range(world)]
    stats = sharded.run_thread_ranks(engines, rot_pops_per_step=8, ramp_to=32)
    assert all(s["status"] == 0 and s["failed_rank"] == -1 for s in stats)
    rots = sum(e.counters.rot_pops for e in engines)
    print("sharded small6, %d ranks: rotation nodes of all ranks %d (reference %d), steps %d, donations %d" % (world, rots, g["rNodeCount"], stats[0]["steps"], stats[0]["donations"]))
    for e in engines:
        assert abs(float(e.get_best_error()) - g["sse"]) <= 1e-5 * g["sse"]
        assert rot_angle(e.optR, np.array(g["R"])) <= 1e-5 and np.linalg.norm(e.optT - np.array(g["t"])) <= 1e-5
        e.registration.close()
    assert 0.98 * g["rNodeCount"] <= rots <= 1.10 * g["rNodeCount"]


def test_e2e_skull_sub_reference_order(pkg):
    """BASELINE configs[2] pinned to the reference: the real GoICP::Register (src/goicp/jly_goicp.cpp:569-585) on the
    skull scan (98 359-point target, the DT over all of it) and every 10th point of the known-motion source
    (tests/golden/e2e_skull_sub.json, oracle/gen_golden.py --sub-configs; mse 1e-3, test/skull_goicp.toml:10-20).
    Reference visit order: pose <= 2e-3 rad / 2e-3, SSE, node counts within 2 %."""
    from conftest import skull_problem
    target, source, _, _ = skull_problem()
    eng, g = _e2e(pkg, "skull_sub", target, np.ascontiguousarray(source[::10]), trans_batch=1, wide_children=0)
    _e2e_counts(eng, g)
    eng.registration.close()


def test_e2e_skull_sub_wide(pkg):
    from conftest import skull_problem
    target, source, _, _ = skull_problem()
    eng, _ = _e2e(pkg, "skull_sub", target, np.ascontiguousarray(source[::10]), strict=False)
    eng.registration.close()


def test_e2e_spanner_sub_reference_order(pkg):
    """BASELINE configs[3] pinned to the reference: GoICP::Register on the noisy spanner (150 000-point target) and every
    50th point of the rotated model, mse 1e-4 (test/spanner_goicp.toml:10-20; tests/golden/e2e_spanner_sub.json: 92
    rotation / 16 918 translation nodes, SSE 0 -- every strided source point ends in a seeded voxel).  Reference visit
    order: pose <= 2e-3 rad / 2e-3, SSE, node counts within 2 %."""
    eng, g = _e2e(pkg, "spanner_sub", cloud("spanner_target"), cloud("spanner_source", 50), trans_batch=1, wide_children=0)
    _e2e_counts(eng, g)
    eng.registration.close()


def test_e2e_spanner_sub_wide(pkg):
    """The same through the default (widened, device-queue) search, and sharded over 2 thread ranks: SSE <= reference."""
    from cuda_go_icp_amd import sharded
    target, source = cloud("spanner_target"), cloud("spanner_source", 50)
    # The landscape is FLAT around this optimum: the target's noise (sigma 0.01 per axis) is half a voxel (0.0205), so a
    # whole neighbourhood of poses puts all 3 000 strided points into seeded voxels -- SSE exactly 0.  The reference's own
    # answer is one of them, 0.0317 rad / 0.0089 from the ground truth the two files define point by point (Kabsch over
    # all 150 000 pairs); another visit order returns another (measured: 0.031 rad from the reference's).  So the pose is
    # held to 8e-2 rad / 2e-2 of the reference's and, like the reference's, to 5e-2 rad / 1.5e-2 of the ground truth.
    Rgt, tgt = _kabsch(cloud("spanner_source").astype(np.float64), target.astype(np.float64))
    eng, g = _e2e(pkg, "spanner_sub", target, source, strict=False, wide_tol=(8e-2, 2e-2))
    assert rot_angle(np.array(g["R"]), Rgt) <= 5e-2 and np.linalg.norm(np.array(g["t"]) - tgt) <= 1.5e-2      # the reference
    assert rot_angle(eng.optR, Rgt) <= 5e-2 and np.linalg.norm(eng.optT - tgt) <= 1.5e-2
    eng.registration.close()
    engines = [pkg.FastGoICP(target, source, g["mse_threshold"]) for _ in range(2)]
    sharded.run_thread_ranks(engines, rot_pops_per_step=8)
    for e in engines:
        assert float(e.get_best_error()) <= 1.02 * g["sse"] + 1e-2 * g["sse_threshold"]
        assert rot_angle(e.optR, Rgt) <= 5e-2 and np.linalg.norm(e.optT - tgt) <= 1.5e-2
        e.registration.close()


def test_e2e_spanner_sparse_reference_order(pkg):
    """BASELINE configs[3] with a NON-ZERO reference SSE (tests/golden/e2e_spanner_sparse.json: every 8th target point, every 50th
    source point, mse 3e-4; the reference's Register: SSE 0.10996, 50 rotation / 3 150 translation nodes): the 2 % SSE bar that
    e2e_spanner_sub (SSE exactly 0) cannot exercise.  Reference visit order: pose <= 2e-3 rad / 2e-3, SSE 2 %, node counts 2 %."""
    eng, g = _e2e(pkg, "spanner_sparse", np.ascontiguousarray(cloud("spanner_target")[::8]), cloud("spanner_source", 50), trans_batch=1, wide_children=0)
    assert g["sse"] > 0.05 and abs(eng.get_best_error() - g["sse"]) <= 0.02 * g["sse"]          # no absolute slack here
    _e2e_counts(eng, g)
    eng.registration.close()


def test_e2e_spanner_sparse_wide(pkg):
    eng, g = _e2e(pkg, "spanner_sparse", np.ascontiguousarray(cloud("spanner_target")[::8]), cloud("spanner_source", 50), strict=False)
    eng.registration.close()


def _second_data_set(tag):
    if tag == "spanner":
        return cloud("spanner_target"), cloud("spanner_source", 50)
    from conftest import skull_problem
    target, source, _, _ = skull_problem()
    return target, np.ascontiguousarray(source[::10])


@pytest.mark.parametrize("tag", ["spanner", "skull"])
def test_reference_units_on_other_data_sets(pkg, oracle_mod, tag):
    """The other BASELINE data sets and a THREE-level box hierarchy against the reference's own values (oracle/gen_golden.py --sub-configs -> the
    harness's `units` on the noisy spanner, 150 000 target points, and on the skull scan, 98 359: tests/golden/{dt_lookup,nn,icp_iter,icp_dt_score}_<tag>.json):
      * geometry of the distance transform bit-exact, 4 096 DT3D::Distance samples <= 0.35 voxel (jly_3ddt.cpp:981-1026);
      * 4 096 nearest-neighbour squared distances of the reference's nanoflann tree BIT-EQUAL through the K = 3 hierarchy
        (nanoflann_goicp.hpp:1137-1184; until now K = 3 was held to brute force only);
      * ICP3D::Run trajectories with 1 / 2 / 10 forced iterations and to convergence from two start poses (jly_icp3d.hpp:181-295);
      * the DT-scored error of a pose (jly_goicp.cpp:93-132)."""
    tgt, src = _second_data_set(tag)
    reg = pkg.Registration(tgt, src, 1e-3, trans_batch=1, wide_children=0)
    g = golden("dt_lookup_" + tag)
    V, scale, origin = reg.dt_info()
    assert V == g["SIZE"] and scale == g["scale"] and origin == (g["xmin"], g["ymin"], g["zmin"])
    v = np.array(g["voxel"]).reshape(-1, 3)
    ref = np.array(g["voxel_distance"], dtype=np.float32)
    mine = reg.dt_download()[v[:, 2], v[:, 1], v[:, 0]]
    vox = 1.0 / g["scale"]
    assert np.all(mine <= ref + 1e-7) and np.max(ref - mine) <= 0.35 * vox
    g = golden("nn_" + tag)
    q = np.array(g["query"], np.float32).reshape(-1, 3)
    idx, d2 = reg.nn_query(q)
    assert np.array_equal(d2, np.array(g["dist_sq"], np.float32))
    assert np.mean(idx == np.array(g["index"])) > 0.999
    g = golden("icp_iter_" + tag)
    assert g["Nd"] == len(src)
    # Tolerances: 1e-4 for 1 and 2 iterations; 2e-3 on the pose and on the error from 10 on (SURVEY 8c's e2e bar).  The reference accumulates the
    # means and the 3 x 3 covariance of ~3 000 correspondences SEQUENTIALLY IN FLOAT (jly_icp3d.hpp:243-267: ~1e-5 relative of rounding noise that
    # the oracle's restatement reproduces to 1e-6 and a parallel sum cannot).  Measured, HIP against oracle along the skull's trajectories
    # (tools/icp_parity_probe.py): 7e-7 after one iteration, 6e-5 after two; from the first start pose it stays there (1e-5 at ten); from the
    # second it is amplified while the cloud moves fastest -- 4e-4 at five iterations, 1.1e-3 at seven, 6e-4 at ten (error 60.28 against 60.38)
    # -- and contracts again as both converge to the same fixed point: 9e-5 at twenty, 5e-6 at forty.  The bunny's fixture holds 1e-4 at ten.
    for c in g["cases"]:
        icp = pkg.IterativeClosestPoint3D(reg, c["max_iter"], c["err_diff"], c["R0"], c["t0"])
        err, R, t = icp.run()
        tol = 1e-4 if c["max_iter"] <= 2 else 2e-3
        dR, dt_ = np.abs(R.ravel() - np.array(c["R"])).max(), np.abs(t - np.array(c["t"])).max()
        print("icp %s max_iter %5d: max|dR| %.2e max|dt| %.2e err %.6g (reference %.6g)" % (tag, c["max_iter"], dR, dt_, err, c["err"]))
        assert dR <= tol and dt_ <= tol, (c["max_iter"], dR, dt_)
        assert abs(err - c["err"]) <= (1e-3 if c["max_iter"] <= 2 else 2e-3) * c["err"]
        if c["max_iter"] <= 10:
            assert icp.iters == c["max_iter"]
    g = golden("icp_dt_score_" + tag)
    sse = reg.compute_sse_error(np.array(g["R"]), np.array(g["t"]))
    assert abs(sse - g["dt_sse"]) <= 1e-4 * max(g["dt_sse"], 1e-3)
    reg.close()


@pytest.mark.parametrize("tag", ["spanner", "skull"])
def test_golden_inner_bnb_other_data_sets(pkg, tag):
    """The reference's InnerBnB on the spanner DT and on the skull DT (tests/golden/inner_bnb_<tag>.json): single expansions rel 1e-4 +
    arg-min child, full searches in the reference visit order value rel 1e-3, pops within 1 %."""
    g = golden("inner_bnb_" + tag)
    tgt, src = _second_data_set(tag)
    reg = pkg.Registration(tgt, src, 1e-3, trans_batch=1, wide_children=0)
    for case in g["cases"]:
        R = np.array(case["R"], np.float32)
        for s in case["single"]:
            px, py, pz, pw = map(np.float32, s["parent"])
            w = pw / np.float32(2)
            kids = []
            for j in range(8):
                cx = px + np.float32(j & 1) * w; cy = py + np.float32(j >> 1 & 1) * w; cz = pz + np.float32(j >> 2 & 1) * w
                kids.append([cx + w / np.float32(2), cy + w / np.float32(2), cz + w / np.float32(2), w, cx, cy, cz])
            kids = np.array(kids, np.float32)
            ub, _ = reg.eval_bounds(R, kids[:, :4], s["level"])
            j = int(np.argmin(ub))
            assert abs(ub[j] - s["min_ub"]) <= 1e-4 * max(s["min_ub"], 1e-3)
            if np.sum(np.abs(ub - ub[j]) <= 2e-5 * max(ub[j], 1e-3)) == 1:
                assert np.array_equal(kids[j, 4:7], np.array(s["best"][:3], np.float32))
        for s in case["full"]:
            v, best, cnt = reg.inner_bnb(R, s["level"], s["incumbent"])
            assert abs(v - s["value"]) <= 1e-3 * max(s["value"], 1e-3)
            assert abs(cnt.trans_pops - s["pops"]) <= max(2, 0.01 * s["pops"])
    reg.close()


def test_sharded_two_ranks_same_optimum(pkg, bunny_model, bunny_data10):
    """Rotation cubes dealt to 2 ranks (two engines on this GPU, one host thread each, the LIBRARY's protocol over its
    in-process communicator), bulk-synchronous and with the one-step-stale exchange: both reach the single-rank
    optimum (SURVEY 8e invariant) and end with the same global best."""
    from cuda_go_icp_amd import sharded
    g = golden("e2e_bunny10")
    for stale in (False, True):
        engines = [pkg.FastGoICP(bunny_model, bunny_data10, g["mse_threshold"]) for _ in range(2)]
        stats = sharded.run_thread_ranks(engines, rot_pops_per_step=4, stale=stale)
        sse = [float(e.get_best_error()) for e in engines]
        assert sse[0] == sse[1] and all(s["status"] == 0 for s in stats)
        for e in engines:
            _pose_close("sharded2 stale=%d" % stale, e.optR, e.optT, g, POSE_TOL, sse=sse[0])
            assert sse[0] <= 1.02 * g["sse"] and sse[0] < g["sse_threshold"]
            e.registration.close()


# ----------------------------------------------------------------------------------------------
# full-size, size-independent properties (BASELINE sizes; the oracle would take minutes here)
# ----------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def s1(pkg):
    from cuda_go_icp_amd import synth
    return synth.make_pair(**{k: synth.S1[k] for k in ("seed", "M", "N")})


def test_fullsize_additivity_and_monotonicity(pkg, s1):
    """N = 40 000, V = 300: (i) bounds are sums over points -> evaluating the cloud concatenated with
    itself doubles them; (ii) lb <= ub; (iii) at a fixed centre lb grows as the cube shrinks and the
    radius level deepens; (iv) Morton order is only a permutation."""
    target, source, _, _ = s1
    rng = np.random.default_rng(1)
    cubes = _cubes(rng, 256)
    R = pkg.fgoicp.rodrigues([0.4, -0.3, 0.8])
    a = pkg.Registration(target, source, 1e-3)
    b = pkg.Registration(target, np.concatenate([source, source]), 1e-3)
    c = pkg.Registration(target, source, 1e-3, morton_sort=0)
    for level in (-1, 4):
        ua, la = a.eval_bounds(R, cubes, level)
        ub2, lb2 = b.eval_bounds(R, cubes, level)
        uc, lc = c.eval_bounds(R, cubes, level)
        assert np.allclose(ub2, 2 * ua, rtol=1e-5) and np.allclose(lb2, 2 * la, rtol=1e-5, atol=1e-6)
        assert np.allclose(uc, ua, rtol=1e-5) and np.allclose(lc, la, rtol=1e-5, atol=1e-6)
        assert np.all(la <= ua)
    centre = np.tile(np.array([[0.1, -0.05, 0.2]], np.float32), (6, 1))
    ws = np.array([[1.0], [0.5], [0.25], [0.125], [0.0625], [0.03125]], np.float32)
    _, lbs = a.eval_bounds(R, np.concatenate([centre, ws], 1), -1)
    assert np.all(np.diff(lbs) >= 0)
    ubs = [a.eval_bounds(R, np.concatenate([centre[:1], ws[3:4]], 1), lv)[0][0] for lv in (2, 4, 6, 8, -1)]
    assert all(x <= y * (1 + 1e-6) for x, y in zip(ubs, ubs[1:]))
    for r in (a, b, c):
        r.close()


def test_fullsize_registration_recovers_ground_truth(pkg, s1):
    """S1 (N = M = 40 000): the known rigid motion is recovered (noise sigma 0.002 + DT quantisation
    -> SSE ~ 1-2, below SSEThresh = 4)."""
    target, source, Rgt, tgt = s1
    eng = pkg.FastGoICP(target, source, 1e-4)     # SSEThresh = 4.0: only the true basin gets below it
    eng.run()
    assert rot_angle(eng.optR, Rgt) <= 5e-3 and np.linalg.norm(eng.optT - tgt) <= 5e-3
    assert eng.get_best_error() < eng.sse_threshold


def test_skull_scan_known_motion(pkg, oracle_mod):
    """BASELINE configs[2] (skull_goicp.toml: larger cloud, k-d NN path stressed), full size.  The config's target
    model_skull.ply is missing from the reference checkout, so -- as SURVEY 8d prescribes -- the problem is
    built from the scan the reference does hold (data_skull.ply, 98 359 points, resize 0.01; committed
    fixture): target = the scan, source = a seeded 30 % subsample moved by a known rigid motion + N(0, 1e-3)
    noise (conftest.skull_problem).  The motion must be recovered.  Parity with the reference on this problem is pinned
    by test_e2e_skull_sub_* (the reference's own GoICP::Register on every 10th source point); at full size the bar is the
    ground truth."""
    from conftest import skull_problem
    target, source, Rgt, tgt = skull_problem()
    eng = pkg.FastGoICP(target, source, 1e-3)
    eng.run()
    assert eng.finished and eng.get_best_error() < eng.sse_threshold
    assert rot_angle(eng.optR, Rgt) <= 5e-3 and np.linalg.norm(eng.optT - tgt) <= 5e-3
    # the exact NN operator on the same hierarchy (98 359 points: 8 192 leaves, three box levels)
    rng = np.random.default_rng(99)
    q = np.concatenate([source[:1500], rng.uniform(-1.5, 1.5, (500, 3)).astype(np.float32)])
    idx, d2 = eng.registration.nn_query(q)
    bi, bd = oracle_mod.nn_brute(target, q)
    assert np.array_equal(d2, bd) and np.array_equal(idx, bi)


def test_fullsize_nn_idempotent(pkg, s1):
    target, source, _, _ = s1
    reg = pkg.Registration(target, source[:64], 1e-3)
    idx, d2 = reg.nn_query(target)                       # every target point is its own nearest neighbour
    assert np.all(d2 == 0)
    assert np.all(np.all(target[idx] == target, axis=1))
    reg.close()


# ----------------------------------------------------------------------------------------------
# error behaviour of the boundary
# ----------------------------------------------------------------------------------------------
def test_invalid_arguments(pkg, bunny_model, bunny_data10):
    with pytest.raises(pkg.GoicpError):
        pkg.Registration(np.zeros((0, 3), np.float32), bunny_data10)
    with pytest.raises(pkg.GoicpError):
        pkg.Registration(bunny_model, bunny_data10, dt_size=4)
    bad = bunny_data10.copy(); bad[7, 1] = np.nan
    with pytest.raises(pkg.GoicpError) as e:
        pkg.Registration(bunny_model, bad)
    assert e.value.code == -1 and "non-finite" in str(e.value)
    with pytest.raises(pkg.GoicpError):
        pkg.Registration(np.tile(bunny_model[:1], (10, 1)), bunny_data10)            # zero-extent target
    with pytest.raises(pkg.GoicpError):
        pkg.Registration(bunny_model, bunny_data10, trim_fraction=1.0)
    reg = pkg.Registration(bunny_model[:5], bunny_data10[:1], 1e-3, dt_size=32)     # tiny clouds are legal
    ub, lb = reg.eval_bounds(np.eye(3), np.array([[0, 0, 0, 0.5]], np.float32), -1)
    assert np.isfinite(ub[0]) and lb[0] <= ub[0]
    idx, d2 = reg.nn_query(bunny_model[:5])
    assert np.array_equal(idx, np.arange(5)) and np.all(d2 == 0)
    reg.close()


# ----------------------------------------------------------------------------------------------
# trimming: GoICP::trimFraction > 0
# ----------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def reg10_trim(pkg, bunny_model, bunny_data10):
    r = pkg.Registration(bunny_model, bunny_data10, 1e-3, trans_batch=1, wide_children=0, trim_fraction=0.1)
    yield r
    r.close()


def test_trimmed_bounds_vs_oracle(pkg, oracle_mod, oracle_dt_bunny, bunny_data10, rho10, reg10_trim):
    """k-th-smallest selection on the GPU (radix select) vs sort-based selection in the oracle: rel 1e-4."""
    k = int(len(bunny_data10) * (1 - np.float32(0.1)))
    rng = np.random.default_rng(9)
    R = pkg.fgoicp.rodrigues([0.3, -0.2, 0.9])
    prot = oracle_mod.rotate(R, bunny_data10)
    cubes = _cubes(rng, 40)
    for level in (-1, 5):
        ub, lb = reg10_trim.eval_bounds(R, cubes, level)
        for i, c in enumerate(cubes):
            oub, olb = oracle_mod.cube_bound_trim(oracle_dt_bunny, prot, rho10[level] if level >= 0 else None, c[:3], c[3], k)
            assert abs(ub[i] - oub) <= 1e-4 * max(oub, 1e-3) and abs(lb[i] - olb) <= 1e-4 * max(olb, 1e-3)
    I, Z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    assert abs(reg10_trim.compute_sse_error(I, Z) - oracle_mod.dt_sse_trim(oracle_dt_bunny, bunny_data10, I, Z, k)) <= 1e-4 * 100


def test_trimmed_inner_bnb_golden(pkg, reg10_trim):
    """The reference's own trimmed InnerBnB (trimFraction 0.1 set in the harness)."""
    g = golden("inner_bnb_trim")
    for case in g["cases"]:
        R = np.array(case["R"], np.float32)
        for s in case["full"]:
            v, best, cnt = reg10_trim.inner_bnb(R, s["level"], s["incumbent"])
            assert abs(v - s["value"]) <= 1e-3 * max(s["value"], 1e-3)
            assert abs(cnt.trans_pops - s["pops"]) <= max(2, 0.01 * s["pops"])
            if s["level"] < 0:
                assert np.array_equal(best, np.array(s["best"], np.float32))


def test_trimmed_icp_vs_oracle(pkg, oracle_mod, bunny_model, bunny_data10, reg10_trim):
    """Trimmed ICP (the num nearest correspondences; means over num -- the reference's /n is App. B-12,
    so this one is checked against the oracle's definition: parity with the reference unpinned)."""
    k = int(len(bunny_data10) * (1 - np.float32(0.1)))
    kd = oracle_mod.KdTree(bunny_model)
    for iters in (1, 3, 10):
        err, R, t = pkg.IterativeClosestPoint3D(reg10_trim, iters, 1e-9).run()
        oerr, oR, ot, _ = kd.icp_run_trim(bunny_data10, k, np.eye(3), np.zeros(3), iters, 1e-9)
        assert abs(err - oerr) <= 1e-3 * oerr and np.abs(R - oR).max() <= 2e-4 and np.abs(t - ot).max() <= 2e-4


def test_trimmed_e2e_vs_oracle(pkg, oracle_mod, oracle_dt_bunny, bunny_model, bunny_data10):
    eng = pkg.FastGoICP(bunny_model, bunny_data10, 1e-3, trim_fraction=0.1, trans_batch=1, wide_children=0)
    eng.run()
    # the oracle's trimmed registration of the same clouds takes ~50 s of CPU: its result is a committed fixture
    # (tests/golden/e2e_bunny10_trim_oracle.json, written by oracle/gen_oracle_fixtures.py; the slow CPU test
    # test_oracle_trim_fixture_is_current re-derives it)
    o = golden("e2e_bunny10_trim_oracle")
    assert rot_angle(eng.optR, o["R"]) <= 2e-3 and np.linalg.norm(eng.optT - o["t"]) <= 2e-3
    assert abs(eng.get_best_error() - o["sse"]) <= 0.02 * o["sse"]
    assert eng.get_best_error() < eng.registration.params.mse_threshold * int(len(bunny_data10) * 0.9) * 1.0001


def test_trimmed_tiny_proven_optimum_vs_oracle(pkg, oracle_mod):
    """Trimming (GoICP::trimFraction, jly_goicp.cpp:201,293-315; the reference hard-wires 0) on a CONVERGED search: conftest.tiny_problem(1) at
    trim_fraction 0.1 (the 54 best of 60 points), mse 4e-3 (SSEThresh 0.216 under the trimmed optimum's 0.247: 2.3 k rotation / 0.27 M translation nodes) -- the oracle's trimmed registration (seconds of CPU, run here) against the engine in
    the reference visit order: same optimum, rotation nodes within 0.5 %, translation nodes within 1 % (measured: 2 313 = 2 313 and 270 629 =
    270 629); the default search an optimum at least as good, within SSEThresh of it."""
    from conftest import tiny_problem
    tgt, src = tiny_problem(1)
    dt = oracle_mod.DistanceTransform(tgt, 300, 2.0)
    o = oracle_mod.register(dt, tgt, src, 4e-3, trim_fraction=0.1)
    eng = pkg.FastGoICP(tgt, src, 4e-3, trim_fraction=0.1, trans_batch=1, wide_children=0)
    eng.run()
    c = eng.counters
    sse = float(eng.get_best_error())
    print("trimmed tiny1: sse %.7g (oracle %.7g) rotation nodes %d (oracle %d) translation nodes %d (oracle %d) rot_error %.2e" % (
        sse, o["sse"], c.rot_pops, o["rot_pops"], c.trans_pops, o["trans_pops"], rot_angle(eng.optR, o["R"])))
    assert o["rot_pops"] > 500 and o["sse"] > eng.sse_threshold                # converged, not an early exit
    assert abs(sse - o["sse"]) <= 1e-4 * o["sse"] and rot_angle(eng.optR, o["R"]) <= 1e-4 and np.linalg.norm(eng.optT - o["t"]) <= 1e-4
    assert abs(c.rot_pops - o["rot_pops"]) <= max(2, 0.005 * o["rot_pops"]) and abs(c.trans_pops - o["trans_pops"]) <= 0.01 * o["trans_pops"]
    eng.registration.close()
    # the default (widened) search: Go-ICP guarantees the error to within SSEThresh, not the pose -- another visit order may end in another optimum
    # inside that band (measured: 0.2430613, the optimum the oracle itself finds at mse 3.5e-3, against the strict order's 0.2468403)
    eng = pkg.FastGoICP(tgt, src, 4e-3, trim_fraction=0.1)
    eng.run()
    w = float(eng.get_best_error())
    print("trimmed tiny1, default mode: sse %.7g" % w)
    assert w <= o["sse"] * (1 + 1e-4) and o["sse"] - w <= eng.sse_threshold
    cw, Rw, tw = eng.counters, eng.optR.copy(), eng.optT.copy()
    eng.registration.close()
    # ... and the same search cut into four lanes per batch (trimmed bounds: one workgroup per expansion, own lists per lane): the same optimum
    eng = pkg.FastGoICP(tgt, src, 4e-3, trim_fraction=0.1, lanes=4, lane_min_searches=2)
    eng.run()
    cl = eng.counters
    print("trimmed tiny1, four lanes: sse %.7g, cube bounds %d vs %d, two-or-more-lane batches %d" % (eng.get_best_error(), cl.cubes, cw.cubes, cl.lane_batches))
    assert cl.lane_batches > 0 and abs(float(eng.get_best_error()) - w) <= 1e-6 * w
    assert rot_angle(eng.optR, Rw) <= 1e-6 and np.linalg.norm(eng.optT - tw) <= 1e-6
    assert cl.rot_pops == cw.rot_pops and abs(cl.cubes - cw.cubes) <= 0.01 * cw.cubes
    eng.registration.close()


# ----------------------------------------------------------------------------------------------
# BASELINE configs[0]: plain ICP (modes 0-2), step API, bunny-scale clouds
# ----------------------------------------------------------------------------------------------
def test_plain_icp_steps_vs_oracle(pkg, oracle_mod, s1):
    """ICP::kdTreeGPUStep x 20 on a 40 000 / 40 000 pair (the bun000/bun045 PLYs of bunny_icp.toml cannot
    travel; S1 is their synthetic twin): every step = one oracle ICP iteration with fresh
    means from the accumulated pose.  1e-4 abs on R, t after 20 steps."""
    target, source, Rgt, tgt = s1
    # source = the target under a small known motion: plain ICP converges without the global search
    src = (target @ pkg.fgoicp.rodrigues([0.05, -0.04, 0.03]).astype(np.float32) + np.float32(0.01)).astype(np.float32)
    reg = pkg.Registration(target, src, 1e-5, dt_size=128)
    kd = oracle_mod.KdTree(target)
    R, t = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    for _ in range(20):
        snap = reg.icp_step()
        _, R, t, _ = kd.icp_run(src, R, t, 1, -1e30)
    assert np.abs(np.array(snap.curR, np.float32).reshape(3, 3) - R).max() <= 1e-4
    assert np.abs(np.array(snap.curT, np.float32) - t).max() <= 1e-4
    assert snap.best_sse < 1e-3 * len(src)                       # converged onto the target
    reg.close()


def test_cli_end_to_end(pkg, tmp_path):
    """goicp_cli with a reference-style .toml (TXT clouds, relative paths, output.toml + viz.ply)."""
    import subprocess
    from conftest import ROOT
    for name in ("model_rand", "data_rand"):
        pts = cloud(name)
        with open(tmp_path / (name + ".txt"), "w") as f:
            f.write("%d\n" % len(pts))
            for q in pts:
                f.write("%.9g %.9g %.9g\n" % tuple(q))
    (tmp_path / "cfg.toml").write_text(
        '[info]\ndescription = "cli test"\n[io]\ntarget = "model_rand.txt"\nsource = "data_rand.txt"\n'
        'output = "%s"\nvisualization = "%s"\n[params]\nmode = 4\nsubsample = 1.0\nmse_threshold = 1e-3\nresize = 1.0\n'
        % (tmp_path / "output.toml", tmp_path / "viz.ply"))
    exe = os.path.join(ROOT, "cuda-go-icp_amd", "goicp_cli")
    out = subprocess.run([exe, str(tmp_path / "cfg.toml")], check=True, capture_output=True, text=True, timeout=120).stdout
    g = golden("e2e_rand100")
    assert "Optimal Rotation Matrix" in out and "Optimal Translation Vector" in out
    txt = (tmp_path / "output.toml").read_text()
    sse = float([l for l in txt.splitlines() if l.startswith("sse =")][0].split("=")[1])
    assert sse <= 1.02 * g["sse"]
    assert pkg.load_cloud(tmp_path / "viz.ply").shape == (200, 3)
    bad = subprocess.run([exe, str(tmp_path / "missing.toml")], capture_output=True, text=True)
    assert bad.returncode == 1 and "error" in bad.stderr.lower()
    # a config that carries the [params.translation] table of the reference's files (+-1.0: a root of width 2, which the reference
    # itself never applies): with --reference-root the CLI searches the CPU path's roots and reproduces the range-less run node for
    # node; without the flag the range is applied (a larger root: more translation nodes).  --trim-fraction: the console's MSE is
    # over the inliers, like output.toml's
    (tmp_path / "cfg_ranges.toml").write_text((tmp_path / "cfg.toml").read_text() +
                                              "[params.translation]\nxmin = -1.0\nxmax = 1.0\nymin = -1.0\nymax = 1.0\nzmin = -1.0\nzmax = 1.0\nsearch_depth = 12\n")
    nodes = lambda o: int([l for l in o.splitlines() if l.startswith("Total Translation Nodes Searched")][0].split(":")[1])
    ref_root = subprocess.run([exe, str(tmp_path / "cfg_ranges.toml"), "--reference-root"], check=True, capture_output=True, text=True, timeout=120).stdout
    applied = subprocess.run([exe, str(tmp_path / "cfg_ranges.toml")], check=True, capture_output=True, text=True, timeout=120).stdout
    assert nodes(ref_root) == nodes(out) and nodes(applied) != nodes(out)
    trimmed = subprocess.run([exe, str(tmp_path / "cfg.toml"), "--trim-fraction", "0.1"], check=True, capture_output=True, text=True, timeout=120).stdout
    line = [l for l in trimmed.splitlines() if l.startswith("Searching over!")][0]
    best, mse = float(line.split("Best Error:")[1].split()[0]), float(line.split("MSE")[1].strip(" )"))
    assert abs(mse - best / int(100 * np.float32(0.9))) <= 1e-6 * max(mse, 1e-9)


def test_cli_with_roctx_ranges(pkg, tmp_path):
    """GOICP_ROCTX=1 (csrc/trace.hpp): the engine looks the roctx library up with dlopen and brackets its phases; with or
    without a profiler attached, with or without the library, the registration is the same."""
    import subprocess
    from conftest import ROOT
    for name in ("model_rand", "data_rand"):
        pts = cloud(name)
        with open(tmp_path / (name + ".txt"), "w") as f:
            f.write("%d\n" % len(pts))
            for q in pts:
                f.write("%.9g %.9g %.9g\n" % tuple(q))
    sses = []
    for tag, env in (("off", {}), ("on", {"GOICP_ROCTX": "1"})):
        out_toml = tmp_path / ("output_%s.toml" % tag)
        (tmp_path / "cfg.toml").write_text(
            '[info]\ndescription = "roctx test"\n[io]\ntarget = "model_rand.txt"\nsource = "data_rand.txt"\n'
            'output = "%s"\nvisualization = "%s"\n[params]\nmode = 4\nsubsample = 1.0\nmse_threshold = 1e-3\nresize = 1.0\n'
            % (out_toml, tmp_path / "viz.ply"))
        exe = os.path.join(ROOT, "cuda-go-icp_amd", "goicp_cli")
        subprocess.run([exe, str(tmp_path / "cfg.toml")], check=True, capture_output=True, text=True, timeout=120, env=dict(os.environ, **env))
        sses.append(float([l for l in out_toml.read_text().splitlines() if l.startswith("sse =")][0].split("=")[1]))
    assert sses[0] == sses[1]


# ----------------------------------------------------------------------------------------------
# device-side k-d tree build
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M", [35947, 120000, 700])
def test_gpu_built_hierarchy_exact(pkg, oracle_mod, bunny_model, M):
    """The Morton/rocPRIM-built box hierarchy gives the same exact neighbours as brute force (K = 1, 2, 3)."""
    from cuda_go_icp_amd import synth
    target = bunny_model if M == 35947 else synth.make_pair(seed=5, M=M, N=8)[0]
    reg = pkg.Registration(target, target[:256], 1e-3, dt_size=96, kd_gpu_build=1)
    rng = np.random.default_rng(8)
    near = target[300:700]
    q = np.concatenate([rng.uniform(-2.5, 2.5, (1500, 3)).astype(np.float32), target[:300],
                        (near + rng.normal(0, 0.01, near.shape)).astype(np.float32)])
    idx, d2 = reg.nn_query(q)
    bi, bd = oracle_mod.nn_brute(target, q)
    assert np.array_equal(d2, bd) and np.array_equal(idx, bi)
    reg.close()


def test_gpu_built_hierarchy_e2e(pkg, bunny_model, bunny_data10):
    _e2e(pkg, "bunny10", bunny_model, bunny_data10, strict=True, trans_batch=1, wide_children=0, kd_gpu_build=1)


# ----------------------------------------------------------------------------------------------
# result API under concurrency (the reference's viewer polls the worker: src/goicp_kernel.cu:161-177)
# ----------------------------------------------------------------------------------------------
def test_rerun_and_concurrent_engines(pkg, bunny_model, bunny_data10):
    """An engine can register again (same answer, bit for bit), and two engines of one process running at the
    same time on their own streams do not disturb each other (no shared mutable state in the library)."""
    import threading
    a = pkg.FastGoICP(bunny_model, bunny_data10, 1e-3)
    a.run()
    first = (a.get_best_error(), a.optR.copy(), a.optT.copy(), a.registration.poll().counters.cubes)
    a.run()
    again = (a.get_best_error(), a.optR.copy(), a.optT.copy(), a.registration.poll().counters.cubes)
    assert first[0] == again[0] and np.array_equal(first[1], again[1]) and np.array_equal(first[2], again[2]) and first[3] == again[3]
    engines = [pkg.FastGoICP(bunny_model, bunny_data10, 1e-3) for _ in range(3)]
    threads = [threading.Thread(target=e.run) for e in engines]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in engines:
        assert e.finished and e.get_best_error() == first[0] and np.array_equal(e.optR, first[1]) and np.array_equal(e.optT, first[2])


def test_poll_and_cancel_while_running(pkg, bunny_model, bunny_data):
    """goicp_poll from another thread returns consistent snapshots (best error never increases, optR stays
    a rotation) while goicp_register runs; goicp_cancel (the reference's `goicp_finished` flag) stops it."""
    import threading
    import time
    eng = pkg.FastGoICP(bunny_model, bunny_data, 1e-9, trans_batch=1, wide_children=0)   # threshold too tight to finish soon
    th = threading.Thread(target=eng.run)
    th.start()
    seen = []
    t0 = time.time()
    while time.time() - t0 < 1.5:
        r = eng.registration.poll()
        R = np.array(r.optR, np.float64).reshape(3, 3)
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-4
        seen.append(float(r.best_sse))
        assert not r.finished
        time.sleep(0.01)
    assert all(b <= a for a, b in zip(seen, seen[1:]))
    assert th.is_alive()
    eng.cancel()
    th.join(timeout=30)
    assert not th.is_alive() and eng.finished
    assert eng.get_best_error() <= seen[-1]
    # the default (device-queue) search with every batch cut into lanes: cancelled in the middle of multi-stream rounds it returns promptly,
    # and the engine registers again afterwards (streams drained, per-lane state reusable) with a result that is still an upper bound
    eng2 = pkg.FastGoICP(bunny_model, bunny_data, 1e-9, lanes=3, lane_min_searches=2)
    th = threading.Thread(target=eng2.run)
    th.start()
    time.sleep(0.5)
    assert th.is_alive()
    t1 = time.time()
    eng2.cancel()
    th.join(timeout=30)
    assert not th.is_alive() and eng2.finished and time.time() - t1 < 5.0
    first = float(eng2.get_best_error())
    assert eng2.counters.lane_batches > 0 and np.isfinite(first)
    th = threading.Thread(target=eng2.run)
    th.start()
    time.sleep(0.3)
    eng2.cancel()
    th.join(timeout=30)
    assert not th.is_alive() and np.isfinite(float(eng2.get_best_error()))
    eng2.registration.close()


# ----------------------------------------------------------------------------------------------
# round 2: point-by-point lookup, device SVD, the two remaining BASELINE configs, search ranges
# ----------------------------------------------------------------------------------------------
def test_dt_lookup_point_by_point(pkg, oracle_dt_bunny, bunny_model):
    """SURVEY a-1, point by point: an engine whose source is the single point (0,0,0) makes
    goicp_eval_bounds(I, cube centre q, w = 0, fix_rot) return Distance(q)^2 exactly, so the 4 096 reference
    DT3D::Distance samples of tests/golden/dt_lookup.json (jly_3ddt.cpp:981-1026; 1 024 of them around index 0 /
    V-1, out-of-grid and negative-overshoot cases included) reach the HIP lookup one by one.  The golden queries
    are doubles; the engine takes floats, so: bit-equal to the oracle at the float-rounded query, and against the
    reference's own values <= 0.35 voxel wherever the rounding did not move the query into a neighbouring voxel."""
    g = golden("dt_lookup")
    q64 = np.array(g["query"]).reshape(-1, 3)
    qf = q64.astype(np.float32)
    ref = np.array(g["distance"], dtype=np.float32)
    vox = 1.0 / g["scale"]
    for layout in (1, 0):
        reg = pkg.Registration(bunny_model, np.zeros((1, 3), np.float32), 1e-3, dt_layout=layout)
        cubes = np.concatenate([qf, np.zeros((len(qf), 1), np.float32)], 1)
        ub, lb = reg.eval_bounds(np.eye(3), cubes, -1)
        d = oracle_dt_bunny.distance(qf.astype(np.float64)).astype(np.float32)
        assert np.array_equal(ub, d * d), "HIP lookup differs from the oracle's DT3D::Distance restatement"
        assert np.array_equal(lb, ub)                                    # w = 0: no translation radius
        got = np.sqrt(ub.astype(np.float64))
        same_voxel = np.all(np.floor((q64 - [g["xmin"], g["ymin"], g["zmin"]]) * g["scale"] + 0.5)
                            == np.floor((qf.astype(np.float64) - [g["xmin"], g["ymin"], g["zmin"]]) * g["scale"] + 0.5), axis=1)
        assert same_voxel.mean() > 0.99
        assert np.all(got[same_voxel] <= ref[same_voxel] + 1e-6) and np.max(ref[same_voxel] - got[same_voxel]) <= 0.35 * vox
        assert np.mean(np.abs(got[same_voxel] - ref[same_voxel]) <= 1e-6) > 0.999          # index math exact
        assert np.max(np.abs(got - ref)) <= (np.sqrt(3) + 0.35) * vox                       # a neighbouring voxel at worst
        reg.close()


def test_device_svd_golden(pkg):
    """SURVEY a-6: the device-side Kabsch / SVD routine (one-sided Jacobi in fp64, device.hip kabsch_rows: the routine the ICP finalize runs, three lanes wide)
    fed the reference's own Matrix::svd cases (tests/golden/svd3x3.json: 32 H, half of them x1000 scale),
    R_ = V diag(1,1,det(V U^T)) U^T as jly_icp3d.hpp:266-285.  1e-5 as SURVEY 8c-6."""
    import ctypes as C
    lib = pkg.load_library()
    worst = 0.0
    for c in golden("svd3x3")["cases"]:
        H = np.array(c["H"], np.float32)
        R = np.empty(9, np.float32)
        pkg.binding.check(lib.goicp_debug_kabsch(H.ctypes.data_as(C.POINTER(C.c_float)), R.ctypes.data_as(C.POINTER(C.c_float))))
        worst = max(worst, np.abs(R - np.array(c["R"], np.float32)).max())
        assert abs(np.linalg.det(R.reshape(3, 3).astype(np.float64)) - 1) <= 1e-5
    assert worst <= 1e-5, worst


def _kabsch(src, dst):
    ms, md = src.mean(0), dst.mean(0)
    U, _, Vt = np.linalg.svd((src - ms).T @ (dst - md))
    R = Vt.T @ np.diag([1, 1, np.linalg.det(Vt.T @ U.T)]) @ U.T
    return R, md - R @ ms


def test_spanner_noisy(pkg, oracle_mod):
    """BASELINE configs[3], test/spanner_goicp.toml:10-20: target noisy_flipped_model_spanner.ply (150 000 points,
    sigma 0.5 * resize 0.02 = 0.01 noise per axis), resize 0.02, mse_threshold 1e-4 -> SSEThresh 15.  The config's source
    model_spanner.ply is missing from the reference checkout (.MISSING_LARGE_BLOBS:4); SURVEY 8d's substitute
    rotated_model_spanner.ply (the model under a random rotation, 150 000 points) is used.  Parity with the reference on
    this pair is pinned by test_e2e_spanner_sub_* and test_golden_inner_bnb_spanner (the reference's own GoICP::Register /
    InnerBnB on the full target and every 50th source point); at full size (days of reference CPU time) the bar is the
    ground truth, which the two files define through their point-by-point correspondence.  Checked: single engine below SSEThresh at the true pose;
    2 and 4 sharded engines reach the same optimum (SURVEY 8e invariant); exact NN on this hierarchy; cube bounds
    against the oracle on a 1/10 subsample."""
    from cuda_go_icp_amd import sharded
    target, source = cloud("spanner_target"), cloud("spanner_source")
    assert target.shape == source.shape == (150000, 3)
    Rgt, tgt = _kabsch(source.astype(np.float64), target.astype(np.float64))
    eng = pkg.FastGoICP(target, source, 1e-4)
    assert abs(float(eng.sse_threshold) - 15.0) < 1e-3
    eng.run()
    sse1 = float(eng.get_best_error())
    assert eng.finished and sse1 < eng.sse_threshold
    # the spanner is flat and elongated: a 180-degree flip about its long axis is a deep local minimum, far above the threshold
    assert rot_angle(eng.optR, Rgt) <= 1e-2 and np.linalg.norm(eng.optT - tgt) <= 5e-3
    c1 = eng.counters
    assert c1.cubes > 0 and c1.icp_iters > 0
    # NN exact vs brute force on this hierarchy (150 000 points: three box levels)
    rng = np.random.default_rng(7)
    q = np.concatenate([source[rng.choice(len(source), 1500, replace=False)] @ Rgt.T.astype(np.float32) + tgt.astype(np.float32),
                        rng.uniform(-1.6, 1.6, (500, 3)).astype(np.float32)]).astype(np.float32)
    idx, d2 = eng.registration.nn_query(q)
    bi, bd = oracle_mod.nn_brute(target, q)
    assert np.array_equal(d2, bd) and np.array_equal(idx, bi)
    eng.registration.close()
    # sharded: the rotation cubes dealt to 2 / 4 engines on this GPU, the library's protocol (csrc/shard.cpp) over its
    # in-process communicator, one host thread per rank
    for world, stale in ((2, False), (4, False), (4, True)):
        engines = [pkg.FastGoICP(target, source, 1e-4) for _ in range(world)]
        stats = sharded.run_thread_ranks(engines, rot_pops_per_step=8, stale=stale)
        best = [float(e.get_best_error()) for e in engines]
        assert max(best) == min(best) and all(s["status"] == 0 for s in stats)
        assert best[0] < engines[0].sse_threshold and abs(best[0] - sse1) <= 0.05 * sse1
        assert all(rot_angle(e.optR, Rgt) <= 1e-2 and np.linalg.norm(e.optT - tgt) <= 5e-3 for e in engines)
        for e in engines:
            e.registration.close()
    # cube bounds vs the oracle on every 10th source point (same DT: V = 300 over the noisy target)
    sub = np.ascontiguousarray(source[::10])
    reg = pkg.Registration(target, sub, 1e-4)
    dt = oracle_mod.DistanceTransform(target, 300, 2.0)
    assert np.array_equal(reg.dt_download(), dt.grid())
    R = pkg.fgoicp.rodrigues([2.5, -0.8, 1.2])
    prot = oracle_mod.rotate(R, sub)
    _, rho = oracle_mod.rot_radii(sub)
    cubes = _cubes(rng, 48)
    for level in (-1, 4):
        ub, lb = reg.eval_bounds(R, cubes, level)
        for i, c in enumerate(cubes):
            oub, olb = oracle_mod.cube_bound(dt, prot, rho[level] if level >= 0 else None, c[:3], c[3])
            assert abs(ub[i] - oub) <= 1e-4 * max(oub, 1e-3) and abs(lb[i] - olb) <= 1e-4 * max(olb, 1e-3), (i, level)
    reg.close()


S2_AMP = 0.15        # relief of the synthetic surface: low enough that ICP's basin is small and the BnB has to dig
S2_OVER_FLOOR = 1.2  # mse_threshold = 1.2 x the measured floor of the true pose


def test_s2_fullsize(pkg, oracle_mod):
    """BASELINE configs[4]: synthetic S2, N = M = 1 000 000, DT 512^3 (537 MB, HBM-resident), full SE(3) BnB.
    The oracle would take days here, so: size-independent properties (additivity over a split of the cloud,
    lb <= ub, monotone lb along a shrinking cube), exact NN on a sample against brute force, idempotent NN, and a
    registration that makes the outer BnB work: the surface relief is lowered to 0.15 (with synth.S2's default 0.35 the
    first rotation expansion already hands ICP the true basin: 1 rotation node), and mse_threshold is set from the
    MEASURED noise floor -- 1.2 x the DT-scored MSE of the ground-truth pose -- so neither the initial ICP (3 rad away)
    nor a shallow local minimum of this near-spherical shape satisfies it.  Measured on MI355X: 128 rotation nodes,
    ~44 k cube bounds, ~1 400 ICP iterations, 1.2 s.  The landscape around the optimum is flat (relief 0.15, noise sigma
    0.002, voxel 0.0065): poses within 1.2 x floor scatter ~0.02 rad, hence the 3e-2 rad / 1e-2 tolerances."""
    from cuda_go_icp_amd import synth
    target, source, Rgt, tgt = synth.make_pair(seed=synth.S2["seed"], M=synth.S2["M"], N=synth.S2["N"], amp=S2_AMP)
    V = synth.S2["V"]
    reg = pkg.Registration(target, source, 1e-3, dt_size=V)
    floor = float(reg.compute_sse_error(Rgt, tgt)) / len(source)
    assert 1e-6 < floor < 5e-5
    rng = np.random.default_rng(11)
    cubes = _cubes(rng, 64)
    R = pkg.fgoicp.rodrigues([0.4, -0.3, 0.8])
    half_a = pkg.Registration(target, source[:500000], 1e-3, dt_size=V)
    half_b = pkg.Registration(target, source[500000:], 1e-3, dt_size=V)
    for level in (-1, 5):
        u, l = reg.eval_bounds(R, cubes, level)
        ua, la = half_a.eval_bounds(R, cubes, level)
        ub_, lb_ = half_b.eval_bounds(R, cubes, level)
        assert np.allclose(u, ua + ub_, rtol=2e-5) and np.allclose(l, la + lb_, rtol=2e-5, atol=1e-5)
        assert np.all(l <= u)
    half_a.close(); half_b.close()
    centre = np.tile(np.array([[0.1, -0.05, 0.2]], np.float32), (6, 1))
    ws = np.array([[1.0], [0.5], [0.25], [0.125], [0.0625], [0.03125]], np.float32)
    _, lbs = reg.eval_bounds(R, np.concatenate([centre, ws], 1), -1)
    assert np.all(np.diff(lbs) >= 0)
    q = np.concatenate([source[:300], rng.uniform(-1.2, 1.2, (100, 3)).astype(np.float32)])
    idx, d2 = reg.nn_query(q)
    bi, bd = oracle_mod.nn_brute(target, q)
    assert np.array_equal(d2, bd) and np.array_equal(idx, bi)
    idx, d2 = reg.nn_query(target[:20000])
    assert np.all(d2 == 0) and np.all(np.all(target[idx] == target[:20000], axis=1))
    reg.close()
    eng = pkg.FastGoICP(target, source, S2_OVER_FLOOR * floor, dt_size=V)
    eng.run()
    c = eng.counters
    assert eng.finished and eng.get_best_error() < eng.sse_threshold
    assert c.rot_pops >= 50, "the outer BnB did not run (%d rotation nodes)" % c.rot_pops
    assert rot_angle(eng.optR, Rgt) <= 3e-2 and np.linalg.norm(eng.optT - tgt) <= 1e-2
    eng.registration.close()


def test_search_ranges_applied(pkg, bunny_model, bunny_data10):
    """[params.rotation] / [params.translation] / search_depth (test/skull_goicp.toml:22-41) reach the
    engine.  (i) the full +-180 degree range and the CPU path's translation cube reproduce the default search node for
    node; (ii) a rotation + translation box around the known optimum finds the same optimum with no more rotation
    nodes; (iii) a box that excludes the optimum (ICP refinement off, which may leave any box) does not report it;
    (iv) a depth limit bounds the number of nodes."""
    g = golden("e2e_bunny10")
    base = pkg.FastGoICP(bunny_model, bunny_data10, g["mse_threshold"], trans_batch=1, wide_children=0)
    base.run()
    full = pkg.FastGoICP(bunny_model, bunny_data10, g["mse_threshold"], trans_batch=1, wide_children=0,
                         use_rot_range=1, rot_min=[-180] * 3, rot_max=[180] * 3,
                         use_trans_range=1, trans_min=[-0.5] * 3, trans_max=[0.5] * 3)
    full.run()
    assert full.counters.rot_pops == base.counters.rot_pops and full.counters.trans_pops == base.counters.trans_pops
    assert full.get_best_error() == base.get_best_error()
    from scipy.spatial.transform import Rotation
    rv = np.degrees(Rotation.from_matrix(np.array(g["R"]).reshape(3, 3)).as_rotvec())
    t0 = np.array(g["t"])
    box = pkg.FastGoICP(bunny_model, bunny_data10, g["mse_threshold"], use_rot_range=1, rot_min=list(rv - 25), rot_max=list(rv + 20),
                        use_trans_range=1, trans_min=list(t0 - 0.2), trans_max=list(t0 + 0.25))
    box.run()
    assert box.get_best_error() <= 1.02 * g["sse"]
    _pose_close("search box", box.optR, box.optT, g, POSE_TOL)
    assert 0 < box.counters.rot_pops <= base.counters.rot_pops
    away = pkg.FastGoICP(bunny_model, bunny_data10, g["mse_threshold"], use_rot_range=1,
                         rot_min=list(-rv - 15), rot_max=list(-rv + 15), icp_max_iter=0, rot_search_depth=4, trans_search_depth=6)   # depth caps bound the run: without the optimum in reach the gap never closes
    away.run()
    assert rot_angle(away.optR, np.array(g["R"])) > 0.3 or away.get_best_error() > 1.5 * g["sse"]
    shallow = pkg.FastGoICP(bunny_model, bunny_data10, 1e-6, use_rot_range=1, rot_min=[-180] * 3, rot_max=[180] * 3, rot_search_depth=2,
                            use_trans_range=1, trans_min=[-0.5] * 3, trans_max=[0.5] * 3, trans_search_depth=3)
    shallow.run()
    assert shallow.finished and shallow.counters.rot_pops <= 1 + 8 + 64
    for e in (base, full, box, away, shallow):
        e.registration.close()


def test_search_range_with_a_degenerate_axis(pkg, bunny_model):
    """A search range of zero width on an axis (rotation about z only; translation with z fixed) must not cull the whole
    search: centred, the fixed value would sit exactly on the first split plane and the strict box test would drop both
    children (the search would end after one pop with the start pose).  Problem: the bunny model against a copy of itself
    rotated by 0.7 rad about z and shifted in the xy-plane; ICP refinement off, so only the BnB can find the pose."""
    rng = np.random.default_rng(5)
    sub = bunny_model[rng.choice(len(bunny_model), 3000, replace=False)].astype(np.float64)
    c, s_ = np.cos(0.7), np.sin(0.7)
    Rgt, tgt = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]]), np.array([0.10, -0.05, 0.0])
    source = ((sub - tgt) @ Rgt).astype(np.float32)                      # target = Rgt source + tgt
    for dq in (1, 0):                                                    # device queues and host queues
        eng = pkg.FastGoICP(bunny_model, source, 1e-3, icp_max_iter=0, device_queues=dq,
                            use_rot_range=1, rot_min=[0, 0, -180], rot_max=[0, 0, 180],
                            use_trans_range=1, trans_min=[-0.5, -0.5, 0], trans_max=[0.5, 0.5, 0])
        eng.run()
        assert eng.finished and eng.counters.rot_pops > 1 and eng.counters.cubes > 64
        # without ICP the search stops once best - min lb <= SSEThresh: the best cube centre is within SSEThresh of the optimum (~0)
        assert eng.get_best_error() < 2 * eng.sse_threshold, eng.get_best_error()
        # the reported pose is the centre of the best cube (no ICP): half a cube width off the plane and off the truth
        assert rot_angle(eng.optR, Rgt) <= 0.25 and np.linalg.norm(eng.optT - tgt) <= 0.08, (rot_angle(eng.optR, Rgt), eng.optT)
        assert abs(eng.optT[2]) <= 0.02 and eng.optR[2, 2] >= 0.99                # ... but within the last cubes of the plane
        eng.registration.close()
    # a point range: one cube, evaluated and done
    one = pkg.FastGoICP(bunny_model, source, 1e-3, icp_max_iter=0, use_rot_range=1, rot_min=[0, 0, 40], rot_max=[0, 0, 40],
                        use_trans_range=1, trans_min=[0.1, -0.05, 0], trans_max=[0.1, -0.05, 0])
    one.run()
    assert one.finished
    one.registration.close()


def test_flow_with_a_wide_rotation_batch(pkg, bunny_model, bunny_data10):
    """flow > 0 with rot_batch beyond the flow mode's 2 048 search slots / 16 used to spin forever (nothing could be
    admitted even with every slot free): the admission width is clamped, the registration ends with the usual optimum."""
    g = golden("e2e_bunny10")
    eng = pkg.FastGoICP(bunny_model, bunny_data10, g["mse_threshold"], flow=8, rot_batch=256)
    eng.run()
    assert eng.finished and eng.get_best_error() <= 1.02 * g["sse"] and eng.get_best_error() < g["sse_threshold"]
    _pose_close("flow rot_batch 256", eng.optR, eng.optT, g, POSE_TOL, sse=float(eng.get_best_error()))
    eng.registration.close()


def test_progress_callback_and_device(pkg, bunny_model, bunny_data10):
    """goicp_set_progress_callback: snapshots arrive on the registering thread, monotone in best_sse, the last one
    finished; goicp_device reports the ordinal every entry point re-establishes (engine created in one thread,
    driven from another)."""
    import ctypes as C
    import threading
    eng = pkg.FastGoICP(bunny_model, bunny_data10, 1e-3, device=0)
    seen = []
    CB = C.CFUNCTYPE(None, C.POINTER(pkg.binding.CResult), C.c_void_p)
    cb = CB(lambda r, u: seen.append((float(r.contents.best_sse), int(r.contents.finished), threading.get_ident())))
    lib = pkg.load_library()
    pkg.binding.check(lib.goicp_set_progress_callback(eng.registration.handle, C.cast(cb, C.c_void_p), None))
    dev = C.c_int32(-1)
    pkg.binding.check(lib.goicp_device(eng.registration.handle, C.byref(dev)))
    assert dev.value == 0
    tid = []
    th = threading.Thread(target=lambda: (tid.append(threading.get_ident()), eng.run()))
    th.start(); th.join()
    pkg.binding.check(lib.goicp_set_progress_callback(eng.registration.handle, None, None))
    assert len(seen) >= 2 and seen[-1][1] == 1 and all(s[2] == tid[0] for s in seen)
    assert all(b[0] <= a[0] for a, b in zip(seen, seen[1:]))
    assert seen[-1][0] == float(eng.get_best_error())
    eng.registration.close()


def test_device_queues_match_host_queues(pkg, bunny_model, bunny_data10):
    """Inner BnB with the translation queues on the device (bnbqueue.hip: two launches per round, no host work) against
    the host-queue driver on the same searches: same bounds, same stop rule (jly_goicp.cpp:257) -> the values agree to
    within SSEThresh (the order of equal-priority expansions differs), the best node has the same bounds, and both
    reproduce the reference's own InnerBnB values (tests/golden/inner_bnb.json) to the same tolerance."""
    g = golden("inner_bnb")
    dev = pkg.Registration(bunny_model, bunny_data10, 1e-3, device_queues=1)
    host = pkg.Registration(bunny_model, bunny_data10, 1e-3, device_queues=0)
    thr = float(dev.sse_threshold)
    n = 0
    for case in g["cases"]:
        R = np.array(case["R"], np.float32)
        for full in case["full"]:
            vd, nd, cd = dev.inner_bnb(R, full["level"], full["incumbent"])
            vh, nh, ch = host.inner_bnb(R, full["level"], full["incumbent"])
            assert abs(vd - vh) <= thr and abs(vd - full["value"]) <= thr, (vd, vh, full["value"])
            assert cd.cubes == 8 * (cd.trans_pops - 1) or cd.cubes == 8 * cd.trans_pops      # every pop but the rejected one expands 8 children
            assert 0.5 * ch.cubes <= cd.cubes <= 2.0 * ch.cubes
            if nd is not None:
                ub, lb = dev.eval_bounds(R, np.array([[nd[0] + nd[3] / 2, nd[1] + nd[3] / 2, nd[2] + nd[3] / 2, nd[3]]], np.float32), full["level"])
                assert abs(ub[0] - vd) <= 1e-5 * max(vd, 1.0)
            n += 1
    assert n >= 4
    # many searches in lock-step (the shape the outer BnB produces): a whole registration both ways
    a = pkg.FastGoICP(bunny_model, bunny_data10, 1e-3, device_queues=1)
    b = pkg.FastGoICP(bunny_model, bunny_data10, 1e-3, device_queues=0)
    a.run(); b.run()
    ge = golden("e2e_bunny10")
    for e in (a, b):
        assert e.get_best_error() <= 1.02 * ge["sse"] and e.get_best_error() < ge["sse_threshold"]
        _pose_close("device/host queues", e.optR, e.optT, ge, POSE_TOL)
    assert a.counters.bounds_launches > 0 and a.counters.queue_fallbacks == 0
    # the overflow path: with room for only 48 nodes per queue every long search outgrows its slab, the batch is flagged
    # and re-run through the host queues -- same values as the host driver, and the fallback is counted
    tiny = pkg.Registration(bunny_model, bunny_data10, 1e-3, device_queues=1, queue_cap=48)
    case = g["cases"][0]
    R0 = np.array(case["R"], np.float32)
    full = case["full"][0]
    vt, nt, ct = tiny.inner_bnb(R0, full["level"], full["incumbent"])
    assert ct.queue_fallbacks >= 1 and abs(vt - full["value"]) <= thr
    c = pkg.FastGoICP(bunny_model, bunny_data10, 1e-3, device_queues=1, queue_cap=48)
    c.run()
    assert c.counters.queue_fallbacks >= 1
    assert c.get_best_error() <= 1.02 * ge["sse"] and c.get_best_error() < ge["sse_threshold"]
    for r in (dev, host, tiny, a.registration, b.registration, c.registration):
        r.close()
    # ... and on a CONVERGED search (the reference's own GoICP::Register proves this optimum: tests/golden/e2e_small6.json): with room for 512 nodes per
    # queue the long searches stop alone (QSearch::done = 2) and are re-run through the host queues while the rest of each batch stays on the device
    # -- the search as a whole still proves the same optimum with (nearly) the same node counts
    from conftest import small_problem
    tgt, src, _, _ = small_problem(6)
    g6 = golden("e2e_small6")
    e = pkg.FastGoICP(tgt, src, g6["mse_threshold"], queue_cap=512)
    e.run()
    ce = e.counters
    print("queue_cap 512 on small6: %d batches with searches re-run on the host, sse %.7g (reference %.7g), rotation nodes %d (reference %d)" % (
        ce.queue_fallbacks, e.get_best_error(), g6["sse"], ce.rot_pops, g6["rNodeCount"]))
    assert ce.queue_fallbacks >= 1 and abs(float(e.get_best_error()) - g6["sse"]) <= 1e-5 * g6["sse"]
    assert rot_angle(e.optR, np.array(g6["R"])) <= 1e-5 and abs(ce.rot_pops - g6["rNodeCount"]) <= 0.01 * g6["rNodeCount"]
    e.registration.close()


def test_queue_purges_dead_nodes_before_overflow(pkg):
    """A long upper-bound search (synthetic S1 pair at mse 1e-4: one rotation child's translation search runs ~4 700
    expansions while its incumbent keeps falling) pushes more than the 8 192 nodes its slab holds; most of them are dead
    by then -- their lower bound has come within SSEThresh of the incumbent, so the stop rule (jly_goicp.cpp:257) rejects
    them whenever they are popped.  The queue kernel throws those out instead of flagging the batch: no host fallback, and
    the registration is the host-queue driver's (which has no cap) to the same optimum."""
    from cuda_go_icp_amd import synth
    tg, sr, Rgt, tgt = synth.make_pair(**{k: synth.S1[k] for k in ("seed", "M", "N")})
    dev = pkg.FastGoICP(tg, sr, 1e-4)
    host = pkg.FastGoICP(tg, sr, 1e-4, device_queues=0)
    dev.run(); host.run()
    assert dev.counters.queue_fallbacks == 0
    assert dev.get_best_error() < dev.sse_threshold and host.get_best_error() < host.sse_threshold
    assert abs(dev.get_best_error() - host.get_best_error()) <= 1e-3 * host.get_best_error()
    assert rot_angle(dev.optR, Rgt) <= 5e-3 and np.linalg.norm(dev.optT - tgt) <= 5e-3
    dev.registration.close(); host.registration.close()


def test_queue_round_width_and_continuous_flow(pkg, bunny_model, bunny_data10):
    """The two driver options above the device queues.  (1) The round width: a search may expand up to 128 nodes per
    round (trans_batch = 128 directly; adaptive_k widens the stragglers of a batch from 32 to 64 / 128) -- more
    speculation, same bounds, same stop rule, so every width reproduces the reference's InnerBnB values
    (tests/golden/inner_bnb.json) to within SSEThresh.  (2) flow > 0 (opt-in): rotation children are handled as their
    inner searches stop and new parents are admitted while others still run -- the same prune rules in another order,
    so the registration ends at the reference's optimum as well, including through the overflow fallback."""
    g = golden("inner_bnb")
    regs = {w: pkg.Registration(bunny_model, bunny_data10, 1e-3, trans_batch=w, adaptive_k=ak) for w, ak in ((128, 0), (64, 0), (32, 1), (32, 0), (8, 1))}
    thr = float(regs[32].sse_threshold)
    for case in g["cases"]:
        R = np.array(case["R"], np.float32)
        for full in case["full"]:
            cubes = {}
            for w, reg in regs.items():
                v, node, c = reg.inner_bnb(R, full["level"], full["incumbent"])
                assert abs(v - full["value"]) <= thr, (w, v, full["value"])
                assert c.queue_fallbacks == 0
                cubes[w] = c.cubes
            assert cubes[8] <= cubes[128] <= 16 * max(cubes[8], 64)         # wider rounds speculate more, never less
    for r in regs.values():
        r.close()
    ge = golden("e2e_bunny10")
    for kw in (dict(flow=4), dict(flow=48), dict(flow=16, adaptive_k=0), dict(flow=8, queue_cap=48), dict(flow=0, adaptive_k=0)):
        e = pkg.FastGoICP(bunny_model, bunny_data10, 1e-3, **kw)
        e.run()
        assert e.get_best_error() <= 1.02 * ge["sse"] and e.get_best_error() < ge["sse_threshold"], kw
        _pose_close("flow %r" % (kw,), e.optR, e.optT, ge, POSE_TOL, sse=float(e.get_best_error()))
        assert e.counters.bounds_launches > 0
        if "queue_cap" in kw:
            assert e.counters.queue_fallbacks >= 1
        # a second run of the same engine starts from a clean flow state
        e.run()
        assert e.get_best_error() <= 1.02 * ge["sse"], kw
        e.registration.close()


def test_sharded_library_protocol_gpu(pkg, bunny_model, bunny_data10):
    """SURVEY 8(e) with the protocol inside the library (csrc/shard.cpp through goicp_register_sharded): (i) RCCL at
    world 1 -- ncclAllReduce(MIN) of the packed words and ncclBroadcast really execute on this GPU (a one-GPU box cannot
    host a second RCCL rank); (ii) 2 and 4 engines on this GPU, one host thread each, over the in-process communicator:
    the rotation cubes are dealt to the ranks, idle ranks are refilled (rebalancing), every rank ends with the global
    best, and the optimum is the single-engine one."""
    import ctypes as C
    import threading
    from cuda_go_icp_amd import sharded
    B = pkg.binding
    lib = pkg.load_library()
    g = golden("e2e_bunny10")
    # (i) RCCL, world 1
    ident = C.create_string_buffer(128)
    B.check(lib.goicp_rccl_unique_id(ident))
    comm = B.CCommOps()
    B.check(lib.goicp_rccl_comm_create(ident, 0, 1, 0, C.byref(comm)))
    B.check(lib.goicp_comm_set_timeout_ms(C.byref(comm), 5000))         # every collective below polls its stream against this deadline
    words = (C.c_uint64 * 3)(5, 1 << 40, 7)
    assert comm.allreduce_min_u64(comm.ctx, words, 3) == 0 and list(words) == [5, 1 << 40, 7]
    buf = (C.c_float * 12)(*range(12))
    assert comm.bcast(comm.ctx, C.cast(buf, C.c_void_p), 48, 0) == 0 and list(buf) == list(map(float, range(12)))
    eng = pkg.FastGoICP(bunny_model, bunny_data10, g["mse_threshold"])
    st = sharded.run_sharded_library(eng, comm, rot_pops_per_step=8)
    assert st["exchanges"] >= 1 and st["broadcasts"] >= 1 and eng.finished
    assert st["status"] == 0 and st["failed_rank"] == -1 and st["wait_ms"] >= 0 and st["step_ms"] > 0
    assert eng.get_best_error() <= 1.02 * g["sse"] and eng.get_best_error() < g["sse_threshold"]
    _pose_close("rccl world 1", eng.optR, eng.optT, g, POSE_TOL, sse=float(eng.get_best_error()))
    B.check(lib.goicp_rccl_comm_destroy(C.byref(comm)))
    eng.registration.close()
    # the single-process multi-GPU driver behind `goicp_cli --ranks N` (ncclCommInitAll + one host thread per GPU), N = 1 here
    h0, st1 = C.c_void_p(), B.CShardStats()
    p = B.CParams()
    lib.goicp_params_default(C.byref(p))
    p.mse_threshold = g["mse_threshold"]
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    B.check(lib.goicp_register_multi_gpu(C.byref(p), fp(bunny_model), len(bunny_model), fp(bunny_data10), len(bunny_data10), 1, 8, C.byref(h0), C.byref(st1)))
    res = B.CResult()
    B.check(lib.goicp_poll(h0, C.byref(res)))
    assert res.finished and res.best_sse <= 1.02 * g["sse"] and st1.exchanges >= 1
    lib.goicp_destroy(h0)
    # (ii) thread ranks on one GPU
    for world in (2, 4):
        engines = [pkg.FastGoICP(bunny_model, bunny_data10, g["mse_threshold"]) for _ in range(world)]
        comms = sharded.thread_comms(world)
        stats, errs = [None] * world, []

        def worker(r):
            try:
                stats[r] = sharded.run_sharded_library(engines[r], comms[r], rot_pops_per_step=4)
            except Exception as e:       # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        best = [float(e.get_best_error()) for e in engines]
        assert max(best) == min(best)
        assert best[0] <= 1.02 * g["sse"] and best[0] < g["sse_threshold"]
        for e in engines:
            _pose_close("thread ranks world %d" % world, e.optR, e.optT, g, POSE_TOL, sse=best[0])
        assert len({(s["exchanges"], s["broadcasts"], s["donations"]) for s in stats}) == 1
        for r in range(world):
            lib.goicp_thread_comm_destroy(comms[r])
        for e in engines:
            e.registration.close()
    # trimmed engines: the global stop test uses the ENGINE's threshold (mse * inlierNum, jly_goicp.cpp:198-208)
    one = pkg.FastGoICP(bunny_model, bunny_data10, g["mse_threshold"], trim_fraction=0.1)
    assert abs(float(one.sse_threshold) - g["mse_threshold"] * int(len(bunny_data10) * np.float32(0.9))) < 1e-4
    one.run()
    pair = [pkg.FastGoICP(bunny_model, bunny_data10, g["mse_threshold"], trim_fraction=0.1) for _ in range(2)]
    sharded.run_thread_ranks(pair, rot_pops_per_step=4)
    sse, R = float(pair[0].get_best_error()), pair[0].optR
    assert sse == float(pair[1].get_best_error())
    assert sse < pair[0].sse_threshold and abs(sse - float(one.get_best_error())) <= float(one.sse_threshold)
    assert rot_angle(R, one.optR) <= 5e-2
    for e in [one] + pair:
        e.registration.close()


def test_lds_tile_kernel_matches_direct_kernel(pkg, bunny_model, bunny_data):
    """north_star (a) names LDS-staged DT tiles; bounds_tile_kernel is that design for the expansions of one search whose
    translations lie close together (a lane group = one expansion, the DT box a 64-point patch can reach under all of them is
    copied to LDS once; a patch whose box does not fit takes the gathering kernel's sibling path).  The search lists such
    expansions for it (test_lds_tiles_on_the_search_path); this test pins the kernel itself: same per-point arithmetic as
    the direct kernel, so the bounds agree to summation order, for deep blocks (everything staged), shallow ones (nothing
    fits) and ragged segment sizes (1, 17 expansions: several lanes per expansion)."""
    import ctypes as C
    from cuda_go_icp_amd import binding as B
    reg = pkg.Registration(bunny_model, bunny_data, 1e-3)
    lib, h = reg._lib, reg.handle
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    rng = np.random.default_rng(11)
    for depth, n, chunks in ((8, 64, 4), (6, 64, 1), (3, 64, 3), (7, 17, 2), (9, 1, 1)):
        nseg = 6
        w = np.float32(1.0 / (1 << depth))
        rots = np.stack([pkg.fgoicp.rodrigues(rng.uniform(-2.0, 2.0, 3)) for _ in range(nseg)]).astype(np.float32).reshape(-1)
        par = np.zeros((nseg, n, 4), np.float32)
        for i in range(nseg):
            c0 = (np.floor(rng.uniform(-0.3, 0.3, 3) / w) * w).astype(np.float32)
            for k in range(n):
                par[i, k] = (c0[0] + (k & 3) * w, c0[1] + ((k >> 2) & 3) * w, c0[2] + ((k >> 4) & 3) * w, w)
        Bc = nseg * n * 8
        out = [np.zeros(Bc, np.float32) for _ in range(4)]
        ms = (C.c_float * 2)(); st = (C.c_uint32 * 2)()
        B.check(lib.goicp_debug_bounds_tile(h, fp(rots), fp(par.reshape(-1)), nseg, n, 5, chunks, fp(out[0]), fp(out[1]), fp(out[2]), fp(out[3]), ms, st))
        for t, d in ((out[0], out[2]), (out[1], out[3])):
            assert np.all(np.abs(t - d) <= 3e-6 * np.maximum(np.abs(d), 1e-3)), (depth, n, float(np.max(np.abs(t - d))))
        assert st[0] + st[1] > 0
        if depth >= 8:
            assert st[0] > 10 * st[1]                  # deep blocks: (nearly) every patch is staged
        if depth <= 3:
            assert st[0] == 0                          # 43-voxel sibling spacing: no box fits, the fallback carries it all
    # the direct kernel through this entry is the search's kernel: same numbers as goicp_eval_bounds
    ub, lb = reg.eval_bounds(rots[:9].reshape(3, 3), np.array([[par[0, 0, 0] + par[0, 0, 3] / 4, par[0, 0, 1] + par[0, 0, 3] / 4, par[0, 0, 2] + par[0, 0, 3] / 4, par[0, 0, 3] / 2]], np.float32), 5)
    assert abs(ub[0] - out[2][0]) <= 1e-6 * max(abs(ub[0]), 1e-3) and abs(lb[0] - out[3][0]) <= 1e-6 * max(abs(lb[0]), 1e-3)
    reg.close()


def test_lds_tiles_on_the_search_path(pkg, bunny_model, bunny_data10):
    """The tile list of the device-queue search (Params::lds_tiles): a registration that has to dig -- threshold below the
    optimum's error, so the search proves the optimum and its inner searches reach sub-voxel cubes -- run with the tile
    list off, always on, and in the default auto mode.  The tile evaluation uses the same per-point float expressions, so
    the search takes the same decisions: same optimum, node and cube counts within 0.1 %; with the list on a share of the
    cube bounds really comes from LDS tiles; a default (shallow) registration lists nothing there."""
    runs = {}
    for tiles in (0, 1, 2):
        eng = pkg.FastGoICP(bunny_model, bunny_data10, 1e-4, lds_tiles=tiles, tile_spread_vox=16.0)
        eng.run()
        c = eng.counters
        runs[tiles] = (float(eng.get_best_error()), eng.optR.copy(), int(c.cubes), int(c.rot_pops), int(c.tile_expansions))
        eng.registration.close()
    sse0, R0, cubes0, rot0, t0 = runs[0]
    assert t0 == 0
    for tiles in (1, 2):
        sse, R, cubes, rot, te = runs[tiles]
        assert abs(sse - sse0) <= 1e-5 * sse0 and rot_angle(R, R0) <= 1e-5
        assert abs(cubes - cubes0) <= 1e-3 * cubes0 and abs(rot - rot0) <= max(1, 1e-3 * rot0)
    assert runs[1][4] * 8 > 0.005 * runs[1][2]                # always on: a visible share (measured ~1 % on this run, 64-80 % on the full bunny at mse 3e-5)
    assert 0 < runs[2][4] <= runs[1][4]                       # auto: launched only after a read-back saw searches that qualify
    shallow = pkg.FastGoICP(bunny_model, bunny_data10, 1e-3)
    shallow.run()
    assert shallow.counters.tile_expansions == 0
    shallow.registration.close()


def test_twin_fusion_and_footprint_ordered_items(pkg, bunny_model, bunny_data):
    """Two launch-level mechanisms of the device-queue search, on the full bunny (BASELINE configs[1]; 460 inner searches in lock-step):
    * Params::twin_fusion -- a translation node listed in the same round by both searches of a rotation child (GoICP::InnerBnB without and
      with the rotation uncertainty, jly_goicp.cpp:494 / :532) is gathered from the distance transform once; each pass's sums see the
      operations of a separate evaluation, so the registration is BIT-identical with the fusion on and off;
    * Params::sort_items -- large rounds walk their (expansion, chunk) items in the order of the DT cell their gathers land in and cut the
      cloud into 4 096-point chunks: no term of any bound changes, only the chunking of a cube's sum (last-bit differences), so the search
      reaches the same optimum with node counts within 1 %."""
    runs = {}
    for name, kw in (("both", {}), ("no_twin", {"twin_fusion": 0}), ("no_sort", {"sort_items": 0}), ("neither", {"twin_fusion": 0, "sort_items": 0})):
        eng = pkg.FastGoICP(bunny_model, bunny_data, 1e-3, **kw)
        eng.run()
        c = eng.counters
        runs[name] = (float(eng.get_best_error()), eng.optR.copy(), eng.optT.copy(), int(c.cubes), int(c.rot_pops), int(c.trans_pops))
        eng.registration.close()
    for a, b in (("both", "no_twin"), ("no_sort", "neither")):              # twin fusion alone: same bits
        assert runs[a][0] == runs[b][0] and np.array_equal(runs[a][1], runs[b][1]) and np.array_equal(runs[a][2], runs[b][2]), (a, b)
        assert runs[a][3:] == runs[b][3:], (a, b, runs[a][3:], runs[b][3:])
    sse, R, t, cubes, rot, trans = runs["both"]
    sse0, R0, t0, cubes0, rot0, trans0 = runs["neither"]
    assert abs(sse - sse0) <= 1e-5 * sse0 and rot_angle(R, R0) <= 1e-5 and np.abs(t - t0).max() <= 1e-5
    assert abs(cubes - cubes0) <= 1e-2 * cubes0 and abs(rot - rot0) <= max(1, 1e-2 * rot0) and abs(trans - trans0) <= 1e-2 * trans0
    g = golden("e2e_bunny_full")
    assert sse <= 1.02 * g["sse"] and rot_angle(R, np.array(g["R"])) <= 2e-3


def test_compact_selection_of_proving_searches(pkg, bunny_model, bunny_data10):
    """Params::stale_compact: a proving inner search with a large queue takes its nodes in Morton order of their corners instead of by
    lower bound.  Every queued node that passes the stop rule (jly_goicp.cpp:257) is expanded whatever the order, so a registration
    that has to dig (threshold below the optimum's error) proves the same optimum either way; the counts differ by the few nodes an
    incumbent that improves later lets through (measured +1..4 %)."""
    runs = {}
    for sc in (0, 2048, 256):
        eng = pkg.FastGoICP(bunny_model, bunny_data10, 1e-4, stale_compact=sc)
        eng.run()
        c = eng.counters
        runs[sc] = (float(eng.get_best_error()), eng.optR.copy(), eng.optT.copy(), int(c.cubes), int(c.rot_pops))
        eng.registration.close()
    sse0, R0, t0, cubes0, rot0 = runs[0]
    for sc in (2048, 256):
        sse, R, t, cubes, rot = runs[sc]
        assert abs(sse - sse0) <= 1e-5 * sse0 and rot_angle(R, R0) <= 1e-5 and np.abs(t - t0).max() <= 1e-5
        assert rot == rot0 and abs(cubes - cubes0) <= 0.10 * cubes0, (sc, cubes, cubes0)
    assert runs[256][3] != cubes0                                  # the order really changed



# ----------------------------------------------------------------------------------------------
# round 4: the lean sibling path, the queue round and the tile kernel against the oracle, all sixteen outputs
# ----------------------------------------------------------------------------------------------
def _expansion_parents(rng, n_per_depth, depths):
    """Translation nodes (corner xyz + width) of the given depths inside the root cube [-0.5, 0.5]^3, as the BnB makes them:
    corner = -0.5 + k * w with integer k (jly_goicp.cpp:262-273).  No node twice (a queue never holds one twice, and the twin test of
    the evaluation matches nodes by value)."""
    out = []
    for d in depths:
        w = np.float32(1.0) / np.float32(1 << d)
        seen = set()
        for kk in rng.integers(0, 1 << d, (4 * n_per_depth, 3)):
            if tuple(kk) in seen or len(seen) >= min(n_per_depth, 8 ** d):
                continue
            seen.add(tuple(kk))
            out.append([np.float32(-0.5) + np.float32(kk[0]) * w, np.float32(-0.5) + np.float32(kk[1]) * w, np.float32(-0.5) + np.float32(kk[2]) * w, w])
    return np.array(out, np.float32)


def _children(parent):
    px, py, pz, pw = map(np.float32, parent)
    w = pw / np.float32(2)
    kids = []
    for j in range(8):
        cx = px + np.float32(j & 1) * w; cy = py + np.float32(j >> 1 & 1) * w; cz = pz + np.float32(j >> 2 & 1) * w
        kids.append([cx + w / np.float32(2), cy + w / np.float32(2), cz + w / np.float32(2), w])
    return np.array(kids, np.float32)


def test_sibling_expansion_all_sixteen_vs_oracle(pkg, oracle_mod, oracle_dt_bunny, bunny_model, bunny_data):
    """The path every registration and the headline bench run -- the LEAN sibling path (lean_points, device.hip; reference body
    GoICP::InnerBnB, src/goicp/jly_goicp.cpp:262-335) -- on the FULL bunny (N = 30 379, V = 300: a grid that fits the Infinity Cache
    selects it): for 43 parents of depths 0..6 (in-grid) plus 6 whose children straddle or leave the DT grid, under two rotations
    and levels {-1, 5}, ALL sixteen outputs (8 ub and 8 lb of the children) against oracle.cube_bound, rel 1e-4 -- through
      (i)   goicp_eval_bounds (bounds_kernel<1, true>, the microbench's kernel),
      (ii)  one round of the device-resident queues with both passes listed (bnb_queue_kernel + bounds_queue_kernel<1, true> with
            twin fusion: the lower-bound search's items evaluate both passes, lean_points<.., 2>),
      (iii) the same round with twin fusion off (each search's own items),
      (iv)  bounds_tile_kernel (LDS-staged DT boxes; its own lean path for boxes inside the grid, the gathering fall-back otherwise)."""
    import ctypes as C
    B = pkg.binding
    rng = np.random.default_rng(11)
    parents = _expansion_parents(rng, 7, range(0, 7))                   # 1 + 7 x 6 nodes
    # nodes outside the root cube whose children straddle the grid's faces or lie beyond them (the grid is the target's bounding cube
    # expanded twice: about [-1.75, 1.75]^3; the cloud reaches |p| = 1.4)
    far = np.array([[0.75, -0.25, 0.0, 0.5], [-1.5, 0.5, 0.25, 0.5], [0.25, 1.0, -1.25, 0.25], [2.0, 2.0, -2.0, 1.0], [-0.5, -0.5, 0.875, 0.125],
                    [1.0, -1.0, 1.0, 0.0625]], np.float32)
    parents = np.concatenate([parents, far])
    assert len(parents) >= 32
    _, rho = oracle_mod.rot_radii(bunny_data)
    regs = {tw: pkg.Registration(bunny_model, bunny_data, 1e-3, twin_fusion=tw) for tw in (1, 0)}
    reg = regs[1]
    lib, h = reg._lib, reg.handle
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    n = len(parents)
    worst = {}
    for v in ([0.3, -0.2, 0.9], [-2.1, 0.4, 1.1]):
        R = pkg.fgoicp.rodrigues(v)
        prot = oracle_mod.rotate(R, bunny_data)
        level = 5
        want = {}                                               # (pass, parent, child) -> (ub, lb) of the oracle
        for ps, r in ((0, None), (1, rho[level])):
            for e, par in enumerate(parents):
                for c, kid in enumerate(_children(par)):
                    want[(ps, e, c)] = oracle_mod.cube_bound(oracle_dt_bunny, prot, r, kid[:3], kid[3])

        def check(tag, ps, ub, lb):
            for e in range(n):
                for c in range(8):
                    ou, ol = want[(ps, e, c)]
                    du, dl = abs(ub[8 * e + c] - ou) / max(ou, 1e-3), abs(lb[8 * e + c] - ol) / max(ol, 1e-3)
                    worst[tag] = max(worst.get(tag, 0.0), du, dl)
                    assert du <= 1e-4 and dl <= 1e-4, (tag, ps, e, c, ub[8 * e + c], ou, lb[8 * e + c], ol)
                    assert lb[8 * e + c] <= ub[8 * e + c]
        # (i) the operator API, eight siblings per group
        kids = np.concatenate([_children(p) for p in parents])
        for ps, lv in ((0, -1), (1, level)):
            ub, lb = reg.eval_bounds(R, kids, lv)
            check("eval_bounds", ps, ub, lb)
        # (ii) / (iii) one round of the device queues, twin fusion on / off
        for tw, r_ in regs.items():
            out = [np.zeros(8 * n, np.float32) for _ in range(4)]
            info = (C.c_int32 * 2)()
            par = np.ascontiguousarray(parents.reshape(-1))
            B.check(lib.goicp_debug_queue_expand(r_.handle, fp(np.ascontiguousarray(R.reshape(-1).astype(np.float32))), level, fp(par), n,
                                                 fp(out[0]), fp(out[1]), fp(out[2]), fp(out[3]), info))
            assert info[1] == tw
            check("queue twin=%d chunks=%d" % (tw, info[0]), 0, out[0], out[1])
            check("queue twin=%d chunks=%d" % (tw, info[0]), 1, out[2], out[3])
        # (iv) the tile kernel: one segment of n <= 64 expansions, at the lower-bound pass's level and without radii
        for ps, lv in ((0, -1), (1, level)):
            for s0 in range(0, n, 32):
                seg = np.ascontiguousarray(parents[s0:s0 + 32])
                m = len(seg)
                out = [np.zeros(8 * m, np.float32) for _ in range(4)]
                ms, st = (C.c_float * 2)(), (C.c_uint32 * 2)()
                B.check(lib.goicp_debug_bounds_tile(h, fp(np.ascontiguousarray(R.reshape(-1).astype(np.float32))), fp(seg.reshape(-1)), 1, m, lv, 4,
                                                    fp(out[0]), fp(out[1]), fp(out[2]), fp(out[3]), ms, st))
                for e in range(m):
                    for c in range(8):
                        ou, ol = want[(ps, s0 + e, c)]
                        for tag, ub, lb in (("tile", out[0], out[1]), ("direct(parents)", out[2], out[3])):
                            du, dl = abs(ub[8 * e + c] - ou) / max(ou, 1e-3), abs(lb[8 * e + c] - ol) / max(ol, 1e-3)
                            worst[tag] = max(worst.get(tag, 0.0), du, dl)
                            assert du <= 1e-4 and dl <= 1e-4, (tag, ps, s0 + e, c)
    print("sixteen outputs vs oracle, worst relative deviation:", {k: "%.2e" % v for k, v in worst.items()})
    for r_ in regs.values():
        r_.close()


def test_search_range_bound_on_a_split_plane(pkg):
    """The cull of a configured search range (in_box, engine.hpp / bnbqueue.hip; the [params.translation] keys of the reference's
    configs, src/common.h:157-169) takes a cube as the half-open box [x, x + w)^3 against the CLOSED range: the range's high face
    is inclusive.  A z range of a quarter of the root's width puts BOTH its bounds on depth-3 split planes (root [c - w/2, c + w/2],
    range [c - w/8, c + w/8]): of the four depth-3 layers around it the layer that starts at hi is kept, the layer that ends at lo is
    not -- three layers of 64 cubes, not two (the strict form), not four.  A problem whose lower bounds are all 0 down to depth 3 and
    whose upper bounds are never 0 makes the search expand every node in range down to the depth limit, so the pops COUNT the cull:
    1 + 8 + 32 + 192 expansions at depth limit 4.  (Target: a lattice of spacing 0.02 -- no point of space is farther than 0.0173 from
    it, below the 0.054 a depth-3 cube subtracts, so every lower bound is 0; five source points at incommensurate offsets -- no
    cube centre puts all five into seeded voxels, so the incumbent stays above the lower bounds and above SSEThresh.)"""
    ax = np.arange(-0.5, 0.8001, 0.02)
    target = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    source = np.array([[0.0, 0.0, 0.0], [0.3037, 0.0011, 0.0023], [0.0041, 0.2513, 0.0079], [0.1517, 0.1009, 0.2031], [0.2203, 0.0307, 0.1129]], np.float32)
    w = 0.5
    lo, hi = [-w / 2, -w / 2, -w / 8], [w / 2, w / 2, w / 8]
    for dq, tb in ((1, 32), (0, 32), (0, 1)):
        reg = pkg.Registration(target, source, 1e-9, device_queues=dq, trans_batch=tb, wide_children=1 if tb > 1 else 0,
                               use_trans_range=1, trans_min=lo, trans_max=hi, trans_search_depth=4)
        v, best, cnt = reg.inner_bnb(np.eye(3, dtype=np.float32), -1, 1e10)
        assert v > 5 * 1e-9                                                       # the incumbent stayed above the lower bounds (0) and above SSEThresh
        assert cnt.trans_pops == 1 + 8 + 32 + 192, (dq, tb, cnt.trans_pops)
        reg.close()


def test_flow_with_trimming(pkg, bunny_model, bunny_data10):
    """Continuous flow + trimming: the queue kernel widens a stale search's step up to x4, and the trimmed evaluation runs
    one workgroup per listed expansion -- its grid must cover what the round can list (it was sized for q_hi x K: expansions listed
    beyond that kept stale bounds).  Against the oracle's trimmed registration of the same clouds (tests/golden/e2e_bunny10_trim_oracle.json)
    and the host-queue run: the flow run must reach the oracle's optimum (measured: SSE 0.35515 against 0.35528, pose 2e-4 rad away); the
    widened host-queue run ends in a neighbouring optimum 1.7 % higher (0.045 rad away) -- held to the SSE bar only."""
    o = golden("e2e_bunny10_trim_oracle")
    kw = dict(trim_fraction=0.1)
    a = pkg.FastGoICP(bunny_model, bunny_data10, 1e-3, flow=8, **kw)
    b = pkg.FastGoICP(bunny_model, bunny_data10, 1e-3, device_queues=0, **kw)
    a.run(); b.run()
    assert a.finished and b.finished
    for e in (a, b):
        assert e.get_best_error() < e.sse_threshold and e.get_best_error() <= 1.02 * o["sse"]
    print("flow + trim: sse %.6g, host queues %.6g, oracle %.6g; flow vs host queues rot %.3e" % (a.get_best_error(), b.get_best_error(), o["sse"], rot_angle(a.optR, b.optR)))
    _pose_close("flow + trim vs the oracle's trimmed optimum", a.optR, a.optT, o, POSE_TOL, sse=float(a.get_best_error()))
    a.registration.close(); b.registration.close()


def test_tile_list_with_few_search_slots(pkg, bunny_model, bunny_data10):
    """The tile list's segment buffer: with rot_batch = 1 the queues are sized for 16 search slots, and a few deep searches
    widened to 100+ expansions each need more <= 64-expansion segments than 2 x slots.  The buffer holds list_cap / 64 + slots
    segments now and the kernel checks it; a prove-the-optimum search with the tile list always on must agree with tiles off."""
    res = {}
    for tiles in (1, 0):
        e = pkg.FastGoICP(bunny_model, bunny_data10, 1e-4, rot_batch=1, lds_tiles=tiles)
        e.run()
        assert e.finished
        res[tiles] = (float(e.get_best_error()), e.counters.cubes, e.counters.tile_expansions, e.counters.queue_fallbacks, np.array(e.optR), np.array(e.optT))
        e.registration.close()
    assert res[1][2] > 0 and res[0][2] == 0                              # the tile list really ran
    assert abs(res[1][0] - res[0][0]) <= 1e-4 * res[0][0]
    assert abs(res[1][1] - res[0][1]) <= 0.02 * res[0][1]                  # same search (last-bit differences of the sums only)
    assert rot_angle(res[1][4], res[0][4]) <= 1e-4



def test_sharding_work_inflation_is_bounded(pkg, bunny_model, bunny_data):
    """What sharding costs in extra work, so that it cannot regress silently (VERDICT r3 #1c; tools/shard_inflation.py has the full table,
    DESIGN 5): every rank runs its own best-first order over its share of the rotation cubes, so ranks expand cubes -- and, above all, run
    inner searches against incumbents -- that a single global order would have handled more cheaply.  Full bunny at mse 1e-4 (SSEThresh below
    the optimum's error: the search has to prove the optimum; 8.8 M cube bounds, 994 rotation nodes, 0.28 s at world 1), 8 ranks as host threads
    on this GPU over the library's protocol.  Measured (round 4): cube bounds of all ranks / world 1 = 1.37 at 8 parents per step, 1.66 with
    the 8 -> 32 ramp; rotation nodes 1.01; the busiest rank evaluates 0.177 / 0.169 of the total (1/8 = 0.125).  The bars are those + 15 %.
    (The 6.7-second prove-the-optimum run at mse 3e-5 inflates by 1.01 / 1.10 at 8 ranks: the deep searches dominate there.)"""
    from cuda_go_icp_amd import sharded
    e1 = pkg.FastGoICP(bunny_model, bunny_data, 1e-4)
    e1.run()
    c1, r1, sse1 = e1.counters.cubes, e1.counters.rot_pops, float(e1.get_best_error())
    e1.registration.close()
    for ramp, bar in ((0, 1.6), (32, 1.9)):
        engines = [pkg.FastGoICP(bunny_model, bunny_data, 1e-4) for _ in range(8)]
        stats = sharded.run_thread_ranks(engines, rot_pops_per_step=8, ramp_to=ramp)
        cubes = [e.counters.cubes for e in engines]
        rots = sum(e.counters.rot_pops for e in engines)
        infl, share = sum(cubes) / c1, max(cubes) / sum(cubes)
        print("sharding, 8 thread ranks, ramp_to %d: work inflation %.3f, node inflation %.3f, busiest rank's share %.3f, %d steps" % (ramp, infl, rots / r1, share, stats[0]["steps"]))
        assert all(s["status"] == 0 for s in stats)
        assert infl <= bar and rots / r1 <= 1.1 and share <= 0.25
        for e in engines:
            assert abs(float(e.get_best_error()) - sse1) <= 1e-3 * sse1          # the same proven optimum on every rank
            e.registration.close()

def test_bounds_fp16_optin(pkg, bunny_model, bunny_data10):
    """Params::bounds_fp16 (opt-in, not the parity path): the BnB bounds read a half-precision copy of the bricked DT
    rounded toward zero.  Against the fp32 engine on the same cubes: every lower bound is <= the fp32 one (still a valid
    lower bound), upper bounds are low by at most 2 x 2^-10 relative (squares), the DT re-score of a pose stays
    bit-identical (fp32 grid), and a registration reaches the same optimum."""
    a = pkg.Registration(bunny_model, bunny_data10, 1e-3)
    b = pkg.Registration(bunny_model, bunny_data10, 1e-3, bounds_fp16=1)
    rng = np.random.default_rng(5)
    cubes = _cubes(rng, 256)
    R = pkg.fgoicp.rodrigues([0.3, -0.2, 0.9])
    for level in (-1, 4):
        ua, la = a.eval_bounds(R, cubes, level)
        ub, lb = b.eval_bounds(R, cubes, level)
        assert np.all(ub <= ua * (1 + 1e-6)) and np.all(ub >= ua * (1 - 1e-2))      # (v - rho) amplifies the 2^-10 of a half
        assert np.all(lb <= la * (1 + 1e-6) + 1e-7)
        assert not np.array_equal(ua, ub)                      # the half grid really is in use
    t = np.array([0.05, -0.02, 0.01], np.float32)
    assert a.compute_sse_error(R, t) == b.compute_sse_error(R, t)
    a.close(); b.close()
    g = golden("e2e_bunny10")
    e = pkg.FastGoICP(bunny_model, bunny_data10, g["mse_threshold"], bounds_fp16=1)
    e.run()
    assert e.get_best_error() <= 1.02 * g["sse"] and e.get_best_error() < g["sse_threshold"]
    _pose_close("bounds_fp16", e.optR, e.optT, g, POSE_TOL)
    e.registration.close()


def test_bunny_icp_config0(pkg, oracle_mod):
    """BASELINE configs[0], test/bunny_icp.toml:10-20 on the reference's own scans: target bun045.ply (40 097 points), source
    bun000.ply (40 256), resize 15, plain ICP (modes 0-2 are the same arithmetic, src/icp_kernel.cu:48-279; the reference
    iterates forever, its viewer shows the running pose).  The step API (ICP::kdTreeGPUStep -> goicp_icp_step) against the
    oracle's ICP iteration with fresh means, step by step: 1e-4 abs on R, t over 15 steps; the error decreases monotonically."""
    target, source = cloud("bun045"), cloud("bun000")
    assert len(target) == 40097 and len(source) == 40256
    reg = pkg.Registration(target, source, 1e-5)
    kd = oracle_mod.KdTree(target)
    R, t = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    errs = []
    for _ in range(15):
        snap = reg.icp_step()
        e, R, t, _ = kd.icp_run(source, R, t, 1, -1e30)
        errs.append(float(snap.best_sse))
        assert abs(snap.best_sse - e) <= 2e-3 * e          # 40 256 float terms (sequential vs tree sum) on two trajectories that drift apart by ~1e-5 per step
    assert np.abs(np.array(snap.curR, np.float32).reshape(3, 3) - R).max() <= 1e-4
    assert np.abs(np.array(snap.curT, np.float32) - t).max() <= 1e-4
    assert all(b <= a * (1 + 1e-5) for a, b in zip(errs, errs[1:]))
    # exact NN on this target through the same hierarchy
    rng = np.random.default_rng(3)
    q = source[rng.choice(len(source), 1000, replace=False)]
    idx, d2 = reg.nn_query(q)
    bi, bd = oracle_mod.nn_brute(target, q)
    assert np.array_equal(d2, bd) and np.array_equal(idx, bi)
    reg.close()


def test_icp_neighbour_cache_is_exact(pkg, oracle_mod, bunny_model, bunny_data):
    """The ICP pass skips the tree walk of a query when its cached neighbour is PROVABLY still the nearest
    (|q - m| + |q - q_ref| < the 2-nearest bound found at q_ref): not an approximation, so an ICP run with the cache is
    bit-identical to one without it -- pose, error and iteration count -- from two start poses, full cloud; and the
    correspondences behind a pass at the converged pose are the brute-force nearest neighbours."""
    a = pkg.Registration(bunny_model, bunny_data, 1e-3, icp_nn_cache=1)
    b = pkg.Registration(bunny_model, bunny_data, 1e-3, icp_nn_cache=0)
    for R0, t0 in ((np.eye(3), np.zeros(3)), (pkg.fgoicp.rodrigues([0.2, -0.1, 0.15]), np.array([0.05, 0.02, -0.03]))):
        for iters in (1, 7, 60, 10000):
            ea, Ra, ta = pkg.IterativeClosestPoint3D(a, iters, 1e-7, R0, t0).run()
            eb, Rb, tb = pkg.IterativeClosestPoint3D(b, iters, 1e-7, R0, t0).run()
            assert ea == eb and np.array_equal(Ra, Rb) and np.array_equal(ta, tb), (iters, ea, eb)
    # the error of a pass = sum of exact NN distances at that pose (cache warm from the runs above)
    icp = pkg.IterativeClosestPoint3D(a, 10000, 1e-7, np.eye(3), np.zeros(3))
    e, R, t = icp.run()
    moved = (bunny_data @ R.T + t).astype(np.float32)
    e1, _, _ = pkg.IterativeClosestPoint3D(a, 1, 1e-7, R, t).run()
    q = np.ascontiguousarray(a.transform_source(R, t))
    _, bd = oracle_mod.nn_brute(bunny_model, q[::7])
    idx, d2 = a.nn_query(q[::7])
    assert np.array_equal(d2, bd)
    assert abs(e1 - d2.astype(np.float64).sum() * 7) <= 0.05 * e1          # sanity: same scale (every 7th point)
    import ctypes as C
    hits = C.c_int64()
    fpc = lambda x: np.ascontiguousarray(x, np.float32).ctypes.data_as(C.POINTER(C.c_float))
    Rf, tf = np.ascontiguousarray(R, np.float32).reshape(9), np.ascontiguousarray(t, np.float32)
    pkg.binding.check(pkg.load_library().goicp_debug_cache_hits(a.handle, Rf.ctypes.data_as(C.POINTER(C.c_float)), tf.ctypes.data_as(C.POINTER(C.c_float)), C.byref(hits)))
    assert hits.value >= 0.99 * len(bunny_data)                             # a repeated pose: (almost) every query skips its walk
    a.close(); b.close()
