"""The multi-rank path (cuda-go-icp_amd/sharded.py) on CPU: rotation cubes dealt to ranks, best
error min-all-reduced, winner's pose broadcast, global termination -- with torch.distributed's gloo
backend at world_size 2 and with in-process lock-step ranks.  Bounds come from the oracle through
tests/fake_engine.py; the protocol code under test is the product's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, cloud, load_pkg, rot_angle

MSE = 5e-3    # keeps the search non-trivial (no early exit after the first ICP) yet short on a 64^3 DT


def _reference_single(fe_mod):
    from cuda_go_icp_amd import sharded
    e = fe_mod.FakeEngine(cloud("model_rand"), cloud("data_rand"), MSE)
    return sharded.run_local_ranks([e], rot_pops_per_step=2)


@pytest.fixture(scope="module")
def fe_mod(oracle_mod):
    load_pkg()
    import fake_engine
    return fake_engine


@pytest.fixture(scope="module")
def single(fe_mod):
    return _reference_single(fe_mod)


@pytest.mark.parametrize("world", [2, 4])
def test_lockstep_ranks_reach_single_rank_optimum(fe_mod, single, world):
    from cuda_go_icp_amd import sharded
    engines = [fe_mod.FakeEngine(cloud("model_rand"), cloud("data_rand"), MSE) for _ in range(world)]
    sse, R, t, stats = sharded.run_local_ranks(engines, rot_pops_per_step=2)
    s0, R0, t0, _ = single
    thr = float(engines[0].sse_threshold)
    assert abs(sse - s0) <= thr                       # both are within SSEThresh of the global optimum
    if sse < thr or s0 < thr or abs(sse - s0) < 1e-3 * s0:
        assert rot_angle(R, R0) < 0.05 and np.linalg.norm(t - t0) < 0.05
    # every rank ends with the global best (pruning signal propagated)
    assert all(abs(e.pose()[0] - sse) < 1e-6 for e in engines)


WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import torch, torch.distributed as dist
from conftest import cloud, load_pkg
load_pkg()
from cuda_go_icp_amd import sharded
import fake_engine
dist.init_process_group(backend="gloo")
e = fake_engine.FakeEngine(cloud("model_rand"), cloud("data_rand"), {mse})
ex = sharded.TorchExchange(dist, torch.device("cpu"))
sse, R, t, stats = sharded.run_sharded(e, ex, rot_pops_per_step=2)
json.dump({{"sse": float(sse), "R": R.reshape(-1).tolist(), "t": t.tolist(), "stats": stats, "rank": dist.get_rank()}},
          open(os.path.join({out!r}, "rank%d.json" % dist.get_rank()), "w"))
dist.barrier()
dist.destroy_process_group()
"""


def test_gloo_world2(fe_mod, single, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, mse=MSE, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(script)]
    subprocess.run(cmd, check=True, env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    res = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    s0, R0, t0, _ = single
    thr = 100 * MSE
    assert res[0]["sse"] == res[1]["sse"]                                   # identical global best on both ranks
    assert np.allclose(res[0]["R"], res[1]["R"]) and np.allclose(res[0]["t"], res[1]["t"])   # winner's pose broadcast
    assert res[0]["stats"]["exchanges"] == res[1]["stats"]["exchanges"]     # collectives matched
    assert abs(res[0]["sse"] - s0) <= thr


# ----------------------------------------------------------------------------------------------------------------
# the protocol inside the library (csrc/shard.cpp) through the C ABI: goicp_run_sharded with callback tables
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world,rebalance", [(2, True), (4, True), (3, False)])
def test_library_protocol_thread_ranks(fe_mod, single, world, rebalance):
    """`world` host threads, one CPU stand-in engine each, the library's in-process communicator
    (goicp_thread_comm_create): the packed all-reduce, the pose broadcast, termination and rebalancing of
    csrc/shard.cpp reach the single-rank optimum; collectives match on every rank."""
    import threading
    from cuda_go_icp_amd import sharded
    engines = [fe_mod.FakeEngine(cloud("model_rand"), cloud("data_rand"), MSE) for _ in range(world)]
    tables = [sharded.engine_ops(e) for e in engines]
    comms = sharded.thread_comms(world)
    stats, errs = [None] * world, []

    def worker(r):
        try:
            stats[r] = sharded.run_sharded_library(tables[r], comms[r], rot_pops_per_step=2, rebalance=rebalance)
        except Exception as e:       # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    assert not errs and all(s is not None for s in stats), errs
    lib = load_pkg().load_library()
    for r in range(world):
        lib.goicp_thread_comm_destroy(comms[r])
    s0, R0, t0, _ = single
    thr = float(engines[0].sse_threshold)
    best = [e.pose()[0] for e in engines]
    assert max(best) - min(best) < 1e-6                               # every rank ends with the global best
    assert abs(best[0] - s0) <= thr
    assert len({(s["exchanges"], s["broadcasts"], s["donations"]) for s in stats}) == 1      # collectives matched
    if not rebalance:
        assert stats[0]["donations"] == 0


WORKER_LIB = r"""
import json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import torch, torch.distributed as dist
from conftest import cloud, load_pkg
load_pkg().load_library()
from cuda_go_icp_amd import sharded
import fake_engine
dist.init_process_group(backend="gloo")
e = fake_engine.FakeEngine(cloud("model_rand"), cloud("data_rand"), {mse})
comm = sharded.torch_comm_ops(dist, torch.device("cpu"))
stats = sharded.run_sharded_library(sharded.engine_ops(e), comm, rot_pops_per_step=2, rebalance=True)
sse, R, t = e.pose()
json.dump({{"sse": float(sse), "R": R.reshape(-1).tolist(), "t": t.tolist(), "stats": stats, "rank": dist.get_rank()}},
          open(os.path.join({out!r}, "lib_rank%d.json" % dist.get_rank()), "w"))
dist.barrier()
dist.destroy_process_group()
"""


def test_library_protocol_gloo_world2(fe_mod, single, tmp_path):
    """The same library code with torch.distributed's gloo as the communicator, two processes (the shape of the
    one-process-per-GPU launch; RCCL replaces gloo there through goicp_rccl_comm_create)."""
    script = tmp_path / "worker_lib.py"
    script.write_text(WORKER_LIB.format(root=ROOT, mse=MSE, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", str(script)]
    r = subprocess.run(cmd, env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    res = [json.load(open(tmp_path / ("lib_rank%d.json" % k))) for k in range(2)]
    s0, R0, t0, _ = single
    assert res[0]["sse"] == res[1]["sse"]
    assert np.allclose(res[0]["R"], res[1]["R"]) and np.allclose(res[0]["t"], res[1]["t"])
    assert res[0]["stats"]["exchanges"] == res[1]["stats"]["exchanges"] and res[0]["stats"]["broadcasts"] == res[1]["stats"]["broadcasts"]
    assert abs(res[0]["sse"] - s0) <= 100 * MSE
