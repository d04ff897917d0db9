"""Multi-GPU Go-ICP: the rotation-cube queue sharded over ranks, one process per GPU.

New relative to the reference (single CUDA device 0, src/window.cpp:110); SURVEY.md 8(e).
Every rank holds a full replica of both clouds, the distance transform and the k-d tree, owns
every world-th cube of the 64 level-2 rotation cubes (goicp_set_shard) and runs its own
best-first search with goicp_register_step().  Between steps the ranks exchange, over
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests):

    all_reduce(MIN) of [best_sse, frontier_lb, -early_exit, -active]      16 bytes
    all_reduce(MIN) of [owner rank of the global best, -someone_is_behind]  8 bytes
    broadcast of the winner's R|t (12 floats), only when some rank is behind

so every rank prunes against the global best-so-far error.  Payloads are tiny: the exchange is
latency-bound, the data path itself needs no collective.  Termination: any rank's early exit
(best < SSEThresh, jly_goicp.cpp:527), every queue empty/converged, or
global_best - min frontier lb <= SSEThresh (jly_goicp.cpp:416 applied to the union of the queues).

The driver only needs the stepped engine interface (set_shard, register_begin, register_step,
pose, offer_best, register_end, sse_threshold), which is what fgoicp.FastGoICP exposes.
"""
import math

import numpy as np


class TorchExchange:
    """Collectives over an initialised torch.distributed process group."""

    def __init__(self, dist, device):
        import torch
        self.torch, self.dist, self.device = torch, dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.n_exchanges = 0

    def min4(self, vals):
        t = self.torch.tensor(vals, dtype=self.torch.float32, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return t.tolist()

    def min2(self, vals):
        t = self.torch.tensor(vals, dtype=self.torch.float32, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return t.tolist()

    def bcast_pose(self, pose12, src):
        t = self.torch.tensor(pose12, dtype=self.torch.float32, device=self.device)
        self.dist.broadcast(t, src=src)
        return t.cpu().numpy()


def exchange_once(ex, engine, status):
    """One exchange round for this rank.  Returns (stop, global_best)."""
    sse, R, t = engine.pose()
    inf = math.inf
    lb = status["frontier_lb"] if not status["finished"] else inf
    gbest, glb, neg_early, neg_active = ex.min4([sse, lb, -float(status["early_exit"]), -float(not status["finished"])])
    behind = sse > gbest
    owner, neg_behind = ex.min2([float(ex.rank) if sse <= gbest else float(ex.world), -float(behind)])
    if neg_behind < 0:
        pose = ex.bcast_pose(np.concatenate([R.reshape(9), t.reshape(3)]).tolist(), int(owner))
        if behind:
            engine.offer_best(gbest, pose[:9], pose[9:])
    ex.n_exchanges += 1
    stop = (neg_early < 0) or (neg_active == 0) or (gbest - glb <= float(engine.sse_threshold))
    return stop, gbest


def run_sharded(engine, ex, rot_pops_per_step=8, max_steps=1 << 30):
    """Drive one rank to global termination.  Returns (best_sse, R, t, stats)."""
    engine.set_shard(ex.rank, ex.world)
    engine.register_begin()
    status = {"finished": False, "early_exit": False, "frontier_lb": 0.0}
    steps = 0
    while steps < max_steps:
        status = engine.register_step(rot_pops_per_step)
        steps += 1
        stop, _ = exchange_once(ex, engine, status)
        if stop:
            break
    engine.register_end()
    sse, R, t = engine.pose()
    return sse, R.reshape(3, 3), t, {"steps": steps, "exchanges": ex.n_exchanges, "rot_pops": status.get("rot_pops", 0)}


def run_local_ranks(engines, rot_pops_per_step=8, max_steps=1 << 30):
    """Emulate `len(engines)` ranks in one process (lock-step)."""
    world = len(engines)
    for r, e in enumerate(engines):
        e.set_shard(r, world)
        e.register_begin()
    steps = 0
    gbest = math.inf
    while steps < max_steps:
        sts = [e.register_step(rot_pops_per_step) for e in engines]
        steps += 1
        poses = [e.pose() for e in engines]
        gbest = min(p[0] for p in poses)
        glb = min((s["frontier_lb"] if not s["finished"] else math.inf) for s in sts)
        owner = min(r for r in range(world) if poses[r][0] <= gbest)
        for r, e in enumerate(engines):
            if poses[r][0] > gbest:
                e.offer_best(gbest, poses[owner][1], poses[owner][2])
        any_early = any(s["early_exit"] for s in sts)
        any_active = any(not s["finished"] for s in sts)
        if any_early or not any_active or gbest - glb <= float(engines[0].sse_threshold):
            break
    for e in engines:
        e.register_end()
    sse, R, t = engines[owner].pose()
    return sse, np.asarray(R).reshape(3, 3), np.asarray(t), {"steps": steps, "rot_pops": [s["rot_pops"] for s in sts]}
