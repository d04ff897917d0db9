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


# ------------------------------------------------------------------------------------------------------------------
# The protocol inside the library (csrc/shard.cpp, goicp_run_sharded / goicp_register_sharded): one all-reduce(MIN) of
# five packed 64-bit words per step, the winner's pose broadcast when the global best moved, rebalancing of idle ranks.
# The helpers below build the two callback tables from Python objects: a communicator over torch.distributed (gloo in
# the CPU tests; the RCCL communicator of the library needs no Python) and an engine table over any object with the
# stepped interface (the tests' CPU stand-in) -- the product engine is driven by goicp_register_sharded directly.
# ------------------------------------------------------------------------------------------------------------------
def torch_comm_ops(dist, device):
    """goicp_comm_ops over an initialised torch.distributed group.  Keep the returned object alive while it is used."""
    import ctypes as C
    import torch
    from . import binding as B
    SIGN = -(1 << 63)

    def allreduce(ctx, words, n):
        try:
            vals = [int(words[i]) for i in range(n)]
            t = torch.tensor([(v ^ (1 << 63)) - (1 << 64) if (v ^ (1 << 63)) >= (1 << 63) else (v ^ (1 << 63)) for v in vals],
                             dtype=torch.int64, device=device)            # unsigned order -> signed order
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            for i, v in enumerate(t.tolist()):
                words[i] = (v & ((1 << 64) - 1)) ^ (1 << 63)
            return 0
        except Exception:          # noqa: BLE001 -- must not unwind into C
            return -6

    def bcast(ctx, buf, nbytes, root):
        try:
            raw = (C.c_ubyte * nbytes).from_address(buf)
            t = torch.tensor(list(raw), dtype=torch.uint8, device=device)
            dist.broadcast(t, src=root)
            C.memmove(buf, bytes(t.cpu().tolist()), nbytes)
            return 0
        except Exception:          # noqa: BLE001
            return -6

    ops = B.CCommOps()
    ops.ctx, ops.rank, ops.world = None, dist.get_rank(), dist.get_world_size()
    ops._keep = (B.ALLREDUCE_FN(allreduce), B.BCAST_FN(bcast))
    ops.allreduce_min_u64, ops.bcast = ops._keep
    return ops


def engine_ops(engine):
    """goicp_shard_engine_ops over a Python object with the stepped interface (set_shard, register_begin, register_step,
    pose, offer_best, queue_size, donate, receive, register_end, sse_threshold)."""
    import ctypes as C
    from . import binding as B

    def guard(fn):
        def wrapped(*a):
            try:
                fn(*a)
                return 0
            except Exception:      # noqa: BLE001
                import traceback
                traceback.print_exc()
                return -6
        return wrapped

    def begin(ctx, rank, world):
        engine.set_shard(rank, world)
        engine.register_begin()

    def step(ctx, max_pops, out):
        s = engine.register_step(max_pops)
        o = out.contents
        o.finished, o.early_exit, o.best_sse = int(s["finished"]), int(s["early_exit"]), float(s["best_sse"])
        o.frontier_lb, o.rot_pops = float(s["frontier_lb"]), int(s["rot_pops"])

    def pose(ctx, sse, R, t):
        e, Rv, tv = engine.pose()
        sse[0] = float(e)
        for i, v in enumerate(np.asarray(Rv, np.float32).reshape(9)):
            R[i] = float(v)
        for i, v in enumerate(np.asarray(tv, np.float32).reshape(3)):
            t[i] = float(v)

    def offer(ctx, sse, R, t):
        engine.offer_best(sse, np.array([R[i] for i in range(9)], np.float32), np.array([t[i] for i in range(3)], np.float32))

    def qsize(ctx, n):
        n[0] = int(engine.queue_size())

    def donate(ctx, max_nodes, nodes, n):
        give = engine.donate(int(max_nodes))
        for k, node in enumerate(give):
            for j in range(7):
                nodes[7 * k + j] = float(node[j])
        n[0] = len(give)

    def receive(ctx, nodes, n):
        engine.receive([tuple(nodes[7 * k + j] for j in range(7)) for k in range(n)])

    def end(ctx):
        engine.register_end()

    ops = B.CShardEngineOps()
    ops.ctx, ops.sse_threshold = None, float(engine.sse_threshold)
    ops._keep = (B.ENG_BEGIN_FN(guard(begin)), B.ENG_STEP_FN(guard(step)), B.ENG_POSE_FN(guard(pose)), B.ENG_OFFER_FN(guard(offer)),
                 B.ENG_QSIZE_FN(guard(qsize)), B.ENG_DONATE_FN(guard(donate)), B.ENG_RECEIVE_FN(guard(receive)), B.ENG_END_FN(guard(end)))
    (ops.begin, ops.step, ops.pose, ops.offer, ops.queue_size, ops.donate, ops.receive, ops.end) = ops._keep
    return ops


def run_sharded_library(engine_or_ops, comm_ops, rot_pops_per_step=8, rebalance=True):
    """Drive one rank through the library's protocol.  engine_or_ops: a fgoicp.FastGoICP (real engine) or a
    CShardEngineOps table.  Returns the goicp_shard_stats as a dict."""
    import ctypes as C
    from . import binding as B
    lib = B.load_library()
    st = B.CShardStats()
    if isinstance(engine_or_ops, B.CShardEngineOps):
        B.check(lib.goicp_run_sharded(C.byref(engine_or_ops), C.byref(comm_ops), int(rot_pops_per_step), int(bool(rebalance)), C.byref(st)))
    else:
        B.check(lib.goicp_register_sharded(engine_or_ops.registration.handle, C.byref(comm_ops), int(rot_pops_per_step), int(bool(rebalance)), C.byref(st)))
    return {k: getattr(st, k) for k, _ in B.CShardStats._fields_}


def thread_comms(world):
    """`world` in-process communicators (goicp_thread_comm_create): rank r is driven by its own host thread."""
    import ctypes as C
    from . import binding as B
    arr = (B.CCommOps * world)()
    B.check(B.load_library().goicp_thread_comm_create(world, arr))
    return arr
