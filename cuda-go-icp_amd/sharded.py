"""Multi-GPU Go-ICP: Python plumbing around the sharded search of the library (csrc/shard.cpp, csrc/rccl_comm.cpp).

New relative to the reference (single CUDA device 0, src/window.cpp:110); SURVEY.md 8(e).  Every rank holds a full replica
of both clouds, the distance transform and the k-d tree, owns every world-th cube of the 64 level-2 rotation cubes and
runs its own best-first search; the exchange / termination / rebalancing / failure protocol lives in the library
(goicp_run_sharded_opt / goicp_register_sharded_opt) -- there is no second implementation of it here.  This module only
builds the two callback tables the library takes from Python objects:

  * a communicator over torch.distributed (gloo in the CPU tests; on GPUs the library's own RCCL communicator needs no
    Python at all), or the library's in-process thread communicator for N ranks in one process;
  * an engine table over any object with the stepped interface (the tests' CPU stand-in) -- a real engine is driven by
    goicp_register_sharded_opt directly.
"""
import numpy as np


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


def run_sharded_library(engine_or_ops, comm_ops, rot_pops_per_step=8, rebalance=True, stale=False, raise_on_error=True, ramp_to=0):
    """Drive one rank through the library's protocol.  engine_or_ops: a fgoicp.FastGoICP (real engine) or a
    CShardEngineOps table.  Returns the goicp_shard_stats as a dict (plus "status": the call's return code when
    raise_on_error is False)."""
    import ctypes as C
    from . import binding as B
    lib = B.load_library()
    st = B.CShardStats()
    opt = B.CShardOptions(int(rot_pops_per_step), int(bool(rebalance)), int(bool(stale)), int(ramp_to))
    if isinstance(engine_or_ops, B.CShardEngineOps):
        rc = lib.goicp_run_sharded_opt(C.byref(engine_or_ops), C.byref(comm_ops), C.byref(opt), C.byref(st))
    else:
        rc = lib.goicp_register_sharded_opt(engine_or_ops.registration.handle, C.byref(comm_ops), C.byref(opt), C.byref(st))
    if raise_on_error:
        B.check(rc)
    out = {k: getattr(st, k) for k, _ in B.CShardStats._fields_}
    out["status"] = rc
    return out


def run_thread_ranks(engines, rot_pops_per_step=8, rebalance=True, stale=False, timeout_ms=None, raise_on_error=True, ramp_to=0):
    """len(engines) ranks in ONE process: one host thread per rank over the library's in-process communicator
    (goicp_thread_comm_create) -- N engines on one GPU, or N CPU stand-ins.  engines: fgoicp.FastGoICP objects or
    CShardEngineOps tables.  Returns the per-rank stats (run_sharded_library's dicts)."""
    import threading
    from . import binding as B
    world = len(engines)
    comms = thread_comms(world)
    lib = B.load_library()
    if timeout_ms is not None:
        for r in range(world):
            B.check(lib.goicp_comm_set_timeout_ms(comms[r], int(timeout_ms)))
    stats, errs = [None] * world, []

    def worker(r):
        try:
            stats[r] = run_sharded_library(engines[r], comms[r], rot_pops_per_step, rebalance, stale, raise_on_error, ramp_to)
        except Exception as e:       # noqa: BLE001 -- reported to the caller below
            errs.append((r, e))

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for r in range(world):
        lib.goicp_thread_comm_destroy(comms[r])
    if errs:
        raise errs[0][1]
    return stats


def thread_comms(world):
    """`world` in-process communicators (goicp_thread_comm_create): rank r is driven by its own host thread."""
    import ctypes as C
    from . import binding as B
    arr = (B.CCommOps * world)()
    B.check(B.load_library().goicp_thread_comm_create(world, arr))
    return arr
