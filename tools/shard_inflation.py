#!/usr/bin/env python3
"""What sharding the rotation-cube search costs in extra work -- the hardware-independent ceiling of the multi-GPU scaling curve.

Every rank of the sharded search (csrc/shard.cpp) runs its own best-first order over its share of the rotation cubes, so
ranks expand cubes a single global order would have pruned.  This tool runs 1 / 2 / 4 / 8 ranks as host threads over the library's
in-process communicator on ONE GPU (N engines, the library's own protocol -- the same code path as over RCCL) and counts:

  work inflation   = cube bounds of all ranks / cube bounds of the world-1 search            (1.0 = no wasted work)
  node inflation   = rotation nodes of all ranks / world-1's
  critical share   = cube bounds of the busiest rank / cube bounds of all ranks             (1/world = perfectly balanced)
  speedup ceiling  = world-1 cube bounds / busiest rank's cube bounds   (what N GPUs could reach if a cube bound cost the same everywhere
                     and exchanges were free -- wall times here are NOT scaling figures: the thread ranks share one GPU)

for the step rules / protocol options a deployment can choose: a fixed 8 rotation parents per step, the single-GPU driver's ramp
(8 -> 64), rebalancing off, the one-step-stale exchange.

usage (GPU box): python tools/shard_inflation.py [--worlds 1,2,4,8] [--quick] > gpurun_out/shard_inflation.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--quick", action="store_true", help="skip the 6.7-second prove-the-optimum workload")
    ap.add_argument("--md", default="", help="also write the table as markdown to this file")
    ap.add_argument("--only", default="", help="semicolon-separated variant names to run (default: all)")
    ap.add_argument("--workloads", default="", help="comma-separated indices of the workloads to run (default: all)")
    args = ap.parse_args()
    from __graft_entry__ import _pkg
    pkg = _pkg()
    pkg.load_library()
    from cuda_go_icp_amd import sharded
    g = os.path.join(ROOT, "tests", "golden")
    ld = lambda n: np.fromfile(os.path.join(g, n + ".f32"), dtype="<f4").reshape(-1, 3)
    bm, bd = ld("model_bunny"), ld("data_bunny")
    workloads = [("bunny mse 1e-4 (N=30379; proves the optimum: 0.3 s, 8.7 M cube bounds at world 1)", bm, bd, 1e-4)]
    if not args.quick:
        workloads.append(("bunny mse 3e-5 (proves the optimum: 6.7 s, 340 M cube bounds at world 1)", bm, bd, 3e-5))
    workloads.append(("spanner 150k x 150k mse 2e-5 (the reference's noisy scans; 0.2 s at world 1)", ld("spanner_target"), ld("spanner_source"), 2e-5))
    workloads.append(("bunny mse 1e-3 (BASELINE configs[1]: early exit after 384 rotation nodes, 35 ms at world 1)", bm, bd, 1e-3))
    variants = [("fixed 8 parents/step", dict(rot_pops_per_step=8, ramp_to=0)),
                ("ramp 8->64", dict(rot_pops_per_step=8, ramp_to=64)),
                ("ramp 8->64, no rebalancing", dict(rot_pops_per_step=8, ramp_to=64, rebalance=False)),
                ("ramp 8->64, stale exchange", dict(rot_pops_per_step=8, ramp_to=64, stale=True)),
                ("ramp 8->32", dict(rot_pops_per_step=8, ramp_to=32)),
                ("ramp 8->16", dict(rot_pops_per_step=8, ramp_to=16))]
    if args.only:
        variants = [v for v in variants if v[0] in args.only.split(";")]
    worlds = [int(w) for w in args.worlds.split(",")]
    if args.workloads:
        workloads = [workloads[int(i)] for i in args.workloads.split(",")]
    out = {"tool": "tools/shard_inflation.py", "ranks_are": "host threads over goicp_thread_comm_create on ONE GPU (protocol = csrc/shard.cpp, as over RCCL)", "workloads": []}
    rows_md = []
    for name, tgt, src, mse in workloads:
        e1 = pkg.FastGoICP(tgt, src, mse)
        t0 = time.perf_counter()
        e1.run()
        w1 = time.perf_counter() - t0
        c1 = e1.counters
        base = {"wall_s": round(w1, 4), "cube_bounds": int(c1.cubes), "rot_pops": int(c1.rot_pops), "icp_iters": int(c1.icp_iters), "sse": float(e1.get_best_error())}
        e1.registration.close()
        wl = {"workload": name, "world1_goicp_register": base, "runs": []}
        print("# %s: world 1 %.3f s, %d cube bounds, %d rotation nodes" % (name, w1, c1.cubes, c1.rot_pops), file=sys.stderr)
        for vname, kw in variants:
            for world in worlds:
                engines = [pkg.FastGoICP(tgt, src, mse) for _ in range(world)]
                t0 = time.perf_counter()
                stats = sharded.run_thread_ranks(engines, **kw)
                wall = time.perf_counter() - t0
                cubes = [int(e.counters.cubes) for e in engines]
                rots = [int(e.counters.rot_pops) for e in engines]
                sse = [float(e.get_best_error()) for e in engines]
                r = {"variant": vname, "world": world, "wall_s_one_gpu": round(wall, 4), "cube_bounds_all": sum(cubes), "cube_bounds_busiest": max(cubes),
                     "rot_pops_all": sum(rots), "work_inflation": round(sum(cubes) / max(base["cube_bounds"], 1), 4),
                     "node_inflation": round(sum(rots) / max(base["rot_pops"], 1), 4), "critical_share": round(max(cubes) / max(sum(cubes), 1), 4),
                     "speedup_ceiling": round(base["cube_bounds"] / max(max(cubes), 1), 3), "steps": int(stats[0]["steps"]), "exchanges": int(stats[0]["exchanges"]),
                     "donations": int(stats[0]["donations"]), "idle_steps_all": int(sum(s["steps_idle"] for s in stats)), "sse": sse[0], "same_sse_all_ranks": bool(max(sse) == min(sse))}
                wl["runs"].append(r)
                rows_md.append("| %s | %s | %d | %.3f | %.3f | %.3f | %.2f | %d | %d |" % (name.split(" (")[0], vname, world, r["work_inflation"], r["node_inflation"], r["critical_share"],
                                                                                       r["speedup_ceiling"], r["steps"], r["donations"]))
                print("  %-32s world %d: inflation %.3f (nodes %.3f), busiest rank %.3f of all, ceiling %.2fx, %d steps, %.3f s" %
                      (vname, world, r["work_inflation"], r["node_inflation"], r["critical_share"], r["speedup_ceiling"], r["steps"], wall), file=sys.stderr)
                for e in engines:
                    e.registration.close()
        out["workloads"].append(wl)
    print(json.dumps(out, indent=1))
    if args.md:
        with open(args.md, "w") as f:
            f.write("| workload | step rule | ranks | work inflation | node inflation | busiest rank's share | speedup ceiling | steps | donations |\n|---|---|---|---|---|---|---|---|---|\n")
            f.write("\n".join(rows_md) + "\n")


if __name__ == "__main__":
    main()
