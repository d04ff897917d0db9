#!/usr/bin/env python3
"""Why is the sharded driver at world 1 slower than goicp_register on the deep bunny run (7.6 vs 6.7 s, tools/shard_inflation.py)?
Same engine, same problem, four drivers: register() on the main thread / on a second Python thread / the stepped API (begin, step(64)...,
end) on the main thread / the library's sharded protocol over a 1-rank thread communicator."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _pkg
pkg = _pkg(); pkg.load_library()
from cuda_go_icp_amd import sharded
g = os.path.join(ROOT, "tests", "golden")
ld = lambda n: np.fromfile(os.path.join(g, n + ".f32"), dtype="<f4").reshape(-1, 3)
bm, bd = ld("model_bunny"), ld("data_bunny")
mse = float(sys.argv[1]) if len(sys.argv) > 1 else 3e-5
def timed(fn):
    t0 = time.perf_counter(); fn(); return time.perf_counter() - t0
def cnt(e):
    c = e.counters
    return "rot %d trans %d icp %d launches %d fallbacks %d tile_exp %d" % (c.rot_pops, c.trans_pops, c.icp_iters, c.bounds_launches, c.queue_fallbacks, c.tile_expansions)
for rep in range(1):
    e = pkg.FastGoICP(bm, bd, mse)
    print("register, main thread      %.3f s  cubes %d" % (timed(e.run), e.counters.cubes)); e.registration.close()
    e = pkg.FastGoICP(bm, bd, mse)
    def in_thread():
        t = threading.Thread(target=e.run); t.start(); t.join()
    print("register, second thread    %.3f s  cubes %d" % (timed(in_thread), e.counters.cubes)); e.registration.close()
    e = pkg.FastGoICP(bm, bd, mse)
    def stepped():
        e.register_begin()
        while True:
            s = e.register_step(64)
            if s["finished"]:
                break
        e.register_end()
    print("stepped API (64), main     %.3f s  cubes %d" % (timed(stepped), e.counters.cubes), cnt(e)); e.registration.close()
    for w in (8, 32):
        e = pkg.FastGoICP(bm, bd, mse)
        def stepped_w():
            e.register_begin()
            while True:
                s = e.register_step(w)
                if s["finished"]:
                    break
            e.register_end()
        print("stepped API (%d), main     %.3f s  cubes %d" % (w, timed(stepped_w), e.counters.cubes), cnt(e)); e.registration.close()
    e = pkg.FastGoICP(bm, bd, mse)
    comm = sharded.thread_comms(1)
    box = {}
    print("sharded world 1 ramp 64, main thread %.3f s  cubes %d" % (timed(lambda: box.update(sharded.run_sharded_library(e, comm[0], 8, ramp_to=64))), e.counters.cubes), cnt(e), box); e.registration.close()
    e = pkg.FastGoICP(bm, bd, mse)
    print("sharded world 1 ramp 64, thread rank %.3f s  cubes %d" % (timed(lambda: sharded.run_thread_ranks([e], 8, ramp_to=64)), e.counters.cubes)); e.registration.close()
