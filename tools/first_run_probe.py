#!/usr/bin/env python3
"""Where does the first registration of a fresh engine lose time against the second (43 vs 34 ms on the bunny)?  One engine, verbose, run twice."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import cloud, load_pkg
pkg = load_pkg()
model, data = cloud("model_bunny"), cloud("data_bunny")
eng = pkg.FastGoICP(model, data, 1e-3, verbose=2)
for r in range(3):
    t0 = time.perf_counter(); eng.run(); print("run %d: %.2f ms" % (r, 1e3 * (time.perf_counter() - t0)), file=sys.stderr, flush=True)
