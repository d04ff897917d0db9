"""Dev script (not a test): engine creation time split (verbose=1, stderr), bunny then synthetic 1 M."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import cloud, load_pkg  # noqa: E402

pkg = load_pkg()
model, data = cloud("model_bunny"), cloud("data_bunny")
for k in range(3):
    t0 = time.perf_counter()
    reg = pkg.Registration(model, data, 1e-3, verbose=1)
    print("bunny create %.1f ms" % (1e3 * (time.perf_counter() - t0)), file=sys.stderr)
    reg.close()
if len(sys.argv) > 1:
    from cuda_go_icp_amd import synth
    tgt, src, _, _ = synth.make_pair(seed=synth.S2["seed"], M=synth.S2["M"], N=synth.S2["N"])
    for k in range(2):
        t0 = time.perf_counter()
        reg = pkg.Registration(tgt, src, 1e-3, dt_size=512, verbose=1)
        print("S2 create %.1f ms" % (1e3 * (time.perf_counter() - t0)), file=sys.stderr)
        reg.close()
