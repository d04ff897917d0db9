"""Dev script (not a test): the engine's own time split of a full-bunny registration (verbose=1, stderr)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import cloud, load_pkg  # noqa: E402

pkg = load_pkg()
model, data = cloud("model_bunny"), cloud("data_bunny")
for v in (0, 1, 1):
    pkg.FastGoICP(model, data, 1e-3, verbose=v).run()
