#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- fixtures that are outputs of OUR oracle (not of the reference), cached because they take
a minute of CPU: the trimmed end-to-end registration of bunny / 10 (trim_fraction 0.1), which the GPU test
test_trimmed_e2e_vs_oracle compares with.  (The reference hard-wires trimFraction 0, so there is no reference
output for this case; the oracle's trimmed bounds are pinned to the reference's trimmed InnerBnB by
tests/golden/inner_bnb_trim.json.)  usage: python oracle/gen_oracle_fixtures.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import oracle as O  # noqa: E402


def trimmed_bunny10():
    g = os.path.join(ROOT, "tests", "golden")
    model = np.fromfile(os.path.join(g, "model_bunny.f32"), dtype="<f4").reshape(-1, 3)
    data = np.ascontiguousarray(np.fromfile(os.path.join(g, "data_bunny.f32"), dtype="<f4").reshape(-1, 3)[::10])
    dt = O.DistanceTransform(model, 300, 2.0)
    o = O.register(dt, model, data, 1e-3, trim_fraction=0.1)
    return {"what": "oracle.register(bunny model, every 10th data point, mse 1e-3, trim_fraction 0.1)",
            "R": np.asarray(o["R"], np.float64).reshape(-1).tolist(), "t": np.asarray(o["t"], np.float64).reshape(-1).tolist(),
            "sse": float(o["sse"])}


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "e2e_bunny10_trim_oracle.json")
    with open(out, "w") as f:
        json.dump(trimmed_bunny10(), f, indent=1)
    print("wrote", out)
