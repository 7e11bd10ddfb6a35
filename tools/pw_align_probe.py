#!/usr/bin/env python
"""Does the 64-byte misalignment of every other pixel row cost the 1x1 projections anything?  144 (576-byte rows) and 240 (960)
input channels against their aligned neighbours, same map, float32-class products (terms 16): GB/s of the algorithmic bytes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_ops import _run  # noqa: E402

rng = np.random.default_rng(0)
for hw, cout, cins, rows in ((192 * 320, 24, (128, 144, 160), 48), (96 * 160, 40, (224, 240, 256), 96)):
    for cin in cins:
        x = rng.normal(0, 1, (rows, hw, cin)).astype(np.float32)
        w = (rng.normal(0, 1, (cin, cout)) / np.sqrt(cin)).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        sh = rng.normal(0, 0.3, cout).astype(np.float32)
        se = rng.uniform(0.1, 1.0, (rows, cin)).astype(np.float32)
        res = rng.normal(0, 1, (rows, hw, cout)).astype(np.float32)
        by = rows * hw * (cin + 2 * cout) * 4
        _, ms = _run(x, w, None, sc, sh, se, None, res, 1, 0, 16, 7)
        print("%4d -> %-3d @%-6d rows %-3d  %7.3f ms  %6.0f GB/s  (%d-byte input rows)" % (cin, cout, hw, rows, ms, by / ms / 1e6, cin * 4), flush=True)
