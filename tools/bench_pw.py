#!/usr/bin/env python
"""Time the pointwise kernels on the 1x1-conv shapes of the bench workload (D0, 1280x768, 160 sample rows per
chunk by default): f32-input MFMA vs split-bf16 x3 / x6.   python tools/bench_pw.py [--rows 160] [--cfg N]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_ops import _run  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=160)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--terms", default="0,3,6")
ap.add_argument("--only", default="")
a = ap.parse_args()

SHAPES = [  # hw, cin, cout, flags   (per sample row)
    (384 * 640, 32, 16, "bn"), (192 * 320, 96, 24, "bn,se"), (192 * 320, 144, 24, "bn,se,res"),
    (96 * 160, 144, 40, "bn,se"), (96 * 160, 240, 40, "bn,se,res"), (48 * 80, 240, 80, "bn,se"),
    (48 * 80, 80, 480, "bn,act,mask"), (48 * 80, 480, 80, "bn,se,res"), (48 * 80, 480, 112, "bn,se"),
    (48 * 80, 112, 672, "bn,act,mask"), (48 * 80, 672, 112, "bn,se,res"), (24 * 40, 672, 192, "bn,se"),
    (24 * 40, 192, 1152, "bn,act,mask"), (24 * 40, 1152, 192, "bn,se,res"), (24 * 40, 1152, 320, "bn,se"),
    (96 * 160, 64, 64, "bias,bn"), (96 * 160, 64, 63, "bias"), (96 * 160, 64, 72, "bias"), (96 * 160, 40, 64, "bn"),
]
rng = np.random.default_rng(0)
print("%-28s %s" % ("shape", "  ".join("terms=%s: ms  GB/s  TF/s" % t for t in a.terms.split(","))))
for hw, cin, cout, flags in SHAPES:
    name = "%d->%d @%d" % (cin, cout, hw)
    if a.only and a.only not in name:
        continue
    rows = a.rows
    while rows * hw * max(cin, cout) * 4 > 6e9:
        rows //= 2
    f = set(flags.split(","))
    x = rng.normal(0, 1, (rows, hw, cin)).astype(np.float32)
    w = (rng.normal(0, 1, (cin, cout)) / np.sqrt(cin)).astype(np.float32)
    bias = rng.normal(0, 0.5, cout).astype(np.float32) if "bias" in f else None
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32) if "bn" in f else None
    sh = rng.normal(0, 0.3, cout).astype(np.float32) if "bn" in f else None
    se = rng.uniform(0.1, 1.0, (rows, cin)).astype(np.float32) if "se" in f else None
    mask = ((rng.uniform(0, 1, (rows, cout)) >= 0.05) / 0.95).astype(np.float32) if "mask" in f else None
    res = rng.normal(0, 1, (rows, hw, cout)).astype(np.float32) if "res" in f else None
    by = rows * hw * (cin + cout + (cout if res is not None else 0)) * 4
    fl = 2.0 * rows * hw * cin * cout
    cols = []
    ref = None
    for t in [int(v) for v in a.terms.split(",")]:
        out, ms = _run(x, w, bias, sc, sh, se, mask, res, 1, int("act" in f), t, a.reps)
        if ref is None:
            ref = out
        err = float(np.abs(out - ref).max() / np.abs(ref).max())
        cols.append("%7.3f %6.0f %6.1f (d %.1e)" % (ms, by / ms / 1e6, fl / ms / 1e9, err))
    print("%-28s rows=%-4d %s" % (name, rows, "  ".join(cols)), flush=True)
