"""GPU box: randomised stress of the NMS kernels against the oracle's heap (bit for bit) - sizes from a few thousand to the whole
anchor set, 1-6 problems per call (different blocks-per-problem shapes of the cooperative kernel), tied and spread scores, clustered
boxes, soft and hard rules.  One process, one handle; prints the number of problems checked."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from common import make_params, make_weights
from uda_amd.infer_lib import KerasDriver
from oracle import post_ref as P
p = make_params()
d = KerasDriver("_", False, p["name"], batch_size=1, model_params=p, weights=make_weights(p))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
t0 = time.time(); checked = 0; calls = 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
while time.time() - t0 < budget:
    n = int(rng.choice([3000, 9000, 20000, 50000, 120000, 184140]))
    n_img = int(rng.integers(1, 7)) if n < 100000 else int(rng.integers(1, 4))
    mode = rng.choice(["tied", "spread", "clustered", "plateau"])
    sigma = float(rng.choice([0.25, 0.25, 0.5, 0.15, 0.0]))
    thr = 0.001 if sigma > 0 else float(rng.choice([0.3, float("-inf")]))
    boxes = np.zeros((n_img, n, 4), np.float32); scores = np.zeros((n_img, n), np.float32)
    for i in range(n_img):
        span = 400.0 if mode != "spread" else 4000.0
        c = rng.uniform(0, span, (n, 2))
        if mode == "clustered":
            centres = rng.uniform(0, span, (20, 2)); c = centres[rng.integers(0, 20, n)] + rng.normal(0, 6, (n, 2))
        wh = rng.uniform(4, 120, (n, 2))
        boxes[i] = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        if mode == "tied":
            s = (0.01 + rng.normal(0, 1e-4, n)).astype(np.float32); s[rng.integers(0, n, n // 8)] = s[0]
        elif mode == "plateau":
            s = rng.choice(np.float32([0.0101, 0.0102, 0.0103, 0.0099]), n).astype(np.float32)
        else:
            s = rng.uniform(0, 1, n).astype(np.float32)
        scores[i] = s
    idx, sc, valid = d.nms(boxes, scores, 100, 0.5, thr, sigma)
    calls += 1
    for i in range(n_img):
        ridx, rsc, rvalid = P.nms_v5(boxes[i], scores[i], 100, 0.5, thr, sigma, True)
        assert valid[i] == rvalid, (n, n_img, mode, sigma, thr, i)
        assert (idx[i] == ridx).all() and (sc[i] == rsc).all(), (n, n_img, mode, sigma, thr, i)
        checked += 1
print("nms stress ok: %d problems in %d calls, %.0f s; fallbacks coop %d prefix %d" % (checked, calls, time.time() - t0, d.nms_coop_fallbacks(), d.nms_prefix_fallbacks()))
d.close()
