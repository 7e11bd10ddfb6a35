import sys, time, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from common import make_params, make_weights, FULL_MC
from uda_amd.infer_lib import KerasDriver, ServingDriver
p = make_params(**FULL_MC); w = make_weights(p)
d = KerasDriver("_", False, p["name"], batch_size=1, model_params=p, weights=w)
rng = np.random.default_rng(0)
n, K = 112, 5000
c = rng.uniform(0, 500, (n, K, 2)); wh = rng.uniform(5, 80, (n, K, 2))
boxes = np.concatenate([c - wh / 2, c + wh / 2], -1).astype(np.float32)
scores = rng.uniform(0, 1, (n, K)).astype(np.float32)
for _ in range(2):
    t = time.perf_counter(); out = d.nms(boxes, scores); dt = time.perf_counter() - t
print("nms %d problems x %d candidates: %.2f ms (incl. transfers)" % (n, K, dt * 1e3), out[2][:4])
