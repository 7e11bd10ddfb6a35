"""GPU box: which serve deviates after a range demotion (first = replayed, second / third = ordinary runs)?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from common import LOSS_ATT, make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
p = make_params(**LOSS_ATT)
w = dict(make_weights(p, seed=81))
k = [n for n in w if n.endswith("blocks_3/tpu_batch_normalization_1/gamma")]
q = [n for n in w if n.endswith("blocks_3/conv2d_1/kernel")]
w[k[0]] = w[k[0]] * np.float32(3.0e5)
w[q[0]] = w[q[0]] / np.float32(3.0e5)
a = make_images(2, 100, 180, seed=82)
runs = []
d = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
for i in range(3):
    det = d.serve(a)
    cls, box = d.head_outputs(2)
    runs.append((det, cls, box))
    print("serve", i, "demotions", d.range_demotions(), "finite", all(np.isfinite(x).all() for x in list(det) + cls + box))
d.close()
for i in (1, 2):
    for name, x, y in [("det%d" % j, runs[0][0][j], runs[i][0][j]) for j in range(len(runs[0][0]))] + \
                      [("cls%d" % j, runs[0][1][j], runs[i][1][j]) for j in range(5)] + [("box%d" % j, runs[0][2][j], runs[i][2][j]) for j in range(5)]:
        if not np.array_equal(x, y):
            dd = np.abs(x.astype(np.float64) - y.astype(np.float64))
            print("serve 0 vs %d: %s differs: max %g at %s, %d of %d elements, nan %d/%d" % (i, name, np.nanmax(dd), np.unravel_index(np.nanargmax(dd), dd.shape), (dd > 0).sum(), dd.size, np.isnan(x).sum(), np.isnan(y).sum()))
print("serve 1 vs 2 equal:", all(np.array_equal(x, y) for x, y in zip(runs[1][0], runs[2][0])))
