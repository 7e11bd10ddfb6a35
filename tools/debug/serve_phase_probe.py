"""GPU box: where a one-image serve() spends its wall time - upload + preprocess / resident step / collect - against serve() itself."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from common import make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
for variant in ("full", "head"):
    over = dict(image_size="1280x768", mc_dropout=True, mc_dropoutsamp=10, loss_attenuation=True)
    over.update(dict(mc_dropoutrate=0.05) if variant == "full" else dict(mc_classheadrate=0.05, mc_boxheadrate=0.05))
    p = make_params(**over)
    w = make_weights(p, seed=0)
    img = make_images(1, 768, 1280, seed=2)
    d = KerasDriver("_", False, p["name"], 1, False, p, weights=w, chunk_images=1)
    for _ in range(5):
        d.serve(img)
    def med(f, n=40):
        ts = []
        for _ in range(n):
            t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
        return np.median(ts) * 1e3
    t_serve = med(lambda: d.serve(img))
    def up(): d.stage_images(img); d.synchronize()
    t_up = med(up)
    t_run = med(lambda: d.run_resident(sync=True))
    d.run_resident(sync=True)
    t_col = med(lambda: d._collect(1))
    def both(): d.run_resident(sync=True); d._collect(1)
    t_both = med(both)
    print("%s: serve %.2f ms = stage_images+sync %.2f | run_resident(sync) %.2f | collect %.2f (run+collect %.2f)" % (variant, t_serve, t_up, t_run, t_col, t_both))
    d.close()
