"""GPU box: heads of sample t from a T = 3 handle vs from a T = 1 handle told it runs global sample t (no process group needed)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from common import BOX_ONLY_MC, FULL_MC, make_images, make_params, make_weights
from uda_amd.infer_lib import ServingDriver
for name, over in (("box_only", BOX_ONLY_MC), ("full", FULL_MC)):
    p = make_params(**dict(over, mc_dropoutsamp=3))
    w = make_weights(p, seed=33, cls_spread=20.0)
    imgs = make_images(2, 100, 180, seed=34)
    d = ServingDriver(p["name"], 2, True, p, weights=w)
    d.set_dropout_seed(11)
    d.run_network(imgs)
    cls3, box3 = d.head_outputs(2)
    m3 = d.dropout_masks(2)
    d.close()
    for t in range(3):
        p1 = dict(p, mc_dropoutsamp=1, uda_force_sample_axis=True)
        d1 = ServingDriver(p["name"], 2, True, p1, weights=w)
        d1.set_sample_shard(t, 3, 3)
        d1.set_dropout_seed(11)
        d1.run_network(imgs)
        cls1, box1 = d1.head_outputs(2)
        m1 = d1.dropout_masks(2)
        bad_m = [k for k in m1 if not np.array_equal(m1[k][:, 0], m3[k][:, t])]
        for l in range(5):
            b3 = box3[l][t] if box3[l].ndim == 5 else box3[l]
            c3 = cls3[l][t] if cls3[l].ndim == 5 else cls3[l]
            db, dc = np.abs(box1[l] - b3).max(), np.abs(cls1[l] - c3).max()
            if db or dc or bad_m:
                print(name, "sample", t, "level", l, "box diff", db, "cls diff", dc, "mask sites differing", bad_m[:4])
        ops1 = [(o["kind"], o["drop_site"], o["drop_site2"]) for o in d1.plan.ops if o["kind"] == 8 and not o["fuse_in"]]
        d1.close()
    print(name, "done; T=1 head ops (kind, site, in_site):", ops1[:8])
