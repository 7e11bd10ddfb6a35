"""GPU box: score distribution of one full-size image for several class-predict spreads (which spread gives untied, unsaturated tops?)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from common import make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
FULL = dict(image_size="1280x768", mc_dropout=True, mc_dropoutrate=0.05, mc_dropoutsamp=2, loss_attenuation=True)
p = make_params(**FULL)
imgs = make_images(1, 768, 1280, seed=7)
for spread in (1.0, 5.0, 20.0, 60.0):
    w = make_weights(p, seed=0, cls_spread=spread)
    d = KerasDriver("_", False, p["name"], 1, False, p, weights=w)
    d.set_dropout_seed(9)
    det = d.serve(imgs)
    c = d.candidates(1)
    s = np.sort(c["scores"][0])[::-1]
    print("spread", spread, "top", s[:6].tolist(), "q50", float(np.median(s)), "unique among top 1000:", len(np.unique(s[:1000])), "det scores", det[1][0][:5].tolist(), "valid", det[3].tolist())
    d.close()
