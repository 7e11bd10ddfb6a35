import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from common import FULL_MC, make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
for name, kw, hw in (("d0", dict(image_size="192x128", **FULL_MC), (128, 192)), ("d2", dict(model="efficientdet-d2", image_size="128x128", **FULL_MC), (128, 128))):
    p = make_params(**kw)
    w = make_weights(p, seed=31)
    imgs = make_images(2, hw[0] - 28, hw[1] - 12, seed=32)
    d = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
    try:
        d.serve(imgs); print(name, "ok")
    except Exception as e:
        print(name, "FAILED", e)
    d.close()
