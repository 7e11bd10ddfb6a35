"""Head-output error of the current switches against the float32 CPU oracle (D0 at 192x128 and D2 at 128x128, full MC) - the
measurement behind tests/test_gpu_round3.py::test_six_term..., as a tool: prints one line per run."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
from common import FULL_MC, make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
from oracle import effdet_ref as E, philox_ref as R, preprocess_ref as PP
out = {}
for name, kw, hw in (("d0", dict(image_size="192x128", **FULL_MC), (128, 192)),
                     ("d2", dict(model="efficientdet-d2", image_size="128x128", **FULL_MC), (128, 128))):
    p = make_params(**kw)
    w = make_weights(p, seed=31)
    imgs = make_images(2, hw[0] - 28, hw[1] - 12, seed=32)
    d = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
    d.set_dropout_seed(7)
    d.serve(imgs)
    cls, box = d.head_outputs(2)
    x, _ = PP.preprocess(imgs, hw, p["mean_rgb"], p["stddev_rgb"])
    rcls, rbox = E.forward(w, p, x, R.make_masks(E.dropout_sites(p), 7, 2, 3))
    wm = wr = 0.0
    for l in range(5):
        for g, r, groups in ((cls[l], rcls[l], [(0, cls[l].shape[-1])]), (box[l], rbox[l], [(0, 36), (36, 72)])):
            for lo, hi in groups:
                gg, rr = g[..., lo:hi].astype(np.float64), r[..., lo:hi].astype(np.float64)
                wm = max(wm, np.abs(gg - rr).max() / np.abs(rr).max())
                wr = max(wr, np.sqrt(np.mean((gg - rr) ** 2)) / np.sqrt(np.mean(rr * rr)))
    out[name] = "max %.2e rms %.2e" % (wm, wr)
    d.close()
print(os.environ.get("UDA_PW_SCHEME", "default"), os.environ.get("UDA_F16_KINDS", "all"), out, flush=True)
