"""Diagnostic (round 4): full-size batch, network only (no post-process), head outputs checked for non-finite values and
hashed - separates a fault of the conv stack from one of the post-process, and shows whether two builds / switches agree."""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
p = make_params(image_size="1280x768", mc_dropout=True, mc_dropoutrate=0.05, mc_dropoutsamp=10, loss_attenuation=True)
w = make_weights(p, seed=0)
d = KerasDriver("_", False, p["name"], n, False, p, weights=w, chunk_images=n)
d.set_dropout_seed(5)
imgs = make_images(n, 768, 1280, seed=2)
d._feed(imgs)
for rep in range(3):
    d._ck(d._lib.uda_run(d._h, -1, 0), "uda_run")
    d._ck(d._lib.uda_synchronize(d._h), "uda_synchronize")
    d._last_n = n
    cls, box = d.head_outputs(n)
    bad = sum(int((~np.isfinite(a)).sum()) for a in cls + box)
    h = hashlib.sha1(b"".join(np.ascontiguousarray(a[..., :]).tobytes() for a in (cls[2], box[2], cls[4], box[4]))).hexdigest()[:12]
    print("rep %d: non-finite %d, max |cls| %.4g, max |box| %.4g, hash(levels 5,7) %s" % (
        rep, bad, max(float(np.abs(a).max()) for a in cls), max(float(np.abs(a).max()) for a in box), h), flush=True)
d.close()
print("probe done")
