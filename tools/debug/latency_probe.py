"""GPU box: batch-1 detect latency (the reference's protocol: one image per serve) - wall time of serve() against the sum of the
kernel times by kind (HIP events around every launch group) - how much of the latency is launch / host overhead?"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from common import make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
KINDS = {1: "stem", 2: "pw", 3: "dw", 4: "se", 6: "pool", 7: "mbx", 8: "sep", 16: "aggregate", 17: "nms", 18: "preprocess"}
for variant in ("full", "head"):
    over = dict(image_size="1280x768", mc_dropout=True, mc_dropoutsamp=10, loss_attenuation=True)
    over.update(dict(mc_dropoutrate=0.05) if variant == "full" else dict(mc_classheadrate=0.05, mc_boxheadrate=0.05))
    p = make_params(**over)
    w = make_weights(p, seed=0)
    img = make_images(1, 768, 1280, seed=2)
    d = KerasDriver("_", False, p["name"], 1, False, p, weights=w, chunk_images=1)
    for _ in range(5):
        d.serve(img)
    lat = []
    for _ in range(30):
        t = time.perf_counter(); d.serve(img); lat.append(time.perf_counter() - t)
    d.profile_enable(list(KINDS))
    for _ in range(10):
        d.serve(img)
    ks = {}
    for k, name in KINDS.items():
        ms, cnt = d.profile_read(k)
        ks[name] = (round(ms / 10, 3), cnt // 10)
    d.profile_enable([])
    tot = sum(v[0] for v in ks.values())
    print(variant, "p50 wall %.2f ms, min %.2f; kernel time by kind (ms, launches): %s; sum %.2f ms, launches %d" % (
        np.median(lat) * 1e3, min(lat) * 1e3, ks, tot, sum(v[1] for v in ks.values())))
    d.close()
