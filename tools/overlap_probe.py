"""Probe (run on the GPU box): how much of the latency-bound post-process (NMS + aggregate, ~4.3 ms of a 66 ms step) hides
under another step's conv stack?  Two handles of the headline workload run interleaved on one GPU, half a step apart;
compared with one handle running the same number of steps back to back."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from uda_amd import hparams_config, weights as W
from uda_amd.infer_lib import KerasDriver

cfg = hparams_config.get_efficientdet_config("efficientdet-d0")
cfg.override(dict(image_size="1280x768", num_classes=7, loss_attenuation=True, enable_softmax=True, mc_dropout=True,
                  mc_dropoutsamp=10, mc_dropoutrate=0.05))
p = cfg.as_dict(); p["is_training_bn"] = False
w = W.init_weights(p, seed=0)
imgs = np.random.default_rng(2).integers(0, 256, (32, 768, 1280, 3), dtype=np.uint8)
ds = [KerasDriver("_", False, "efficientdet-d0", 32, False, p, weights=w, chunk_images=32) for _ in range(2)]
for d in ds:
    d.stage_images(imgs)
    for _ in range(3):
        d.run_resident(sync=True); d._collect(32)
K = 10
a = ds[0]
a.synchronize(); t0 = time.perf_counter()
for _ in range(2 * K):
    a.run_resident(sync=False); a._collect(32)
t_single = (time.perf_counter() - t0) / (2 * K)
for d in ds: d.synchronize()
t0 = time.perf_counter()
ds[0].run_resident(sync=False)
for k in range(K):
    ds[1].run_resident(sync=False)
    ds[0]._collect(32)
    if k + 1 < K: ds[0].run_resident(sync=False)
    ds[1]._collect(32)
t_pair = (time.perf_counter() - t0) / (2 * K)
print("one handle: %.2f ms/step; two interleaved handles: %.2f ms/step; fallbacks %s %s" % (
    t_single * 1e3, t_pair * 1e3, [d.nms_coop_fallbacks() for d in ds], [d.nms_coop_not_launched() for d in ds]))
