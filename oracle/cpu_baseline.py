"""ORACLE (test infrastructure): times the CPU restatement of the serve path on a bounded
sample for bench.py's `cpu_baseline` ("port": the build's own restatement, NOT TensorFlow —
TF 2.10 cannot be installed here or on the GPU box).  Protocol after the reference's
`ServingDriver._benchmark` / `Validate._process_val_image`: wall clock around one serve call
(src/infer_lib.py:206-224, src/validate_model.py:154-158); one warm-up on a tiny image first.
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--image-size", default="1280x768")
    ap.add_argument("--classes", type=int, default=7)
    ap.add_argument("--samples", type=int, default=2)
    ap.add_argument("--images", type=int, default=1)
    ap.add_argument("--variant", default="full")
    ap.add_argument("--model", default="efficientdet-d0")
    a = ap.parse_args()
    import numpy as np
    import torch
    from oracle import serve_ref
    from uda_amd import hparams_config, weights as weights_mod
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))     # a 1-GPU box's CPU share is 16 cores; more threads only oversubscribe
    torch.set_num_threads(cores)
    cfg = hparams_config.get_efficientdet_config(a.model)
    over = dict(image_size=a.image_size, num_classes=a.classes, mc_dropout=True, mc_dropoutsamp=a.samples,
                loss_attenuation=True, enable_softmax=True)
    over.update(dict(mc_dropoutrate=0.05) if a.variant == "full" else dict(mc_classheadrate=0.05, mc_boxheadrate=0.05))
    cfg.override(over)
    p = cfg.as_dict()
    p["is_training_bn"] = False
    w = weights_mod.init_weights(p, seed=0)
    W_, H_ = [int(v) for v in a.image_size.lower().split("x")]
    imgs = np.random.default_rng(2).integers(0, 256, (a.images, H_, W_, 3), dtype=np.uint8)
    serve_ref.serve(p, w, imgs[:, :64, :96], seed=0)          # warm-up (thread pools, lib load)
    t0 = time.perf_counter()
    serve_ref.serve(p, w, imgs, seed=0)
    dt = time.perf_counter() - t0
    units = a.images * a.samples
    print(json.dumps({"value": round(units / dt, 4), "unit": "images*MC-samples/s", "cores": torch.get_num_threads(),
                      "kind": "port", "seconds": round(dt, 2),
                      "sample": "%d image(s) %s x T=%d through oracle/serve_ref.serve (torch-CPU convs + numpy/C "
                                "post-process; restatement, not TF)" % (a.images, a.image_size, a.samples)}))


if __name__ == "__main__":
    main()
