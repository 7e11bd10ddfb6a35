"""ORACLE (test infrastructure): times the CPU restatement of the serve path on a bounded
sample for bench.py's `cpu_baseline` ("port": the build's own restatement, NOT TensorFlow -
TF 2.10 cannot be installed here or on the GPU box).

Protocol after the reference: `ServingDriver._benchmark` (src/infer_lib.py:214-224) -
warm-up calls, then `bm_runs` timed calls, mean seconds per call - plus the per-call wall clock
`Validate._process_val_image` records around `driver.serve(image)` (src/validate_model.py:154-158),
of which the median is reported.  Both legs run the reference's protocol - 3 warm-ups, 10 timed calls - on a bounded
number of images per call (`--images`, `--config1-images`): about a minute of CPU work in all.

Two legs, one JSON line:
  workload   the GPU line's workload shape (MC dropout, T = --samples) on `--images` image(s) per call
  config1    BASELINE configs[0]: 4 images per call, T = 1 (no MC), the reference's own CPU-runnable case
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed_calls(fn, warmups, runs):
    for _ in range(warmups):
        fn()
    per_call = []
    t0 = time.perf_counter()
    for _ in range(runs):
        ts = time.perf_counter()
        fn()
        per_call.append(time.perf_counter() - ts)
    total = time.perf_counter() - t0
    per_call.sort()
    return total / runs, per_call[len(per_call) // 2], total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--image-size", default="1280x768")
    ap.add_argument("--classes", type=int, default=7)
    ap.add_argument("--samples", type=int, default=10)
    ap.add_argument("--images", type=int, default=1)
    ap.add_argument("--variant", default="full")
    ap.add_argument("--model", default="efficientdet-d0")
    ap.add_argument("--warmups", type=int, default=3)
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--config1-images", type=int, default=4)
    ap.add_argument("--config1-warmups", type=int, default=3)
    ap.add_argument("--config1-runs", type=int, default=10)
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

    def params_for(mc):
        cfg = hparams_config.get_efficientdet_config(a.model)
        over = dict(image_size=a.image_size, num_classes=a.classes, loss_attenuation=True, enable_softmax=True)
        if mc:
            over.update(mc_dropout=True, mc_dropoutsamp=a.samples)
            over.update(dict(mc_dropoutrate=0.05) if a.variant == "full" else dict(mc_classheadrate=0.05, mc_boxheadrate=0.05))
        cfg.override(over)
        p = cfg.as_dict()
        p["is_training_bn"] = False
        return p

    p_mc, p_det = params_for(True), params_for(False)
    w = weights_mod.init_weights(p_mc, seed=0)
    W_, H_ = [int(v) for v in a.image_size.lower().split("x")]
    imgs = np.random.default_rng(2).integers(0, 256, (max(a.images, a.config1_images, 1), H_, W_, 3), dtype=np.uint8)
    serve_ref.serve(p_det, w, imgs[:1, :64, :96], seed=0)          # thread pools, library load
    mean_mc, p50_mc, tot_mc = timed_calls(lambda: serve_ref.serve(p_mc, w, imgs[:a.images], seed=0), a.warmups, a.runs)
    n1 = max(a.config1_images, 1)
    mean_1, p50_1, tot_1 = timed_calls(lambda: serve_ref.serve(p_det, w, imgs[:n1], seed=0), a.config1_warmups, a.config1_runs)
    units = a.images * a.samples
    print(json.dumps({
        "value": round(units / mean_mc, 4), "unit": "images*MC-samples/s", "cores": torch.get_num_threads(), "kind": "port",
        "seconds": round(tot_mc + tot_1, 2),
        "p50_call_s": round(p50_mc, 3), "mean_call_s": round(mean_mc, 3),
        "sample": "%d warm-ups + %d timed calls (the reference's protocol, infer_lib.py:214-224) of oracle/serve_ref.serve on %d "
                  "image(s) %s x T=%d (%s MC) per call, mean per call; torch-CPU convs + numpy/C post-process: restatement, "
                  "not TF" % (a.warmups, a.runs, a.images, a.image_size, a.samples, a.variant),
        "config1": {"value": round(n1 / mean_1, 4), "unit": "images/s", "p50_call_s": round(p50_1, 3), "mean_call_s": round(mean_1, 3),
                    "sample": "BASELINE configs[0]: %d warm-ups + %d timed calls, %d images %s per call, T=1 (no MC)"
                              % (a.config1_warmups, a.config1_runs, n1, a.image_size)}}))


if __name__ == "__main__":
    main()
