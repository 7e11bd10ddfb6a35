"""ORACLE (test infrastructure): the whole `serve` pipeline on the CPU.

preprocess -> network x T (injected Philox masks) -> postprocess_global, i.e. what
`EfficientDetModel.call(images, pre_mode="infer", post_mode="global")` computes
(src/efficientdet_keras.py:1118-1146).  Used as the checker in tests / smoke() and as the
timed CPU baseline ("restatement, not TF") of bench.py.
"""
import time

import numpy as np

from . import effdet_ref, philox_ref, post_ref, preprocess_ref


def serve(params, weights, images_u8, seed=0, per_class=False):
    H, W = post_ref.parse_image_size(params["image_size"])
    mean, std = params["mean_rgb"], params["stddev_rgb"]
    x, scales = preprocess_ref.preprocess(images_u8, (H, W), mean, std)
    sites = effdet_ref.dropout_sites(params)
    T = int(params["mc_dropoutsamp"]) if params["mc_dropout"] else 1
    masks = philox_ref.make_masks(sites, seed, x.shape[0], T) if sites else None
    cls, box = effdet_ref.forward(weights, params, x, masks)
    post = post_ref.postprocess_per_class if per_class else post_ref.postprocess_global
    return post(params, cls, box, scales)


def timed_serve(params, weights, images_u8, seed=0, threads=None):
    """(seconds, units) for one serve call; units = images x MC samples."""
    import torch
    if threads:
        torch.set_num_threads(int(threads))
    t0 = time.perf_counter()
    serve(params, weights, images_u8, seed)
    dt = time.perf_counter() - t0
    T = int(params["mc_dropoutsamp"]) if params["mc_dropout"] else 1
    return dt, len(images_u8) * T
