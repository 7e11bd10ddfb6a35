"""Shared helpers for the test-suite (seeded configs, weights and inputs)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import uda_amd.hparams_config as hp          # noqa: E402
import uda_amd.weights as weights_mod        # noqa: E402


def make_params(image_size="192x128", num_classes=7, model="efficientdet-d0", **over):
    c = hp.get_efficientdet_config(model)
    base = dict(image_size=image_size, num_classes=num_classes, enable_softmax=True)
    base.update(over)
    c.override(base, allow_new_keys=True)
    p = c.as_dict()
    p["is_training_bn"] = False
    return p


FULL_MC = dict(mc_dropout=True, mc_dropoutrate=0.05, mc_dropoutsamp=3, loss_attenuation=True)
HEAD_MC = dict(mc_dropout=True, mc_classheadrate=0.05, mc_boxheadrate=0.05, mc_dropoutsamp=3,
               loss_attenuation=True)
LOSS_ATT = dict(loss_attenuation=True)
PLAIN = dict()
MC_NO_ATT = dict(mc_dropout=True, mc_dropoutrate=0.1, mc_dropoutsamp=4)
BOX_ONLY_MC = dict(mc_dropout=True, mc_boxheadrate=0.1, mc_dropoutsamp=3, loss_attenuation=True)


def make_weights(params, seed=1, cls_spread=1.0):
    return weights_mod.init_weights(params, seed=seed, cls_spread=cls_spread)


def make_images(n, h, w, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (n, h, w, 3), dtype=np.uint8)
