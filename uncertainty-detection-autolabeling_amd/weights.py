"""Reference-named weight sets for the detector.

There is no checkpoint in this environment, so weight sets are created here
with the reference's initialisers, keyed by the reference's variable names so
a checkpoint converter can fill the same dict later (names: SURVEY §9.10;
efficientnet_model.py:322-329,383; efficientdet_keras.py:131,138,169,428-445,
594-625; utils_keras.py:200).

  conv / depthwise / SE kernels   N(0, sqrt(2 / fan_out))        efficientnet_model.py:87-107
  head separable convs            variance_scaling (trunc. normal, fan_in)   efficientdet_keras.py:493-494
  FPN / resample convs            glorot uniform (Keras default)
  class-predict bias              -log((1 - 0.01) / 0.01)         efficientdet_keras.py:510
  fusion weights WSM              ones                            efficientdet_keras.py:152-154

A weight set is `dict[str, np.ndarray(float32)]` with TF kernel layouts
([kh, kw, cin, cout]; depthwise [kh, kw, c, 1]).
"""
import zlib

import numpy as np

from . import arch
from .hparams_config import parse_image_size  # noqa: F401  (re-export for callers)

BN_FIELDS = ("gamma", "beta", "moving_mean", "moving_variance")


def variable_specs(config):
    """Ordered list of (name, shape, kind) for every variable on the path.

    kind ∈ conv, dw, se_w, bias0, vs_dw, vs_pw, glorot, cls_bias, wsm, bn
    """
    from . import plan as plan_mod
    plan_mod.check_model_params(config)
    bb = config["backbone_name"]
    blocks = arch.backbone_blocks(bb, config.get("backbone_config"))
    specs = []
    resample_bn = bool(config.get("apply_bn_for_resampling", True))         # efficientdet_keras.py:313-318
    cba = bool(config.get("conv_bn_act_pattern", False))                    # use_bias = not conv_bn_act_pattern (:218)

    def bn(prefix, c):
        specs.append((prefix, (c,), "bn"))

    stem = arch.stem_filters(bb)
    specs.append((bb + "/stem/conv2d/kernel", (3, 3, 3, stem), "conv"))
    bn(bb + "/stem/tpu_batch_normalization", stem)
    for i, b in enumerate(blocks):
        p = "%s/blocks_%d/" % (bb, i)
        mid = b["cin"] * b["expand"]
        convs, bns = ["conv2d", "conv2d_1"], ["tpu_batch_normalization",
                                              "tpu_batch_normalization_1",
                                              "tpu_batch_normalization_2"]
        ci = bi = 0
        if b["expand"] != 1:
            specs.append((p + convs[ci] + "/kernel", (1, 1, b["cin"], mid), "conv"))
            ci += 1
            bn(p + bns[bi], mid)
            bi += 1
        specs.append((p + "depthwise_conv2d/depthwise_kernel",
                      (b["kernel"], b["kernel"], mid, 1), "dw"))
        bn(p + bns[bi], mid)
        bi += 1
        if b["se"]:
            specs.append((p + "se/conv2d/kernel", (1, 1, mid, b["se"]), "conv"))
            specs.append((p + "se/conv2d/bias", (b["se"],), "bias0"))
            specs.append((p + "se/conv2d_1/kernel", (1, 1, b["se"], mid), "conv"))
            specs.append((p + "se/conv2d_1/bias", (mid,), "bias0"))
        specs.append((p + convs[ci] + "/kernel", (1, 1, mid, b["cout"]), "conv"))
        bn(p + bns[bi], b["cout"])

    F = config["fpn_num_filters"]
    min_l, max_l = config["min_level"], config["max_level"]
    red = arch.reduction_block_ids(blocks)
    feat_ch = [blocks[i]["cout"] for i in red]          # reduction_1..5
    in_ch = feat_ch[min_l - 1:]                          # levels min_l..5 from the backbone
    level_ch = list(in_ch)
    for lvl in range(len(in_ch) + min_l, max_l + 1):
        name = "resample_p%d" % lvl
        if level_ch[-1] != F:
            specs.append((name + "/conv2d/kernel", (1, 1, level_ch[-1], F), "glorot"))
            specs.append((name + "/conv2d/bias", (F,), "bias0"))
            if resample_bn:
                bn(name + "/bn", F)
        level_ch.append(F)

    nodes, method = plan_mod.fpn_nodes(config)
    weighted = method in ("fastattn", "attn")
    for rep in range(config["fpn_cell_repeats"]):
        ch = list(level_ch) if rep == 0 else [F] * len(level_ch)
        for n, node in enumerate(nodes):
            p = "fpn_cells/cell_%d/fnode%d/" % (rep, n)
            nfeats = len(ch)
            for i, off in enumerate(node["inputs_offsets"]):
                if ch[off] != F:
                    rp = p + "resample_%d_%d_%d" % (i, off, nfeats)
                    specs.append((rp + "/conv2d/kernel", (1, 1, ch[off], F), "glorot"))
                    specs.append((rp + "/conv2d/bias", (F,), "bias0"))
                    if resample_bn:
                        bn(rp + "/bn", F)
                if weighted:
                    specs.append((p + "WSM" + ("" if i == 0 else "_%d" % i), (), "wsm"))
            op = p + "op_after_combine%d" % nfeats
            specs.append((op + "/conv/depthwise_kernel", (3, 3, F, 1), "glorot"))
            specs.append((op + "/conv/pointwise_kernel", (1, 1, F, F), "glorot"))
            if not cba:
                specs.append((op + "/conv/bias", (F,), "bias0"))
            bn(op + "/bn", F)
            ch.append(F)

    A = len(config["aspect_ratios"]) * config["num_scales"]
    box_out = (8 if config["loss_attenuation"] else 4) * A
    for net, tag, outc, bias_kind in (("class_net", "class", config["num_classes"] * A, "cls_bias"),
                                      ("box_net", "box", box_out, "bias0")):
        for i in range(config["box_class_repeats"]):
            p = "%s/%s-%d" % (net, tag, i)
            specs.append((p + "/depthwise_kernel", (3, 3, F, 1), "vs_dw"))
            specs.append((p + "/pointwise_kernel", (1, 1, F, F), "vs_pw"))
            specs.append((p + "/bias", (F,), "bias0"))
            for lvl in range(min_l, max_l + 1):
                bn("%s/%s-%d-bn-%d" % (net, tag, i, lvl), F)
        p = "%s/%s-predict" % (net, tag)
        specs.append((p + "/depthwise_kernel", (3, 3, F, 1), "vs_dw"))
        specs.append((p + "/pointwise_kernel", (1, 1, F, outc), "vs_pw"))
        specs.append((p + "/bias", (outc,), bias_kind))
    return specs


def count_trainable(specs):
    """Trainable parameters (BN contributes gamma+beta only)."""
    n = 0
    for _, shape, kind in specs:
        n += 2 * shape[0] if kind == "bn" else int(np.prod(shape)) if shape else 1
    return n


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def _trunc_normal(rng, shape, std):
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2
    return x * std


def init_weights(config, seed=0, randomize_bn=True, cls_spread=1.0, wsm_jitter=True):
    """Create a weight set.

    randomize_bn   BN statistics drawn at random (gamma~U(.5,1.5), beta,mean~N(0,.1),
                   var~U(.5,1.5)) so that BN is not the identity.
    cls_spread     multiplies the class-predict pointwise kernel: 1.0 keeps the
                   reference init (all scores ~0.01, every anchor enters NMS);
                   ~20 gives a spread score distribution.
    wsm_jitter     fusion weights ~N(1, .5) instead of ones, so relu/normalise is exercised.
    """
    w = {}
    for name, shape, kind in variable_specs(config):
        r = _rng(seed, name)
        if kind == "bn":
            c = shape[0]
            if randomize_bn:
                w[name + "/gamma"] = r.uniform(0.5, 1.5, c)
                w[name + "/beta"] = r.normal(0, 0.1, c)
                w[name + "/moving_mean"] = r.normal(0, 0.1, c)
                w[name + "/moving_variance"] = r.uniform(0.5, 1.5, c)
            else:
                w[name + "/gamma"] = np.ones(c)
                w[name + "/beta"] = np.zeros(c)
                w[name + "/moving_mean"] = np.zeros(c)
                w[name + "/moving_variance"] = np.ones(c)
        elif kind in ("conv", "dw"):
            kh, kw, _, co = shape
            w[name] = r.normal(0, np.sqrt(2.0 / (kh * kw * co)), shape)
        elif kind in ("vs_dw", "vs_pw"):
            kh, kw, ci, _ = shape
            v = _trunc_normal(r, shape, np.sqrt(1.0 / (kh * kw * ci)) / 0.87962566103423978)
            if kind == "vs_pw" and name.startswith("class_net/class-predict"):
                v = v * cls_spread
            w[name] = v
        elif kind == "glorot":
            kh, kw, ci, co = shape
            fan_in, fan_out = kh * kw * ci, kh * kw * co
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            w[name] = r.uniform(-lim, lim, shape)
        elif kind == "bias0":
            w[name] = np.zeros(shape)
        elif kind == "cls_bias":
            w[name] = np.full(shape, -np.log((1 - 0.01) / 0.01))
        elif kind == "wsm":
            w[name] = np.asarray(r.normal(1.0, 0.5) if wsm_jitter else 1.0)
        else:  # pragma: no cover
            raise ValueError(kind)
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}


def save_weights(path, w):
    np.savez(path, **w)


def load_weights(path):
    with np.load(path) as z:
        return {k: z[k].astype(np.float32) for k in z.files}


def resolve_weights(path, config):
    """The weight set a driver's path argument names (infer_lib.KerasDriver / SavedModelDriver):
      "_" / "" / None  the reference's test mode ("do not load any ckpt", utils_keras.py:142-144): random init,
                       seed = config["uda_seed"] (default 0)
      *.npz            a weight set saved by `save_weights` (reference variable names)
      anything else    a TF2 checkpoint prefix or directory (utils_keras.restore_ckpt, :125-235), read without
                       TensorFlow by `ckpt_reader` the way the reference's driver restores (infer_lib.py:435:
                       `restore_ckpt(model, ckpt, config.moving_average_decay, skip_mismatch=False)`): EMA shadows
                       only when `moving_average_decay` > 0, a missing or mis-shaped variable raises."""
    if path is None or str(path) in ("", "_"):
        return init_weights(config, seed=int(config.get("uda_seed", 0)))
    path = str(path)
    if path.endswith(".npz"):
        return load_weights(path)
    from . import ckpt_reader
    decay = config.get("moving_average_decay", 0) or 0
    return ckpt_reader.load_checkpoint(path, config, use_ema=float(decay) > 0, skip_mismatch=False)
