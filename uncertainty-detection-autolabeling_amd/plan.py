"""Lower the detector (config + reference-named weights) to the C-ABI op list.

This is the Python host orchestration that the reference keeps in
`EfficientDetNet.__init__/call` (efficientdet_keras.py:850-1070) and
`efficientnet_model.Model._build/call` (backbone/efficientnet_model.py:731-909):
it decides *which* ops run on *which* tensors; the C-ABI library executes them.

What the lowering adds over the reference's graph:
  * the MC sample axis is explicit.  A tensor is "per sample" (rows = images x T) only
    downstream of an active dropout site; everything upstream is computed once per image
    and shared by the T samples (stem for full MC; backbone + BiFPN for head-only MC,
    where SpatialDropout2D(0.0) in the backbone is the identity).  The reference re-runs
    the whole network T times (efficientdet_keras.py:999-1024).
  * BN is folded to one scale/shift pair per channel; BiFPN fusion weights are
    pre-normalised (relu(w_i)/(sum+1e-4), efficientdet_keras.py:101-108).
  * activation memory is planned by liveness into one arena per chunk of images.
"""
import numpy as np

from . import arch, capi
from .hparams_config import get_feat_sizes, parse_image_size

ALIGN = 256  # floats


def dw_tiles_x(C, Wo, k, stride):
    """gridDim.x of the depthwise kernel (mirror of dw_geometry in csrc/kernels_conv.hip)."""
    C4 = C // 4
    cap = 16 if k == 5 else 64
    ncc = (C4 + cap - 1) // cap
    t = (C4 + ncc - 1) // ncc
    p = max(1, 256 // t)
    x = 1 if k == 5 else (4 if stride == 1 else 2)
    need = (Wo + x - 1) // x
    p = min(p, need)
    if k == 5:
        nb = (need + p - 1) // p
        p = (need + nb - 1) // nb
    return (Wo + p * x - 1) // (p * x)


def dw_rows(k, stride):
    """output rows per depthwise block (mirror of dw_rows in csrc/kernels_conv.hip)."""
    return 16 if (k == 5 and stride == 1) else 8


def dw_tiles(C, Ho, Wo, k, stride):
    """SE tile sums one depthwise launch leaves per sample row (mirror of dw_tiles in csrc)."""
    return -(-Ho // dw_rows(k, stride)) * dw_tiles_x(C, Wo, k, stride)


PW_SCHEMES = ("f16x2", "bf16x3", "bf16x2", "f32")
PW_SCHEME_DEFAULT = "f16x2"


def pw_scheme():
    """Split scheme of the 1x1 contractions (mirror of parse_pw_scheme in csrc/uda_api.hip): UDA_PW_SCHEME =
    f16x2 (two fp16 pieces, three cross terms, ~2^-22 per product: the default) | bf16x3 (three bf16 pieces, six terms,
    ~2^-24) | bf16x2 (two bf16 pieces, three terms, ~2^-17) | f32 (exact f32-input MFMA kernels, unfused);
    the older UDA_PW_TERMS = 6 | 3 | 0 selects the last three when UDA_PW_SCHEME is not set."""
    import os
    v = os.environ.get("UDA_PW_SCHEME")
    if v:
        if v not in PW_SCHEMES:
            raise ValueError("UDA_PW_SCHEME=%r: expected one of %s" % (v, ", ".join(PW_SCHEMES)))
        return v
    t = os.environ.get("UDA_PW_TERMS")
    if t is None or t == "":
        return PW_SCHEME_DEFAULT
    t = int(t)
    if t not in (0, 3, 6):
        raise ValueError("UDA_PW_TERMS=%r: expected 6, 3 or 0" % (t,))
    return {0: "f32", 3: "bf16x2", 6: "bf16x3"}[t]


def mbx_deep(cin):
    """Cin > 48: the deep variant of the fused kernel (mbxd_kernel in csrc/kernels_pwb.hip)."""
    return cin > 48


def mbx_tile(k, stride, cin=16, Ho=None, Wo=None):
    """(TH, TW) output tile of the fused expand+depthwise kernels (mirror of mbx_cfg / mbxd_cfg / mbxd_wide in csrc)."""
    if mbx_deep(cin):       # mirror of mbxd_cfg
        if stride == 1:
            import os
            normal, wide = ((12, 16), (8, 20)) if k == 3 else ((8, 16), (6, 20))
            if Ho is not None and int(os.environ.get("UDA_MBXD_WIDE", "1")):
                slots = lambda t: -(-Ho // t[0]) * t[0] * -(-Wo // t[1]) * t[1]
                if slots(wide) < slots(normal):      # mbxd_wide: 20-column tiles cover the map with fewer output slots
                    return wide
            return normal
        return (7, 8) if k == 3 else (4, 10)
    import os
    if pw_scheme() == "f32" or not int(os.environ.get("UDA_MBX_BF16", "1")):
        return (8, 16) if stride == 1 else ((4, 16) if k == 3 else (4, 8))      # f32-MFMA fallback kernel (mbx_cfg)
    s2_k3 = (7, 8) if os.environ.get("UDA_MBXB_S2_TILE", "78") == "78" else (4, 12)     # (A/B builds: -DUDA_MBXB_S2_TILE=412)
    return ((12, 16) if k == 3 else (8, 16)) if stride == 1 else (s2_k3 if k == 3 else (4, 10))


def mbx_tiles(Ho, Wo, k, stride, cin=16):
    th, tw = mbx_tile(k, stride, cin, Ho, Wo)
    return -(-Ho // th) * -(-Wo // tw)


def mbx_supported(cin, cmid, k, stride):
    import os
    if not (int(os.environ.get("UDA_FUSE_MBX", "1")) and cin % 8 == 0 and cmid % 4 == 0 and k in (3, 5)):
        return False
    if pw_scheme() == "bf16x3" and not int(os.environ.get("UDA_FUSE_MBX6", "1")):
        # six cross terms = float32-equivalent products EVERYWHERE.  The fused MBConv kernels have three-piece variants
        # (mbxb_kernel / mbxd_kernel<..., PARTS = 3>, csrc/kernels_pwb.hip) and stay fused; UDA_FUSE_MBX6=0 restores the
        # round-2 behaviour (stand-alone six-term 1x1 convs + depthwise) for A/B runs
        return False
    if mbx_deep(cin):       # mirror of mbxd_supported: 16-deep k-steps of Cin + 1 in {6, 8, 13, 14}, split-bf16 path on
        return (int(os.environ.get("UDA_FUSE_MBXD", "1")) and pw_scheme() != "f32"
                and int(os.environ.get("UDA_MBX_BF16", "1"))
                and (stride == 1 or (stride == 2 and int(os.environ.get("UDA_FUSE_MBXD_S2", "1"))))
                and (cin + 1 + 15) // 16 in (6, 8, 13, 14)
                and (cin + 1 + 15) // 16 <= int(os.environ.get("UDA_MBXD_MAXKSF", "14")))
    return 16 <= cin and stride in (1, 2)


def sepf_supported(C, Cout):
    """The BiFPN fusion of a node can be computed inside its separable conv (mirror of sepf_supported / sepf_lds_bytes in
    csrc/kernels_sep.hip: an 18 x 18 x (32 + 4) float32 tile + the weight fragments of two k-steps, two blocks per CU)."""
    import os
    sch = pw_scheme()
    if sch == "f32" or not int(os.environ.get("UDA_FUSE_IN", "1")):
        return False
    npc = 3 if sch == "bf16x3" else 2
    lds = 18 * 18 * 36 * 4 + 2 * (-(-Cout // 32)) * npc * 1024
    return C % 8 == 0 and 16 <= C <= 128 and Cout % 4 == 0 and 4 <= Cout <= 128 and lds <= 80 * 1024


def sep_tin_supported(C, Cout):
    """A plain separable conv can take a DEFERRED dropout site of its producer (mirror of sep_tin_supported in
    csrc/kernels_sep.hip): 64 .. 128 input channels, two to four 32-column tiles in one block, the per-sample epilogue staging
    inside the A image."""
    import os
    sch = pw_scheme()
    if sch == "f32" or not int(os.environ.get("UDA_DEFER_HEAD", "1")):
        return False
    if C % 8 or not 64 <= C <= 128 or Cout < 1:
        return False
    ks, ntl, npc = -(-C // 16), -(-Cout // 32), (3 if sch == "bf16x3" else 2)
    if not 2 <= ntl <= 4:
        return False
    a_img = 128 * (ks * 64 + 16) if sch == "bf16x3" else npc * 128 * (ks * 32 + 16)
    stg = 4 * 32 * ntl * 32 * 4 if Cout % 4 else 4 * 32 * 68 * 4
    return stg <= a_img and a_img + ks * ntl * npc * 1024 <= 120 * 1024


def same_out(n, s):
    return -(-n // s)


# ---------------------------------------------------------------------------------------------------------------------
# Every key of hparams_config.default_detection_configs() and what the HIP path does with it.  The boundary promises the
# reference's `model_params` dict: a key is either CONSUMED (the planner / the post-process reads it and the result
# changes with it), INERT at inference (the reference itself never reads it on the serve path, or reads it only under
# training=True), or checked by `check_model_params`, which raises the reference's kind of ValueError for a value the
# path does not implement - never a silently different network (tests/test_plan_host.py walks this table).
ACT_CODES = {"swish": capi.ACT_SWISH, "silu": capi.ACT_SWISH, "swish_native": capi.ACT_SWISH,
             "relu": capi.ACT_RELU, "relu6": capi.ACT_RELU6, "hswish": capi.ACT_HSWISH, "mish": capi.ACT_MISH}
MODEL_PARAM_HANDLING = {
    # --- network structure (efficientdet_keras.py:850-970, efficientnet_model.py:731-834)
    "name": "inert: a label (the structure comes from the keys below)",
    "backbone_name": "consumed: arch.backbone_blocks (efficientnet-b0..b7; anything else raises)",
    "backbone_config": "consumed: arch.backbone_blocks (custom block table)",
    "image_size": "consumed", "num_classes": "consumed", "min_level": "consumed", "max_level": "consumed",
    "num_scales": "consumed", "aspect_ratios": "consumed", "anchor_scale": "consumed",
    "mean_rgb": "consumed", "stddev_rgb": "consumed",
    "box_class_repeats": "consumed", "fpn_cell_repeats": "consumed", "fpn_num_filters": "consumed",
    "act_type": "consumed: swish | silu | swish_native | relu | relu6 | hswish | mish; srelu raises (utils.py:42-59)",
    "separable_conv": "checked: False (dense 3x3 convs in BiFPN and heads) raises",
    "apply_bn_for_resampling": "consumed (efficientdet_keras.py:313-318)",
    "conv_after_downsample": "consumed (efficientdet_keras.py:331-338)",
    "conv_bn_act_pattern": "consumed (efficientdet_keras.py:218,229-236)",
    "fpn_name": "checked: None | bifpn | bifpn_dyn; qufpn raises",
    "fpn_weight_method": "consumed: fastattn | attn | sum; channel_attn / channel_fastattn raise",
    "fpn_config": "checked: None, or {nodes, weight_method} with at most three inputs per node",
    "heads": "checked: must be ['object_detection'] (segmentation head is not on the path)",
    "data_format": "checked: channels_last only",
    "survival_prob": "inert: drop_connect acts under training=True only (efficientdet_keras.py:464-465)",
    "is_training_bn": "inert: BN layers run with training=False at inference (moving statistics)",
    "strategy": "inert: selects the BN class / TPU placement, same inference arithmetic",
    "mixed_precision": "inert: the HIP path always computes in float32-class arithmetic (a superset)",
    "grad_checkpoint": "inert: training memory option",
    # --- uncertainty switches
    "loss_attenuation": "consumed", "uncert_adjust_method": "consumed (l-norm | n-flow | falsedec | sample; else raises)",
    "decode_nsamples": "consumed", "mc_dropout": "consumed", "mc_dropoutrate": "consumed", "mc_classheadrate": "consumed",
    "mc_boxheadrate": "consumed", "mc_dropoutsamp": "consumed", "enable_softmax": "consumed",
    "nms_configs": "consumed (method hard | gaussian; anything else raises; pyfunc is served by uda_nms_np)",
    "tflite_max_detections": "inert: TFLite export only",
    "max_instances_per_image": "inert: training / eval padding",
    "clip_min_uncert": "inert: loss only", "clip_max_uncert": "inert: loss only",
    "calibrate_classification": "consumed by calibration.py (above the driver)", "calib_method_class": "consumed by calibration.py",
    "calibrate_regression": "consumed by calibration.py", "calib_method_box": "consumed by calibration.py",
    "infer_draw_uncert": "consumed by visualize", "label_map": "consumed by infer_lib / writers",
    # --- training-only keys the shipped YAMLs set
    "assign_gt_box": "inert: validation matching above the driver", "early_stopping_patience": "inert: training",
    "count_classes": "inert: training", "boxloss_type": "inert: training", "save_freq": "inert: training",
    "sample_images": "inert: training", "sample_images_freq": "inert: training", "save_train_images": "inert: training",
    "autoaugment_policy": "inert: training", "map_freq": "inert: training", "box_loss_weight": "inert: training",
    "moving_average_decay": "consumed by weights.resolve_weights (EMA shadows)",
}


def act_code(cfg):
    """uda_act of config.act_type; the reference's ValueError for anything utils.activation_fn does not know, and for the one
    it knows that has no kernel here (srelu: it carries a trainable beta)."""
    name = cfg.get("act_type", "swish")
    if name in ACT_CODES:
        return ACT_CODES[name]
    if name == "srelu":
        raise ValueError("act_type %r is not available on the HIP path (implemented: %s)" % (name, ", ".join(sorted(ACT_CODES))))
    raise ValueError("Unsupported act_type {}".format(name))


def fpn_nodes(cfg):
    """(nodes, weight_method) the way FPNCells / FPNCell pick them (efficientdet_keras.py:773-781,811-834):
    config.fpn_config when given, else fpn_configs.get_fpn_config(fpn_name, ...)."""
    lo, hi = cfg["min_level"], cfg["max_level"]
    fc = cfg.get("fpn_config")
    if fc:
        fc = fc if isinstance(fc, dict) else fc.as_dict()
        nodes = [dict(feat_level=int(n["feat_level"]), inputs_offsets=[int(o) for o in n["inputs_offsets"]]) for n in fc["nodes"]]
        for n in fc["nodes"]:
            if n.get("weight_method") not in (None, fc.get("weight_method")):
                raise ValueError("fpn_config: per-node weight_method (qufpn) is not available on the HIP path")
        return nodes, fc.get("weight_method") or "fastattn"
    name = cfg.get("fpn_name")
    if name not in (None, "", "bifpn", "bifpn_dyn"):
        if name == "qufpn":
            raise ValueError("fpn_name 'qufpn' is not available on the HIP path (bifpn only)")
        raise KeyError(name)        # fpn_configs.get_fpn_config: name_to_config[fpn_name]
    return arch.bifpn_nodes(lo, hi), cfg.get("fpn_weight_method") or "fastattn"


def check_model_params(cfg):
    """Raise for every architecture switch of `model_params` whose non-default value the HIP path does not implement
    (a planner that builds the default network whatever the switch says is the worst failure mode of a
    drop-in).  Called first thing by Plan.__init__ and by weights.variable_specs."""
    act_code(cfg)
    if not cfg.get("separable_conv", True):
        raise ValueError("separable_conv=False (dense 3x3 convolutions in the BiFPN and the heads) is not available on the HIP path")
    if cfg.get("data_format", "channels_last") != "channels_last":
        raise ValueError("data_format %r: the HIP path is NHWC (channels_last) only" % (cfg.get("data_format"),))
    heads = cfg.get("heads") or ["object_detection"]
    if list(heads) != ["object_detection"]:
        raise ValueError("heads %r: only ['object_detection'] is on the HIP path" % (list(heads),))
    nodes, method = fpn_nodes(cfg)
    if method not in ("fastattn", "attn", "sum"):
        if method in ("channel_attn", "channel_fastattn"):
            raise ValueError("fpn weight_method %r (per-channel fusion weights) is not available on the HIP path" % method)
        raise ValueError("unknown weight_method %s" % method)
    lo, hi = cfg["min_level"], cfg["max_level"]
    for n in nodes:
        if not 1 <= len(n["inputs_offsets"]) <= capi.MAX_FUSE:
            raise ValueError("fpn node with %d inputs: the HIP fusion takes 1..%d" % (len(n["inputs_offsets"]), capi.MAX_FUSE))
        if not lo <= n["feat_level"] <= hi:
            raise ValueError("fpn node at level %d outside [%d, %d]" % (n["feat_level"], lo, hi))
    unknown = sorted(k for k in cfg if k not in MODEL_PARAM_HANDLING and not k.startswith("uda_"))
    return unknown      # keys the reference's defaults do not have (callers' own additions): reported, not refused


class _Buf:
    __slots__ = ("H", "W", "C", "per_sample", "kind", "level", "offset", "first", "last", "name")

    def __init__(self, H, W, C, per_sample, kind=0, level=0, name=""):
        self.H, self.W, self.C, self.per_sample = int(H), int(W), int(C), bool(per_sample)
        self.kind, self.level, self.offset = kind, level, 0
        self.first, self.last, self.name = None, None, name


class Plan:
    """bufs / ops / drop sites / weight blob / anchors ready for `uda_create`."""

    def __init__(self, config, weights, chunk_images=1, max_images=1, post_only=False):
        self.cfg = dict(config)
        self.unknown_keys = check_model_params(self.cfg)
        self.act = act_code(self.cfg)
        self.post_only = bool(post_only)
        self.w = weights
        self.T = arch.mc_flags(self.cfg)[2]
        self.cls_stacked, self.box_stacked, _ = arch.mc_flags(self.cfg)
        # The sample axis exists when there is more than one sample - or when the caller says so (`uda_force_sample_axis`: a rank
        # of a sample-sharded serve that runs ONE of the T samples must lower the network exactly like a handle that runs
        # several, or its heads differ from theirs in the last bit: which kernel variant an op takes follows the axis)
        self.sample_axis = self.T > 1 or bool(self.cfg.get("uda_force_sample_axis"))
        self.chunk_images = int(chunk_images)
        self.max_images = int(max_images)
        self.bufs, self.ops, self.blob, self.blob_len = [], [], [], 0
        self.sites = []          # (name, channels, rate)
        self.site_index = {}
        self.buffer_names = {}
        import os
        self.defer_dropout = bool(int(os.environ.get("UDA_DEFER_DROPOUT", "1")))
        self.fuse_sep = bool(int(os.environ.get("UDA_FUSE_SEP", "1"))) and pw_scheme() != "f32"
        self.fuse_proj = (bool(int(os.environ.get("UDA_FUSE_PROJ", "1"))) and pw_scheme() != "f32"
                          and bool(int(os.environ.get("UDA_MBX_BF16", "1"))) and bool(int(os.environ.get("UDA_FUSE_MBX", "1"))))
        if self.post_only:
            self._post_only_layout()
            return
        self._build_sites()
        self._lower()
        self._plan_memory()

    def _post_only_layout(self):
        """A handle that only post-processes injected head outputs (postprocess.generate_detections on arrays that
        came from elsewhere, ensemble aggregation): no ops, no weights, no arena - just the pyramid geometry and
        which heads carry the sample axis (the reference's stacking rule, efficientdet_keras.py:1026-1049)."""
        cfg = self.cfg
        fs = get_feat_sizes(cfg["image_size"], cfg["max_level"])
        self.level_hw = [tuple(fs[l]) for l in range(cfg["min_level"], cfg["max_level"] + 1)]
        self.cls_stacked_dev = bool(self.cls_stacked and self.sample_axis)
        self.box_stacked_dev = bool(self.box_stacked and self.sample_axis)
        self._buf(1, 1, 4, False, name="unused")
        self.arena_floats = ALIGN
        self.fpn_out, self.head_out = [], {"class": [], "box": []}

    # ------------------------------------------------------------------ weights
    def _pack(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        off = self.blob_len
        pad = (-arr.size) % 4
        self.blob.append(arr)
        if pad:
            self.blob.append(np.zeros(pad, np.float32))
        self.blob_len += arr.size + pad
        return off

    def _bn(self, prefix):
        g, b = self.w[prefix + "/gamma"], self.w[prefix + "/beta"]
        m, v = self.w[prefix + "/moving_mean"], self.w[prefix + "/moving_variance"]
        scale = (g / np.sqrt(v + np.float32(arch.BN_EPS))).astype(np.float32)
        shift = (b - m * scale).astype(np.float32)
        return self._pack(scale), self._pack(shift)

    # ------------------------------------------------------------------ dropout sites
    def _build_sites(self):
        cfg = self.cfg
        if not cfg["mc_dropout"]:
            return
        r_bb, r_cls, r_box = arch.dropout_rates(cfg)
        for i, b in enumerate(arch.backbone_blocks(cfg["backbone_name"], cfg.get("backbone_config"))):
            mid = b["cin"] * b["expand"]
            if b["expand"] != 1:
                self.sites.append(("blocks_%d/expand" % i, mid, r_bb))
            self.sites.append(("blocks_%d/dw" % i, mid, r_bb))
        F = cfg["fpn_num_filters"]
        for tag, r in (("class", r_cls), ("box", r_box)):
            for i in range(cfg["box_class_repeats"]):
                for lvl in range(cfg["min_level"], cfg["max_level"] + 1):
                    self.sites.append(("%s-%d-%d" % (tag, i, lvl), F, r))
        self.site_index = {name: i for i, (name, _, _) in enumerate(self.sites)}

    def _site(self, name):
        """site index if the site is active (rate > 0), else -1."""
        if name not in self.site_index:
            return -1
        i = self.site_index[name]
        return i if self.sites[i][2] > 0 else -1

    # ------------------------------------------------------------------ graph building
    def _buf(self, H, W, C, per_sample, kind=0, level=0, name=""):
        self.bufs.append(_Buf(H, W, C, per_sample and self.sample_axis, kind, level, name))
        if name:
            self.buffer_names[name] = len(self.bufs) - 1
        return len(self.bufs) - 1

    def _op(self, kind, ins, out, **kw):
        o = dict(kind=kind, ins=list(ins), out=out, se_scale=-1, se_partial=-1, residual=-1, k=0,
                 stride=1, act=capi.ACT_NONE, w_off=-1, bias_off=-1, bn_scale_off=-1, bn_shift_off=-1,
                 se_w1_off=-1, se_b1_off=-1, se_w2_off=-1, se_b2_off=-1, se_mid=0, drop_site=-1,
                 resample=[0, 0, 0], fuse_w=[0.0, 0.0, 0.0], drop_site2=-1, w2_off=-1, bn2_scale_off=-1,
                 bn2_shift_off=-1, launch_group=0, fuse_in=0, fuse_act=capi.ACT_NONE)
        o.update(kw)
        self.ops.append(o)
        return out

    def _pw(self, x, cout, kernel, name, bias=None, bn=None, act=capi.ACT_NONE, site=-1, se=-1,
            residual=-1, out_kind=0, level=0):
        xb = self.bufs[x]
        ps = xb.per_sample or site >= 0 or (se >= 0 and self.bufs[se].per_sample)
        out = self._buf(xb.H, xb.W, cout, ps, out_kind, level, name)
        kw = dict(w_off=self._pack(self.w[kernel]), act=act, drop_site=site, se_scale=se, residual=residual)
        if bias is not None:
            kw["bias_off"] = self._pack(self.w[bias])
        if bn is not None:
            kw["bn_scale_off"], kw["bn_shift_off"] = self._bn(bn)
        return self._op(capi.OP_PW, [x], out, **kw)

    def _dw(self, x, k, stride, kernel, name, bn=None, act=capi.ACT_NONE, site=-1, with_se=False, defer_site=False):
        """defer_site: the dropout site after this depthwise is NOT applied here.  Its keep-scale is per
        (sample, channel) and the consumers are the SE squeeze (a spatial mean) and a 1x1 conv, both linear in a
        per-channel scale of their input: the SE op folds it into the mean and into the gate, so the depthwise
        output stays one row per image instead of one per (image, sample)."""
        xb = self.bufs[x]
        Ho, Wo = same_out(xb.H, stride), same_out(xb.W, stride)
        if defer_site:
            site = -1
        ps = xb.per_sample or site >= 0
        out = self._buf(Ho, Wo, xb.C, ps, name=name)
        kw = dict(k=k, stride=stride, w_off=self._pack(self.w[kernel]), act=act, drop_site=site)
        if bn is not None:
            kw["bn_scale_off"], kw["bn_shift_off"] = self._bn(bn)
        if with_se:
            kw["se_partial"] = self._buf(dw_tiles(xb.C, Ho, Wo, k, stride), 1, xb.C, ps, name=name + "/se_partial")
        self._op(capi.OP_DW, [x], out, **kw)
        return out, kw.get("se_partial", -1)

    def _sepconv(self, x, cout, dw_kernel, pw_kernel, name, bias=None, bn=None, act=capi.ACT_NONE, site=-1,
                 out_kind=0, level=0, fusion=None, in_site=-1):
        """SeparableConv2D = depthwise 3x3 (no bias / BN / act) -> 1x1 + bias (+BN)(+act)(+dropout).  One fused op
        when the kernel supports the channel count, else the depthwise / pointwise pair.
        fusion = dict(ins, resample, fuse_w, H, W): the conv's input is the BiFPN fusion of `ins` (x is None), computed by
        the conv kernel for its tile - the caller has checked sepf_supported."""
        if fusion is not None:
            ibs = [self.bufs[i] for i in fusion["ins"]]
            ps = any(b.per_sample for b in ibs) or site >= 0
            out = self._buf(fusion["H"], fusion["W"], cout, ps, out_kind, level, name)
            kw = dict(k=3, stride=1, w_off=self._pack(self.w[pw_kernel]), w2_off=self._pack(self.w[dw_kernel]), act=act,
                      drop_site=site, fuse_in=1, fuse_act=fusion.get("act", capi.ACT_SWISH),
                      resample=(list(fusion["resample"]) + [0, 0, 0])[:3],
                      fuse_w=(list(fusion["fuse_w"]) + [0, 0, 0])[:3])
            if bias is not None:
                kw["bias_off"] = self._pack(self.w[bias])
            if bn is not None:
                kw["bn_scale_off"], kw["bn_shift_off"] = self._bn(bn)
            return self._op(capi.OP_SEP, list(fusion["ins"]), out, **kw)
        xb = self.bufs[x]
        assert in_site < 0 or (self.fuse_sep and sep_tin_supported(xb.C, cout) and not xb.per_sample), "deferred site without a taker"
        if not (self.fuse_sep and xb.C % 8 == 0 and 16 <= xb.C <= 128):
            d, _ = self._dw(x, 3, 1, dw_kernel, name + "/dw")
            return self._pw(d, cout, pw_kernel, name, bias=bias, bn=bn, act=act, site=site, out_kind=out_kind, level=level)
        ps = xb.per_sample or site >= 0 or in_site >= 0
        out = self._buf(xb.H, xb.W, cout, ps, out_kind, level, name)
        kw = dict(k=3, stride=1, w_off=self._pack(self.w[pw_kernel]), w2_off=self._pack(self.w[dw_kernel]), act=act,
                  drop_site=site, drop_site2=in_site)
        if bias is not None:
            kw["bias_off"] = self._pack(self.w[bias])
        if bn is not None:
            kw["bn_scale_off"], kw["bn_shift_off"] = self._bn(bn)
        return self._op(capi.OP_SEP, [x], out, **kw)

    def _mark_launch_group(self, first):
        """ops[first:] were just emitted for the pyramid levels of one head layer.  If they are fused separable convs of one
        shape class (same channels, activation, sample axes) and mutually independent, the first one is marked with their
        count: the executor may run them as one launch.  _plan_memory keeps every buffer the run touches alive to its end
        (concurrent problems must not reuse each other's freed inputs)."""
        import os
        run = self.ops[first:]
        if len(run) < 2 or len(run) > 8 or not int(os.environ.get("UDA_SEP_MULTI", "1")):
            return
        if any(o["kind"] != capi.OP_SEP for o in run):
            return

        def shape_class(o):
            ib, ob = self.bufs[o["ins"][0]], self.bufs[o["out"]]
            return (ib.C, ob.C, o["act"], ib.per_sample, ob.per_sample, o["drop_site2"] >= 0)
        if len({shape_class(o) for o in run}) != 1:
            return
        outs = {o["out"] for o in run}
        if any(o["ins"][0] in outs for o in run) or len(outs) != len(run):
            return
        run[0]["launch_group"] = len(run)

    def _resample(self, x, th, tw, prefix, name):
        """ResampleFeatureMap.call: optional 1x1+BN to F channels, then (mode for the consumer)."""
        F = self.cfg["fpn_num_filters"]
        xb = self.bufs[x]
        bn = prefix + "/bn" if self.cfg.get("apply_bn_for_resampling", True) else None      # (:313-318)
        if xb.C != F and xb.H > th and xb.W > tw and self.cfg.get("conv_after_downsample", False):
            # downsampling with conv_after_downsample: pool the wide tensor first, 1x1 (+BN) on the small map (:331-338)
            pooled = self._op(capi.OP_POOL, [x], self._buf(th, tw, xb.C, xb.per_sample, name=name + "/pool"),
                              resample=[capi.RS_MAXPOOL, 0, 0], fuse_w=[1.0, 0, 0])
            return self._pw(pooled, F, prefix + "/conv2d/kernel", name + "/conv", bias=prefix + "/conv2d/bias", bn=bn), capi.RS_NONE
        if xb.C != F:
            x = self._pw(x, F, prefix + "/conv2d/kernel", name + "/conv", bias=prefix + "/conv2d/bias", bn=bn)
            xb = self.bufs[x]
        if xb.H > th and xb.W > tw:
            return x, capi.RS_MAXPOOL
        if xb.H <= th and xb.W <= tw:
            return x, (capi.RS_NEAREST_UP if (xb.H < th or xb.W < tw) else capi.RS_NONE)
        raise ValueError("Incompatible Resampling : feat shape {}x{} target_shape: {}x{}".format(
            xb.H, xb.W, th, tw))

    def _lower(self):
        cfg, w = self.cfg, self.w
        bb = cfg["backbone_name"]
        H, W = parse_image_size(cfg["image_size"])
        blocks = arch.backbone_blocks(bb, cfg.get("backbone_config"))
        img = self._buf(H, W, 3, False, kind=1, name="image")
        # ---- backbone
        sc, sh = self._bn(bb + "/stem/tpu_batch_normalization")
        x = self._op(capi.OP_STEM, [img],
                     self._buf(same_out(H, 2), same_out(W, 2), arch.stem_filters(bb), False, name="stem"),
                     w_off=self._pack(w[bb + "/stem/conv2d/kernel"]), bn_scale_off=sc, bn_shift_off=sh,
                     act=self.act, k=3, stride=2)
        reductions = []
        red_ids = set(arch.reduction_block_ids(blocks))
        pending_proj = None      # (gate buffer, projection kernel, BN name) of a block whose 1x1 projection the next block absorbs
        for i, b in enumerate(blocks):
            p = "%s/blocks_%d/" % (bb, i)
            bn_names = [p + "tpu_batch_normalization" + ("" if j == 0 else "_%d" % j) for j in range(3)]
            inp, nb = x, 0
            deferred = -1
            mid = b["cin"] * b["expand"]
            swish = self.act == capi.ACT_SWISH     # the fused MBConv kernels fold the swish into their BN scales: other activations stay unfused
            if b["expand"] != 1 and swish and mbx_supported(b["cin"], mid, b["kernel"], b["stride"]):
                # fused expand + depthwise: the expanded tensor stays on-chip
                xb = self.bufs[x]
                Ho, Wo = same_out(xb.H, b["stride"]), same_out(xb.W, b["stride"])
                s0, s1 = self._site("blocks_%d/expand" % i), self._site("blocks_%d/dw" % i)
                ps = xb.per_sample or s0 >= 0 or s1 >= 0
                kw = {}
                if pending_proj is not None:        # x is the previous block's gated-depthwise tensor, not its output
                    gate_b, proj_k, proj_bn = pending_proj
                    ps = ps or self.bufs[gate_b].per_sample
                    kw.update(se_scale=gate_b, se_w1_off=self._pack(w[proj_k]), se_mid=b["cin"])
                    kw["se_b1_off"], kw["se_w2_off"] = self._bn(proj_bn)
                    pending_proj = None
                out = self._buf(Ho, Wo, mid, ps, name="blocks_%d/dw" % i)
                part = -1
                kw.update(k=b["kernel"], stride=b["stride"], w_off=self._pack(w[p + "conv2d/kernel"]),
                          drop_site=s0, drop_site2=s1, act=capi.ACT_SWISH,
                          w2_off=self._pack(w[p + "depthwise_conv2d/depthwise_kernel"]))
                kw["bn_scale_off"], kw["bn_shift_off"] = self._bn(bn_names[0])
                kw["bn2_scale_off"], kw["bn2_shift_off"] = self._bn(bn_names[1])
                if b["se"]:
                    part = self._buf(mbx_tiles(Ho, Wo, b["kernel"], b["stride"], b["cin"]), 1, mid, ps,
                                     name="blocks_%d/dw/se_partial" % i)
                    kw["se_partial"] = part
                x = self._op(capi.OP_MBX, [x], out, **kw)
                nb = 2
                proj = p + "conv2d_1/kernel"
            else:
                if b["expand"] != 1:
                    x = self._pw(x, mid, p + "conv2d/kernel", "blocks_%d/expand" % i,
                                 bn=bn_names[nb], act=self.act, site=self._site("blocks_%d/expand" % i))
                    nb += 1
                    proj = p + "conv2d_1/kernel"
                else:
                    proj = p + "conv2d/kernel"
                dsite = self._site("blocks_%d/dw" % i)
                # a shared (per-image) depthwise input + SE: the dropout after the depthwise is deferred into
                # the SE gate, the depthwise runs once per image (block 0 under full MC dropout)
                deferred = dsite if (dsite >= 0 and b["se"] and not self.bufs[x].per_sample and self.defer_dropout) else -1
                x, part = self._dw(x, b["kernel"], b["stride"], p + "depthwise_conv2d/depthwise_kernel",
                                   "blocks_%d/dw" % i, bn=bn_names[nb], act=self.act,
                                   site=dsite, with_se=bool(b["se"]), defer_site=deferred >= 0)
                nb += 1
            gate = -1
            if b["se"]:
                xb = self.bufs[x]
                gate = self._op(capi.OP_SE, [part, x],
                                self._buf(1, 1, xb.C, xb.per_sample or deferred >= 0, name="blocks_%d/se" % i),
                                drop_site=deferred, act=self.act,
                                k=b["kernel"], stride=b["stride"], se_mid=b["se"],
                                se_w1_off=self._pack(w[p + "se/conv2d/kernel"]),
                                se_b1_off=self._pack(w[p + "se/conv2d/bias"]),
                                se_w2_off=self._pack(w[p + "se/conv2d_1/kernel"]),
                                se_b2_off=self._pack(w[p + "se/conv2d_1/bias"]))
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            absorb = (self.fuse_proj and swish and nxt is not None and b["expand"] == 1 and gate >= 0 and not b["skip"] and b["cout"] == 16
                      and nxt["cin"] == 16 and nxt["expand"] != 1 and not nxt["skip"] and self.bufs[x].C <= 32
                      and mbx_supported(16, 16 * nxt["expand"], nxt["kernel"], nxt["stride"])
                      and (i not in red_ids or len([r for r in red_ids if r <= i]) < cfg["min_level"]))
            if absorb:
                # the next block's fused kernel computes this projection in its prologue: the 16-channel tensor is never stored
                pending_proj = (gate, proj, bn_names[nb])
                if i in red_ids:
                    reductions.append(-1)           # an endpoint below min_level: not an FPN input
                continue
            x = self._pw(x, b["cout"], proj, "blocks_%d/out" % i, bn=bn_names[nb], se=gate,
                         residual=inp if b["skip"] else -1)
            if i in red_ids:
                reductions.append(x)
        lo, hi = cfg["min_level"], cfg["max_level"]
        feats = reductions[lo - 1:]
        # ---- extra levels P6, P7 (efficientdet_keras.py:886-899, 1004-1005)
        F = cfg["fpn_num_filters"]
        for lvl in range(len(feats) + lo, hi + 1):
            src = self.bufs[feats[-1]]
            th, tw = (src.H + 1) // 2, (src.W + 1) // 2
            xr, mode = self._resample(feats[-1], th, tw, "resample_p%d" % lvl, "resample_p%d" % lvl)
            if mode == capi.RS_NONE:        # a 1x1 map cannot shrink further: the level repeats (keras :339-342); or pooled + 1x1 already
                feats.append(xr)
                self.buffer_names["p%d_in" % lvl] = xr
                continue
            out = self._buf(th, tw, F, self.bufs[xr].per_sample, name="p%d_in" % lvl)
            feats.append(self._op(capi.OP_POOL, [xr], out, resample=[capi.RS_MAXPOOL, 0, 0], fuse_w=[1.0, 0, 0]))
        # ---- BiFPN
        nodes, method = fpn_nodes(cfg)
        cba = bool(cfg.get("conv_bn_act_pattern", False))       # conv -> BN -> act instead of act -> conv(+bias) -> BN (:218,229-236)
        for rep in range(cfg["fpn_cell_repeats"]):
            cell = list(feats)
            for n, node in enumerate(nodes):
                p = "fpn_cells/cell_%d/fnode%d/" % (rep, n)
                nf = len(cell)
                tgt = self.bufs[cell[node["feat_level"] - lo]]
                ins, modes = [], []
                for i, off in enumerate(node["inputs_offsets"]):
                    xi, mode = self._resample(cell[off], tgt.H, tgt.W, p + "resample_%d_%d_%d" % (i, off, nf),
                                              "cell%d/fnode%d/in%d" % (rep, n, i))
                    ins.append(xi)
                    modes.append(mode)
                if method == "fastattn":
                    ew = [np.maximum(np.float32(np.asarray(w[p + "WSM" + ("" if i == 0 else "_%d" % i)]).reshape(())), np.float32(0))
                          for i in range(len(ins))]
                    tot = np.float32(0)
                    for e in ew:
                        tot = np.float32(tot + e)
                    fw = [float(np.float32(e / np.float32(tot + np.float32(0.0001)))) for e in ew]
                elif method == "attn":      # softmax of the scalar edge weights (:96-99), float32 like the reference's
                    ev = np.asarray([np.asarray(w[p + "WSM" + ("" if i == 0 else "_%d" % i)], np.float32).reshape(())
                                     for i in range(len(ins))], np.float32)
                    ex = np.exp(ev - ev.max(), dtype=np.float32)
                    fw = [float(v) for v in (ex / ex.sum(dtype=np.float32)).astype(np.float32)]
                elif method == "sum":
                    fw = [1.0] * len(ins)
                else:
                    raise ValueError("unknown weight_method %s" % method)
                ps = any(self.bufs[i].per_sample for i in ins)
                op = p + "op_after_combine%d" % nf
                fuse_act, conv_act = (capi.ACT_NONE, self.act) if cba else (self.act, capi.ACT_NONE)
                bias = None if cba else op + "/conv/bias"       # use_bias = not conv_bn_act_pattern (:218)
                if self.fuse_sep and sepf_supported(F, F):
                    # the fusion is computed inside the node's separable conv: no fused tensor, no fuse launch
                    cell.append(self._sepconv(None, F, op + "/conv/depthwise_kernel", op + "/conv/pointwise_kernel",
                                              "cell%d/fnode%d/out" % (rep, n), bias=bias, bn=op + "/bn", act=conv_act,
                                              fusion=dict(ins=ins, resample=modes, fuse_w=fw, H=tgt.H, W=tgt.W, act=fuse_act)))
                    continue
                fused = self._op(capi.OP_FUSE, ins, self._buf(tgt.H, tgt.W, F, ps, name="cell%d/fnode%d/fused" % (rep, n)),
                                 act=fuse_act, resample=(modes + [0, 0, 0])[:3], fuse_w=(fw + [0, 0, 0])[:3])
                cell.append(self._sepconv(fused, F, op + "/conv/depthwise_kernel", op + "/conv/pointwise_kernel",
                                          "cell%d/fnode%d/out" % (rep, n), bias=bias, bn=op + "/bn", act=conv_act))
            feats = []
            for lvl in range(lo, hi + 1):
                for i, node in enumerate(reversed(nodes)):
                    if node["feat_level"] == lvl:
                        feats.append(cell[-1 - i])
                        break
        self.fpn_out = list(feats)
        # ---- heads
        self.level_hw = [(self.bufs[f].H, self.bufs[f].W) for f in feats]
        A = len(cfg["aspect_ratios"]) * cfg["num_scales"]
        cls_ch = A * cfg["num_classes"]
        box_ch = A * (8 if cfg["loss_attenuation"] else 4)
        self.head_out = {"class": [], "box": []}
        # Layer-major: a head layer is emitted for all pyramid levels before the next layer (the reference loops level-major,
        # efficientdet_keras.py:470-486; the levels are independent, so the order is free).  The ops of one layer are
        # consecutive and of one shape class: the executor runs them as ONE launch (launch_group), which spares the
        # small levels their launch latency and lets them run in the shadow of the large ones.
        # Deferred head dropout (round 5): a head layer whose INPUT is still one row per image (head-only MC dropout: the whole
        # backbone + BiFPN is per image) does not apply its dropout site - its output stays per image, 1 / T of the bytes -
        # and hands the site to the layer that reads it: the keep-scale is per (sample, channel), a per-channel factor
        # commutes with that layer's depthwise conv, so its kernel computes the depthwise result once per image and serves the
        # T samples from it (drop_site2 of a SEP op; sep_kernel's TIN mode).  Same arithmetic up to the position of one multiply.
        for net, tag, outc, kind in (("class_net", "class", cls_ch, 2), ("box_net", "box", box_ch, 3)):
            xs = list(feats)
            pending = [-1] * len(feats)              # per level: the deferred site the next layer has to apply to its input
            nrep = cfg["box_class_repeats"]
            for i in range(nrep):
                pre = "%s/%s-%d" % (net, tag, i)
                first = len(self.ops)
                next_cout = F if i + 1 < nrep else outc
                for li in range(len(feats)):
                    site = self._site("%s-%d-%d" % (tag, i, lo + li))
                    # (T == 1 included - a rank of a sample-sharded serve may run ONE sample: the position of the multiply
                    # must not depend on how many samples a handle runs, or the shards would differ in the last bit)
                    defer = (site >= 0 and pending[li] < 0 and not self.bufs[xs[li]].per_sample
                             and self.fuse_sep and sep_tin_supported(F, next_cout))
                    xs[li] = self._sepconv(xs[li], F, pre + "/depthwise_kernel", pre + "/pointwise_kernel",
                                           "%s-%d-%d" % (tag, i, lo + li), bias=pre + "/bias",
                                           bn="%s/%s-%d-bn-%d" % (net, tag, i, lo + li), act=self.act,
                                           site=-1 if defer else site, in_site=pending[li])
                    pending[li] = site if defer else -1
                self._mark_launch_group(first)
            pre = "%s/%s-predict" % (net, tag)
            first = len(self.ops)
            for li in range(len(feats)):
                out = self._sepconv(xs[li], outc, pre + "/depthwise_kernel", pre + "/pointwise_kernel",
                                    "%s-predict-%d" % (tag, lo + li), bias=pre + "/bias", out_kind=kind, level=li,
                                    in_site=pending[li])
                self.head_out[tag].append(out)
            self._mark_launch_group(first)
        # the head buffers must carry the sample axis exactly when the reference stacks them
        for tag, stacked in (("class", self.cls_stacked), ("box", self.box_stacked)):
            for o in self.head_out[tag]:
                want = bool(stacked and self.sample_axis)
                if self.bufs[o].per_sample != want:
                    # stacked in the reference but every sample identical here (all rates zero):
                    # keep one copy; the driver broadcasts when it returns `predict` outputs.
                    assert not self.bufs[o].per_sample, "unexpected sample axis on head %s" % tag
        self.cls_stacked_dev = bool(self.bufs[self.head_out["class"][0]].per_sample)
        self.box_stacked_dev = bool(self.bufs[self.head_out["box"][0]].per_sample)

    # ------------------------------------------------------------------ memory planning
    def _rows(self, b):
        return self.chunk_images * (self.T if b.per_sample else 1)

    def _plan_memory(self):
        for oi, o in enumerate(self.ops):
            touched = list(o["ins"]) + [o["out"]] + [o[k] for k in ("se_scale", "se_partial", "residual") if o[k] >= 0]
            for b in touched:
                buf = self.bufs[b]
                if buf.first is None:
                    buf.first = oi
                buf.last = oi
        for f in self.fpn_out:
            self.bufs[f].last = len(self.ops)
        if self.cfg.get("uda_keep_buffers"):
            # parity aid (tests): no arena recycling - every named activation of the last chunk can be read back after the run
            # (ServingDriver.read_buffer) and compared with the oracle's taps; costs memory, changes no arithmetic
            for b in self.bufs:
                if b.first is not None:
                    b.last = len(self.ops)
        # ops that share a launch run concurrently: nothing they touch may be recycled before the last of them
        for oi, o in enumerate(self.ops):
            n = o.get("launch_group", 0)
            if n > 1:
                end = oi + n - 1
                for g in self.ops[oi:oi + n]:
                    for b in list(g["ins"]) + [g["out"]]:
                        buf = self.bufs[b]
                        buf.last = max(buf.last, end)
                        buf.first = min(buf.first, oi)
        free, top = [], 0  # free: list of (offset, size)

        def alloc(size):
            nonlocal top
            for i, (off, sz) in enumerate(free):
                if sz >= size:
                    if sz == size:
                        free.pop(i)
                    else:
                        free[i] = (off + size, sz - size)
                    return off
            off = top
            top += size
            return off

        def release(off, size):
            free.append((off, size))
            free.sort()
            merged = []
            for o_, s_ in free:
                if merged and merged[-1][0] + merged[-1][1] == o_:
                    merged[-1] = (merged[-1][0], merged[-1][1] + s_)
                else:
                    merged.append((o_, s_))
            free[:] = merged

        sizes = {}
        for oi in range(len(self.ops)):
            for bi, b in enumerate(self.bufs):
                if b.kind == 0 and b.first == oi:
                    size = -(-(self._rows(b) * b.H * b.W * b.C) // ALIGN) * ALIGN
                    sizes[bi] = size
                    b.offset = alloc(size)
            for bi, b in enumerate(self.bufs):
                if b.kind == 0 and b.last == oi and bi in sizes:
                    release(b.offset, sizes[bi])
        self.arena_floats = int(top)

    # ------------------------------------------------------------------ C structures
    def anchors(self):
        """[A_tot, 4] float32 (ymin, xmin, ymax, xmax); order level, y, x, (octave, aspect)
        (reference anchors.py:138-218)."""
        cfg = self.cfg
        lo, hi = cfg["min_level"], cfg["max_level"]
        H, W = parse_image_size(cfg["image_size"])
        fs = get_feat_sizes(cfg["image_size"], hi)
        scales = cfg["anchor_scale"]
        if not isinstance(scales, (list, tuple)):
            scales = [scales] * (hi - lo + 1)
        levels = []
        for lvl in range(lo, hi + 1):
            sy, sx = H / float(fs[lvl][0]), W / float(fs[lvl][1])
            ys = np.arange(sy / 2, H, sy)
            xs = np.arange(sx / 2, W, sx)
            yv, xv = np.meshgrid(ys, xs, indexing="ij")
            per = []
            for octave in range(cfg["num_scales"]):
                for aspect in cfg["aspect_ratios"]:
                    base_x = scales[lvl - lo] * sx * 2 ** (octave / float(cfg["num_scales"]))
                    base_y = scales[lvl - lo] * sy * 2 ** (octave / float(cfg["num_scales"]))
                    if isinstance(aspect, (list, tuple)):
                        ax, ay = aspect
                    else:
                        ax = np.sqrt(aspect)
                        ay = 1.0 / ax
                    hx, hy = base_x * ax / 2.0, base_y * ay / 2.0
                    per.append(np.stack([yv - hy, xv - hx, yv + hy, xv + hx], axis=-1))
            levels.append(np.stack(per, axis=2).reshape(-1, 4))
        return np.concatenate(levels).astype(np.float32)

    def to_c(self, post_mode=capi.POST_GLOBAL):
        cfg = self.cfg
        H, W = parse_image_size(cfg["image_size"])
        m = capi.Model()
        m.abi_version = capi.UDA_ABI_VERSION
        m.image_h, m.image_w = H, W
        mean, std = cfg["mean_rgb"], cfg["stddev_rgb"]
        if not isinstance(mean, (list, tuple)):
            mean, std = [mean] * 3, [std] * 3
        for i in range(3):
            m.mean_rgb[i] = np.float32(mean[i])
            m.stddev_rgb[i] = np.float32(std[i])
        m.num_levels = len(self.level_hw)
        for i, (h, w_) in enumerate(self.level_hw):
            m.level_h[i], m.level_w[i] = h, w_
        m.anchors_per_loc = len(cfg["aspect_ratios"]) * cfg["num_scales"]
        m.num_classes = cfg["num_classes"]
        m.loss_attenuation = int(bool(cfg["loss_attenuation"]))
        m.mc_samples = self.T
        m.cls_stacked = int(self.cls_stacked_dev)
        m.box_stacked = int(self.box_stacked_dev)
        m.has_uncert = int(bool(cfg["loss_attenuation"] or cfg["mc_dropout"]))
        method = cfg["uncert_adjust_method"]
        if not cfg["loss_attenuation"]:
            m.decode_method = capi.DECODE_PLAIN
        elif method in ("l-norm", "n-flow"):     # n-flow is analytically the l-norm closed form
            m.decode_method = capi.DECODE_LNORM
        elif method == "falsedec":
            m.decode_method = capi.DECODE_FALSEDEC
        elif method == "sample":
            m.decode_method = capi.DECODE_SAMPLE
            m.decode_nsamples = int(cfg.get("decode_nsamples", 100))
            if not 1 <= m.decode_nsamples <= 4096:
                raise ValueError("decode_nsamples must be in [1, 4096]")
        else:
            raise ValueError("uncert_adjust_method %r is not available on the HIP path" % method)
        m.enable_softmax = int(bool(cfg["enable_softmax"]))
        sigma2, iou, thr = nms_params(cfg)
        m.nms_soft_sigma, m.nms_iou_thresh, m.nms_score_thresh = sigma2, iou, thr
        m.max_output_size = int(cfg["nms_configs"]["max_output_size"])
        m.max_nms_inputs = int(cfg["nms_configs"].get("max_nms_inputs", 0) or 0)
        m.post_mode = post_mode
        m.chunk_images, m.max_images = self.chunk_images, self.max_images
        m.arena_floats = self.arena_floats
        m.n_drop_sites = len(self.sites)

        bufs = (capi.BufDesc * len(self.bufs))()
        for i, b in enumerate(self.bufs):
            bufs[i].H, bufs[i].W, bufs[i].C = b.H, b.W, b.C
            bufs[i].per_sample = int(b.per_sample)
            bufs[i].offset, bufs[i].kind, bufs[i].level = int(b.offset), b.kind, b.level
        ops = (capi.Op * max(1, len(self.ops)))()
        for i, o in enumerate(self.ops):
            c = ops[i]
            c.kind = o["kind"]
            ins = (o["ins"] + [-1, -1, -1])[:3]
            for j in range(3):
                c.in_[j] = ins[j]
                c.resample[j] = o["resample"][j]
                c.fuse_w[j] = o["fuse_w"][j]
            c.n_in = len(o["ins"])
            for k in ("out", "se_scale", "se_partial", "residual", "k", "stride", "act", "w_off", "bias_off",
                      "bn_scale_off", "bn_shift_off", "se_w1_off", "se_b1_off", "se_w2_off", "se_b2_off",
                      "se_mid", "drop_site", "drop_site2", "w2_off", "bn2_scale_off", "bn2_shift_off", "launch_group"):
                setattr(c, k, int(o[k]))
            c.fuse_in = int(o.get("fuse_in", 0))
            c.fuse_act = int(o.get("fuse_act", 0))
        sites = (capi.DropSite * max(1, len(self.sites)))()
        for i, (_, ch, r) in enumerate(self.sites):
            sites[i].channels, sites[i].rate = ch, np.float32(r)
        blob = np.concatenate(self.blob) if self.blob else np.zeros(4, np.float32)
        return m, bufs, ops, sites, blob, self.anchors()

    def summary(self):
        return dict(n_ops=len(self.ops), n_bufs=len(self.bufs), arena_mb=self.arena_floats * 4 / 2 ** 20,
                    weights_mb=self.blob_len * 4 / 2 ** 20, sites=len(self.sites), T=self.T)


def nms_params(cfg):
    """(sigma/2, iou_thresh, score_thresh) exactly as postprocess.nms derives them (:373-398)."""
    nc = cfg["nms_configs"]
    method = nc["method"]
    if method == "hard" or not method:
        return 0.0, float(nc["iou_thresh"] or 0.5), float(nc["score_thresh"] or float("-inf"))
    if method == "gaussian":
        return float(nc["sigma"] or 0.5) / 2, 0.5, float(nc["score_thresh"] or 0.001)
    raise ValueError("Inference has invalid nms method {}".format(method))


def op_costs(plan, n_images):
    """Layer-wise ALGORITHMIC cost of one pass of the op list over `n_images` images:
    {op kind: dict(bytes, flops, launches)}.  Per op: every input tensor is read once, the
    weights once, the output written once (BN / bias / activation / dropout folded into the
    producing op) — the accounting SURVEY §8d uses for its MB/W figures."""
    import math
    T = plan.T
    chunks = math.ceil(n_images / plan.chunk_images)
    out = {}

    def size(bi):
        b = plan.bufs[bi]
        return n_images * (T if b.per_sample else 1) * b.H * b.W * b.C

    for o in plan.ops:
        ob = plan.bufs[o["out"]]
        rows = n_images * (T if ob.per_sample else 1)
        by = size(o["out"])
        fl = 0
        for bi in (o["ins"][:1] if o["kind"] == capi.OP_SE else o["ins"]):
            by += size(bi)          # SE reads only the tile sums the depthwise kernel left
        for key in ("se_scale", "residual", "se_partial"):
            if o[key] >= 0 and o["kind"] != capi.OP_SE:
                by += size(o[key])
        k = o["kind"]
        if k == capi.OP_PW:
            cin = plan.bufs[o["ins"][0]].C
            by += cin * ob.C
            fl = 2 * rows * ob.H * ob.W * cin * ob.C
        elif k == capi.OP_DW:
            by += o["k"] * o["k"] * ob.C
            fl = 2 * rows * ob.H * ob.W * ob.C * o["k"] * o["k"]
        elif k == capi.OP_STEM:
            by += 27 * ob.C
            fl = 2 * rows * ob.H * ob.W * 27 * ob.C
        elif k == capi.OP_MBX:
            ib = plan.bufs[o["ins"][0]]
            cin = o["se_mid"] if o["se_scale"] >= 0 else ib.C
            by += cin * ob.C + o["k"] * o["k"] * ob.C
            fl = 2 * rows * (ib.H * ib.W * cin * ob.C + ob.H * ob.W * ob.C * o["k"] * o["k"])
            if o["se_scale"] >= 0:      # absorbed projection of the previous block
                by += ib.C * cin
                fl += 2 * rows * ib.H * ib.W * ib.C * cin
        elif k == capi.OP_SEP:
            ib = plan.bufs[o["ins"][0]]
            by += ib.C * ob.C + 9 * ib.C
            fl = 2 * rows * ob.H * ob.W * (9 * ib.C + ib.C * ob.C)
            if o.get("fuse_in"):
                fl += rows * ob.H * ob.W * ib.C * len(o["ins"]) * 2
        elif k == capi.OP_SE:
            fl = 4 * rows * ob.C * o["se_mid"]
        elif k in (capi.OP_FUSE, capi.OP_POOL):
            fl = rows * ob.H * ob.W * ob.C * len(o["ins"]) * 2
        d = out.setdefault(k, dict(bytes=0, flops=0, launches=0))
        d["bytes"] += 4 * by
        d["flops"] += fl
        d["launches"] += chunks
    return out
