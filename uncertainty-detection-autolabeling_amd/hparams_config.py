"""Detection config for the hot path: same keys, defaults and override rules as the reference.

Mirrors (behaviour, not code) ``src/hparams_config.py`` of the reference:
  * ``Config`` with attribute + item access and recursive ``override`` from a
    dict, a ``.yaml`` path or an ``"a.b=1,c=2*3"`` string       (hparams_config.py:38-178)
  * ``default_detection_configs`` – only the keys the inference path reads
    (uncertainty switches :193-213, anchors :277-282, preprocessing :298-299,
    architecture :325-340)                                       (hparams_config.py:183-370)
  * model table ``efficientdet-d0 … d7x``                        (hparams_config.py:373-452)

Training-only keys (optimizer, lr schedule, losses, augmentation …) are kept
with their reference defaults where a YAML from ``configs/train`` sets them, so
those YAML files load unchanged; nothing on this path reads them.
"""
import ast
import copy

import yaml


def _eval_str(val):
    if val in ("true", "True"):
        return True
    if val in ("false", "False"):
        return False
    try:
        return ast.literal_eval(val)
    except (ValueError, SyntaxError):
        return val


def _wrap(value):
    """Nested dicts become Config nodes; everything else is stored by value."""
    return Config(value) if isinstance(value, dict) else copy.deepcopy(value)


def _plain(value):
    return value.as_dict() if isinstance(value, Config) else copy.deepcopy(value)


def _merge(node, updates, allow_new_keys):
    """Recursive update of a Config node from a plain mapping: existing sub-configs are merged key by key, leaves are
    replaced, unknown keys raise unless `allow_new_keys` (the override rules of the reference's Config)."""
    for key, value in (updates or {}).items():
        store = node._store
        if key not in store:
            if not allow_new_keys:
                raise KeyError("Key `{}` does not exist for overriding.".format(key))
            store[key] = _wrap(value)
            continue
        current = store[key]
        if isinstance(current, Config) and isinstance(value, (dict, Config)):
            _merge(current, _plain(value), allow_new_keys)
        else:
            store[key] = _wrap(value)


class Config:
    """A config tree whose members are reachable as attributes and items.

    Values live in one ordered mapping per node (`_store`); attribute and item access are two views of it.  The public
    behaviour - `update` (new keys allowed), `override` from a dict / another Config / a ``.yaml`` path / an
    ``"a.b=1,c=2*3"`` string (unknown keys refused unless asked), `as_dict`, `get`, `keys` - is the reference's."""

    __slots__ = ("_store",)

    def __init__(self, config_dict=None):
        object.__setattr__(self, "_store", {})
        if config_dict:
            self.update(config_dict)

    # attribute / item views ------------------------------------------------------
    def __getattr__(self, name):
        try:
            return object.__getattribute__(self, "_store")[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        self._store[name] = _wrap(value)

    __setitem__ = __setattr__

    def __getitem__(self, name):
        return self._store[name]

    def __contains__(self, name):
        return name in self._store

    def __iter__(self):
        return iter(self._store)

    def __len__(self):
        return len(self._store)

    def __repr__(self):
        return repr(self.as_dict())

    def __deepcopy__(self, memo):
        return Config(self.as_dict())

    def __getstate__(self):
        return self.as_dict()

    def __setstate__(self, state):
        object.__setattr__(self, "_store", {})
        self.update(state)

    def get(self, name, default=None):
        return self._store.get(name, default)

    def keys(self):
        return self._store.keys()

    def items(self):
        return self._store.items()

    # update / override ----------------------------------------------------------
    def update(self, config_dict):
        """Merge a mapping; keys that do not exist yet are added."""
        _merge(self, _plain(config_dict) if isinstance(config_dict, Config) else config_dict, True)

    def override(self, config_dict_or_str, allow_new_keys=False):
        """Merge a dict, a Config, a yaml file or a `k=v,...` string; unknown keys are an error by default."""
        src = config_dict_or_str
        if isinstance(src, Config):
            src = src.as_dict()
        elif isinstance(src, str):
            if not src:
                return
            if "=" in src:
                src = self.parse_from_str(src)
            elif src.endswith(".yaml"):
                src = self.parse_from_yaml(src)
            else:
                raise ValueError('Invalid string {}, must end with .yaml or contains "=".'.format(src))
        elif not isinstance(src, dict):
            raise ValueError("Unknown value type: {}".format(src))
        _merge(self, src, allow_new_keys)

    @staticmethod
    def parse_from_yaml(path):
        with open(path, "r") as f:
            return yaml.load(f, Loader=yaml.FullLoader)

    def save_to_yaml(self, path):
        with open(path, "w") as f:
            yaml.dump(self.as_dict(), f, default_flow_style=False)

    @staticmethod
    def parse_from_str(config_str):
        """'x.y=1,x.z=2,a=3*4' -> {x: {y: 1, z: 2}, a: [3, 4]}."""
        out = {}
        for assignment in filter(None, (config_str or "").split(",")):
            try:
                dotted, raw = assignment.split("=")
            except ValueError:
                raise ValueError("Invalid config_str: {}".format(config_str)) from None
            leaf = [_eval_str(v) for v in raw.split("*")] if "*" in raw else _eval_str(raw)
            *parents, last = dotted.strip().split(".")
            node = out
            for name in parents:
                node = node.setdefault(name, {})
                if not isinstance(node, dict):
                    raise ValueError("Invalid config_str: {}".format(config_str))
            node[last] = leaf
        return out

    def as_dict(self):
        return {k: _plain(v) for k, v in self._store.items()}


def default_detection_configs():
    """Defaults (reference hparams_config.py:183-370)."""
    h = Config()

    # --- uncertainty switches (hparams_config.py:193-213) ---
    h.loss_attenuation = False
    h.clip_min_uncert = 0.01
    h.clip_max_uncert = 1024
    h.uncert_adjust_method = "l-norm"   # l-norm | n-flow | falsedec | sample
    h.decode_nsamples = 100
    h.mc_dropout = False
    h.mc_dropoutrate = 0.0
    h.mc_classheadrate = 0.0
    h.mc_boxheadrate = 0.0
    h.mc_dropoutsamp = 10
    h.assign_gt_box = "IoU"
    h.enable_softmax = False
    h.calibrate_classification = True
    h.calib_method_class = "iso_percls"
    h.calibrate_regression = True
    h.calib_method_box = "iso_perclscoo"
    h.infer_draw_uncert = True
    h.early_stopping_patience = 0

    # --- training-only keys that the shipped YAMLs set (kept so they load) ---
    h.count_classes = False
    h.boxloss_type = "huber"
    h.save_freq = 1
    h.sample_images = None
    h.sample_images_freq = None
    h.save_train_images = False
    h.autoaugment_policy = None
    h.map_freq = 5
    h.box_loss_weight = 50.0
    h.moving_average_decay = 0.9998
    h.mixed_precision = False
    h.label_map = None
    h.max_instances_per_image = 100
    h.strategy = None

    # --- model / preprocessing ---
    h.name = "efficientdet-d1"
    h.act_type = "swish"
    h.image_size = 640                      # int, "WxH" string or (H, W)
    h.num_classes = 90
    h.heads = ["object_detection"]
    h.min_level = 3
    h.max_level = 7
    h.num_scales = 3
    h.aspect_ratios = [1.0, 2.0, 0.5]
    h.anchor_scale = 4.0
    h.is_training_bn = True
    h.data_format = "channels_last"
    h.mean_rgb = [0.485 * 255, 0.456 * 255, 0.406 * 255]
    h.stddev_rgb = [0.229 * 255, 0.224 * 255, 0.225 * 255]

    h.box_class_repeats = 3
    h.fpn_cell_repeats = 3
    h.fpn_num_filters = 88
    h.separable_conv = True
    h.apply_bn_for_resampling = True
    h.conv_after_downsample = False
    h.conv_bn_act_pattern = False

    h.nms_configs = {
        "method": "gaussian",
        "iou_thresh": None,
        "score_thresh": 0.0,
        "sigma": None,
        "pyfunc": False,
        "max_nms_inputs": 0,
        "max_output_size": 100,
    }
    h.tflite_max_detections = 100

    h.fpn_name = None
    h.fpn_weight_method = None
    h.fpn_config = None
    h.survival_prob = None
    h.backbone_name = "efficientnet-b1"
    h.backbone_config = None
    h.grad_checkpoint = False
    return h


# (name, backbone, image_size, fpn filters, fpn cells, head repeats)  hparams_config.py:373-452
efficientdet_model_param_dict = {
    "efficientdet-d0": dict(name="efficientdet-d0", backbone_name="efficientnet-b0",
                            image_size=512, fpn_num_filters=64, fpn_cell_repeats=3,
                            box_class_repeats=3),
    "efficientdet-d1": dict(name="efficientdet-d1", backbone_name="efficientnet-b1",
                            image_size=640, fpn_num_filters=88, fpn_cell_repeats=4,
                            box_class_repeats=3),
    "efficientdet-d2": dict(name="efficientdet-d2", backbone_name="efficientnet-b2",
                            image_size=768, fpn_num_filters=112, fpn_cell_repeats=5,
                            box_class_repeats=3),
    "efficientdet-d3": dict(name="efficientdet-d3", backbone_name="efficientnet-b3",
                            image_size=896, fpn_num_filters=160, fpn_cell_repeats=6,
                            box_class_repeats=4),
    "efficientdet-d4": dict(name="efficientdet-d4", backbone_name="efficientnet-b4",
                            image_size=1024, fpn_num_filters=224, fpn_cell_repeats=7,
                            box_class_repeats=4),
    "efficientdet-d5": dict(name="efficientdet-d5", backbone_name="efficientnet-b5",
                            image_size=1280, fpn_num_filters=288, fpn_cell_repeats=7,
                            box_class_repeats=4),
    "efficientdet-d6": dict(name="efficientdet-d6", backbone_name="efficientnet-b6",
                            image_size=1280, fpn_num_filters=384, fpn_cell_repeats=8,
                            box_class_repeats=5, fpn_weight_method="sum"),
    "efficientdet-d7": dict(name="efficientdet-d7", backbone_name="efficientnet-b6",
                            image_size=1536, fpn_num_filters=384, fpn_cell_repeats=8,
                            box_class_repeats=5, anchor_scale=5.0, fpn_weight_method="sum"),
    "efficientdet-d7x": dict(name="efficientdet-d7x", backbone_name="efficientnet-b7",
                             image_size=1536, fpn_num_filters=384, fpn_cell_repeats=8,
                             box_class_repeats=5, anchor_scale=4.0, max_level=8,
                             fpn_weight_method="sum"),
}


def get_efficientdet_config(model_name="efficientdet-d1"):
    h = default_detection_configs()
    if model_name in efficientdet_model_param_dict:
        h.override(efficientdet_model_param_dict[model_name])
    else:
        raise ValueError("Unknown model name: {}".format(model_name))
    return h


def get_detection_config(model_name):
    if model_name.startswith("efficientdet"):
        return get_efficientdet_config(model_name)
    raise ValueError("model name must start with efficientdet.")


def parse_image_size(image_size):
    """int | "WxH" | (H, W)  ->  (H, W)        (reference utils.py:516-540)."""
    if isinstance(image_size, int):
        return (image_size, image_size)
    if isinstance(image_size, str):
        width, height = image_size.lower().split("x")
        return (int(height), int(width))
    if isinstance(image_size, (tuple, list)):
        return (int(image_size[0]), int(image_size[1]))
    raise ValueError("image_size must be an int, WxH string, or (height, width)"
                     "tuple. Was %r" % (image_size,))


def get_feat_sizes(image_size, max_level):
    """[(H, W)] for levels 0..max_level, each level ceil-halved (utils.py:543-559)."""
    h, w = parse_image_size(image_size)
    sizes = [(h, w)]
    for _ in range(max_level):
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        sizes.append((h, w))
    return sizes
