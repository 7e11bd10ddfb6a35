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


class Config:
    """A dict-like config whose members are reachable as attributes and items."""

    def __init__(self, config_dict=None):
        if config_dict:
            self.update(config_dict)

    # attribute / item plumbing -------------------------------------------
    def __setattr__(self, k, v):
        self.__dict__[k] = Config(v) if isinstance(v, dict) else copy.deepcopy(v)

    def __getattr__(self, k):
        try:
            return self.__dict__[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __getitem__(self, k):
        return self.__dict__[k]

    def __setitem__(self, k, v):
        self.__setattr__(k, v)

    def __contains__(self, k):
        return k in self.__dict__

    def __repr__(self):
        return repr(self.as_dict())

    def __deepcopy__(self, memo):
        return Config(self.as_dict())

    def get(self, k, default=None):
        return self.__dict__.get(k, default)

    def keys(self):
        return self.__dict__.keys()

    # update / override ------------------------------------------------------
    def _update(self, config_dict, allow_new_keys):
        if not config_dict:
            return
        for k, v in config_dict.items():
            if k not in self.__dict__:
                if not allow_new_keys:
                    raise KeyError("Key `{}` does not exist for overriding.".format(k))
                self.__setattr__(k, v)
            elif isinstance(self.__dict__[k], Config) and isinstance(v, dict):
                self.__dict__[k]._update(v, allow_new_keys)
            elif isinstance(self.__dict__[k], Config) and isinstance(v, Config):
                self.__dict__[k]._update(v.as_dict(), allow_new_keys)
            else:
                self.__setattr__(k, v)

    def update(self, config_dict):
        """Update members, new keys allowed (reference `Config.update`)."""
        self._update(config_dict, allow_new_keys=True)

    def override(self, config_dict_or_str, allow_new_keys=False):
        """Update members from a dict, a yaml path or a `k=v,...` string."""
        if isinstance(config_dict_or_str, str):
            if not config_dict_or_str:
                return
            if "=" in config_dict_or_str:
                config_dict = self.parse_from_str(config_dict_or_str)
            elif config_dict_or_str.endswith(".yaml"):
                config_dict = self.parse_from_yaml(config_dict_or_str)
            else:
                raise ValueError(
                    'Invalid string {}, must end with .yaml or contains "=".'.format(
                        config_dict_or_str))
        elif isinstance(config_dict_or_str, dict):
            config_dict = config_dict_or_str
        elif isinstance(config_dict_or_str, Config):
            config_dict = config_dict_or_str.as_dict()
        else:
            raise ValueError("Unknown value type: {}".format(config_dict_or_str))
        self._update(config_dict, allow_new_keys)

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
        if not config_str:
            return out
        try:
            for kv in config_str.split(","):
                if not kv:
                    continue
                key, val = kv.split("=")
                key = key.strip()
                if "*" in val:
                    leaf = [_eval_str(v) for v in val.split("*")]
                else:
                    leaf = _eval_str(val)
                node = out
                parts = key.split(".")
                for p in parts[:-1]:
                    node = node.setdefault(p, {})
                    if not isinstance(node, dict):
                        raise ValueError(key)
                node[parts[-1]] = leaf
        except ValueError:
            raise ValueError("Invalid config_str: {}".format(config_str))
        return out

    def as_dict(self):
        d = {}
        for k, v in self.__dict__.items():
            d[k] = v.as_dict() if isinstance(v, Config) else copy.deepcopy(v)
        return d


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
