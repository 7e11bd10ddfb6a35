"""`EfficientDetNet` / `EfficientDetModel`-shaped callables over the HIP handle.

The reference's callers outside `infer_lib` build the Keras model themselves and call it
(src/eval.py:98-123, src/train_lib.py:368-380):

    model = efficientdet_keras.EfficientDetNet(config=config)
    model.build((None, *config.image_size, 3))
    utils_keras.restore_ckpt(model, ckpt, config.moving_average_decay, skip_mismatch=False)
    cls_outputs, box_outputs = model(images, training=False)          # or mc_eval(model, images, config)
    detections = postprocess.generate_detections(config, cls_outputs, box_outputs, image_scales, source_ids)

These classes give that call shape (constructor arguments, `build`, `__call__`, output structure of
`EfficientDetNet.call`, src/efficientdet_keras.py:979-1070, and of `EfficientDetModel.call`, :1118-1146)
on top of `infer_lib.ServingDriver`: the arithmetic runs in the HIP library; the returned head outputs are
`infer_lib.DeviceHeads` (list-like, downloaded only when indexed) so that `generate_detections`
post-processes them where they are.  No Keras layers, variables or training here (out of scope).
"""
import numpy as np

from . import hparams_config
from .infer_lib import ServingDriver


def _as_config(model_name, config, params):
    if config is None:
        config = hparams_config.get_efficientdet_config(model_name or "efficientdet-d0")
    elif isinstance(config, dict):
        config = hparams_config.Config(config)
    else:
        config = hparams_config.Config(config.as_dict())
    if params:
        config.override(params)
    return config


class EfficientDetNet:
    """`EfficientDetNet(model_name=None, config=None, params=None, name="")` (efficientdet_keras.py:850-870).

    `model(images, training=False)`: float32 [N,H,W,3] network inputs -> (cls_outputs, box_outputs), five levels
    each.  As in the reference's `call`: with `config.mc_dropout and not config.is_training_bn` the T =
    `mc_dropoutsamp` stochastic passes run inside the call and the heads with a non-zero rate come back stacked
    [T,N,h,w,ch]; otherwise one pass ([N,h,w,ch]; MC dropout layers still draw a mask when `mc_dropout` is set,
    since the reference builds them with training=True)."""

    def __init__(self, model_name=None, config=None, params=None, name="", *, device=0, weights=None, weights_path=None):
        self.config = _as_config(model_name, config, params)
        self.name = name
        self._device = device
        self._weights = weights
        self._weights_path = weights_path
        self._drivers = {}        # (T, capacity) -> ServingDriver
        self._batch = None

    # ------------------------------------------------------------------ Keras-shaped plumbing
    def build(self, input_shape):
        h, w = hparams_config.parse_image_size(self.config.image_size)
        if tuple(input_shape[1:]) != (h, w, 3):
            raise ValueError("input shape %s does not match image_size %s" % (tuple(input_shape), self.config.image_size))
        self._batch = input_shape[0]

    def load_weights(self, path_or_weights):
        """Weight set (dict of reference-named arrays), `.npz`, or TF2 checkpoint prefix (weights.resolve_weights)."""
        for d in self._drivers.values():
            d.close()
        self._drivers = {}
        if isinstance(path_or_weights, dict):
            self._weights, self._weights_path = path_or_weights, None
        else:
            self._weights, self._weights_path = None, path_or_weights

    def close(self):
        for d in self._drivers.values():
            d.close()
        self._drivers = {}

    def _driver(self, n, mc):
        cfg = self.config.as_dict()
        T = int(cfg["mc_dropoutsamp"]) if mc else 1
        cap = max(int(n), int(self._batch or 0), 1)
        for (t, c), d in self._drivers.items():
            if t == T and c >= n:
                return d
        params = dict(cfg, mc_dropoutsamp=T)
        name = params.get("name") or "efficientdet-d0"
        d = ServingDriver(name, cap, True, params, device=self._device, weights=self._weights,
                          weights_path=self._weights_path)
        if self._weights is None:
            self._weights = d.weights          # later handles share the set the first one resolved
        self._drivers[(T, cap)] = d
        return d

    def _run(self, inputs, mc):
        x = np.ascontiguousarray(inputs, dtype=np.float32)
        if x.ndim != 4:
            raise ValueError("inputs must be [batch, height, width, 3], got %s" % (x.shape,))
        d = self._driver(x.shape[0], mc)
        d.predict_resident(x)
        return d, x.shape[0]

    def __call__(self, inputs, training=False):
        if training:
            raise ValueError("the HIP path serves inference only (training is out of scope)")
        cfg = self.config
        mc = bool(cfg.mc_dropout and not cfg.is_training_bn)
        d, n = self._run(inputs, mc)
        cls, box = d.device_heads(n)
        if not mc:                      # one pass: drop the (length-1) sample axis of heads the plan stacked
            cls, box = _Squeezed(cls), _Squeezed(box)
        return cls, box

    call = __call__

    def mc_forward(self, inputs):
        """All `mc_dropoutsamp` stochastic passes in one run (what `utils_extra.mc_eval` loops over)."""
        d, n = self._run(inputs, True)
        return list(d.device_heads(n))


class _Squeezed:
    """DeviceHeads of a T = 1 run presented without the sample axis."""

    def __init__(self, heads):
        self.heads = heads
        self.driver, self.n, self.which, self.run_id = heads.driver, heads.n, heads.which, heads.run_id

    def __len__(self):
        return len(self.heads)

    def __getitem__(self, i):
        a = self.heads[i]
        return a[0] if a.ndim == 5 else a

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class EfficientDetModel(EfficientDetNet):
    """EfficientDet with pre- and post-processing (efficientdet_keras.py:1073-1146):
    `model(uint8_images, training=False, pre_mode="infer", post_mode="global")` -> the post-process tuple."""

    def _serving(self, n):
        cfg = self.config.as_dict()
        cap = max(int(n), int(self._batch or 0), 1)
        for (t, c), d in self._drivers.items():
            if t == "serve" and c >= n:
                return d
        d = ServingDriver(cfg.get("name") or "efficientdet-d0", cap, False, dict(cfg, is_training_bn=False),
                          device=self._device, weights=self._weights, weights_path=self._weights_path)
        if self._weights is None:
            self._weights = d.weights
        self._drivers[("serve", cap)] = d
        return d

    def __call__(self, inputs, training=False, pre_mode="infer", post_mode="global"):
        if training:
            raise ValueError("the HIP path serves inference only (training is out of scope)")
        if pre_mode not in (None, "", "infer"):
            raise ValueError("preprocessing must be infer or empty")
        if post_mode not in (None, "", "global", "per_class"):
            raise ValueError("Unsupported postprocess mode {}".format(post_mode))
        if not pre_mode:
            if post_mode:
                x = np.ascontiguousarray(inputs, dtype=np.float32)
                d = self._serving(x.shape[0])
                d.predict_resident(x)
                cls, box = d.device_heads(x.shape[0])
                return d.postprocess(cls, box, None, post_mode=post_mode)
            return EfficientDetNet.__call__(self, inputs, training)
        a = np.asarray(inputs)
        d = self._serving(a.shape[0] if a.ndim == 4 else 1)
        if post_mode:
            return d.serve(a, post_mode=post_mode)
        d.serve(a)                              # pre-process + network; the detections are discarded
        return d.device_heads(d._last_n)

    call = __call__
