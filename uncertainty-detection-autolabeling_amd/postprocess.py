"""The reference's `postprocess` entry points on the HIP path (src/postprocess.py:472-887), same argument order.

  postprocess_global(params, cls_outputs, box_outputs, image_scales=None)      :472-621   (rows a8-a16)
  postprocess_per_class(params, cls_outputs, box_outputs, image_scales=None)   :719-740   (row a17)
  generate_detections(params, cls_outputs, box_outputs, image_scales, image_ids, flip=False, per_class_nms=True)  :788-871
  generate_detections_from_nms_output / transform_detections                   :743-785, :874-887   (row a19)

The arithmetic (MC aggregation, decode, NMS) runs on the GPU in a `ServingDriver` handle.  Head outputs that are
`infer_lib.DeviceHeads` of a driver's last run (what `efficientdet_keras.EfficientDetNet` / `utils_extra.mc_eval`
of this package return) are post-processed where they are; plain arrays are uploaded into a post-process-only
handle that is created on first use and cached per configuration.  The legacy row formats are pure re-packing
of <= 100 rows per image, which the reference does with tf.stack on the host side of the path.
"""
import collections
import json

import numpy as np

from .hparams_config import parse_image_size

CLASS_OFFSET = 1
# post-process-only handles for head outputs that arrive as arrays: keyed by what SHAPES a handle (geometry, sample axis,
# classes, decode / NMS settings), least recently used first, at most _POST_CACHE_MAX alive (each holds head buffers of
# n x T x all levels plus candidate and NMS workspaces - hundreds of MB at T = 30 or D2)
_POST_DRIVERS = collections.OrderedDict()
_POST_CACHE_MAX = 4
_SHAPING_KEYS = ("name", "image_size", "min_level", "max_level", "num_scales", "aspect_ratios", "anchor_scale", "num_classes",
                 "mc_dropout", "mc_dropoutsamp", "mc_dropoutrate", "mc_classheadrate", "mc_boxheadrate", "loss_attenuation",
                 "enable_softmax", "uncert_adjust_method", "decode_nsamples", "nms_configs")


def _params_dict(params):
    return params.as_dict() if hasattr(params, "as_dict") else dict(params)


def _cache_key(p):
    from . import arch
    shaped = {k: p.get(k) for k in _SHAPING_KEYS}
    # the dropout RATES do not shape a post-process handle, only which heads carry the sample axis does
    cls_st, box_st, T = arch.mc_flags(p)
    for k in ("mc_dropoutrate", "mc_classheadrate", "mc_boxheadrate", "mc_dropoutsamp"):
        shaped.pop(k)
    shaped["stacking"] = (bool(cls_st), bool(box_st), int(T))
    return json.dumps(shaped, sort_keys=True, default=str)


def close_cached():
    """Close every cached post-process handle (frees their device buffers)."""
    while _POST_DRIVERS:
        _, d = _POST_DRIVERS.popitem(last=False)
        d.close()


def _post_driver(params, n):
    """A post-process-only handle for `params` holding at least n images (LRU cache of _POST_CACHE_MAX handles)."""
    from .infer_lib import ServingDriver
    p = _params_dict(params)
    key = _cache_key(p)
    have = _POST_DRIVERS.get(key)
    if have is not None and have._cap >= n:
        _POST_DRIVERS.move_to_end(key)
        return have
    if have is not None:
        del _POST_DRIVERS[key]
        have.close()
    while len(_POST_DRIVERS) >= _POST_CACHE_MAX:
        _, old = _POST_DRIVERS.popitem(last=False)
        old.close()
    d = ServingDriver(p.get("name") or "efficientdet-d0", max(int(n), 1), False, p, post_only=True, chunk_images=1)
    _POST_DRIVERS[key] = d
    return d


def _resident_driver(cls_outputs, box_outputs):
    d = getattr(cls_outputs, "driver", None)
    if d is not None and getattr(box_outputs, "driver", None) is d and cls_outputs.run_id == d._run_id == box_outputs.run_id:
        return d
    return None


def _run(params, cls_outputs, box_outputs, image_scales, mode, driver=None):
    d = driver or _resident_driver(cls_outputs, box_outputs)
    if d is not None and driver is None:       # resident heads: hand over the DeviceHeads (their T = 1 view included)
        c = getattr(cls_outputs, "heads", cls_outputs)
        b = getattr(box_outputs, "heads", box_outputs)
        return d.postprocess(c, b, image_scales, post_mode=mode)
    if d is None:
        n = np.shape(cls_outputs[0])[-4]
        d = _post_driver(params, n)
    return d.postprocess(cls_outputs, box_outputs, image_scales, post_mode=mode)


def postprocess_global(params, cls_outputs, box_outputs, image_scales=None):
    """(boxes [N,M,4(+4)(+4)], scores, classes [N,M(,1+C)], valid_len[, logits]) - src/postprocess.py:472-621."""
    return _run(params, cls_outputs, box_outputs, image_scales, "global")


def postprocess_per_class(params, cls_outputs, box_outputs, image_scales=None):
    """(boxes [N,M,4], scores, classes, valid_len) - src/postprocess.py:719-740 (the reference's logits output of this
    mode is corrupted by a variable overwrite, :659-666, and is not produced)."""
    return _run(params, cls_outputs, box_outputs, image_scales, "per_class")


def generate_detections_from_nms_output(nms_boxes_bs, nms_classes_bs, nms_scores_bs, image_ids,
                                        original_image_widths=None, flip=False, nms_multi_class_bs=None):
    """[id, x, y, x2, y2, score, class(, logits...)] rows, float32 [N, M, 7(+C)] (postprocess.py:743-785)."""
    boxes = np.asarray(nms_boxes_bs, dtype=np.float32)
    scores = np.asarray(nms_scores_bs, dtype=np.float32)
    classes = np.asarray(nms_classes_bs, dtype=np.float32)
    ids = np.asarray(image_ids, dtype=np.float32)[:, None] * np.ones_like(scores)
    if flip:
        w = np.asarray(original_image_widths, dtype=np.float32)
        cols = [ids, w - boxes[:, :, 3], boxes[:, :, 0], w - boxes[:, :, 1], boxes[:, :, 2], scores, classes]
    else:
        cols = [ids, boxes[:, :, 1], boxes[:, :, 0], boxes[:, :, 3], boxes[:, :, 2], scores, classes]
    if nms_multi_class_bs is not None:
        m = np.asarray(nms_multi_class_bs, dtype=np.float32)
        cols += [m[:, :, i] for i in range(m.shape[-1])]
    return np.stack(cols, axis=-1).astype(np.float32)


def generate_detections(params, cls_outputs, box_outputs, image_scales, image_ids, flip=False, per_class_nms=True,
                        driver=None):
    """The legacy interface over raw head outputs (src/postprocess.py:788-871; caller eval.py:117-123): post-process on
    the GPU, then pack [id, x, y, x2, y2, score, class(, logits...)] rows.  `driver` (keyword-only use) pins the
    handle; by default resident heads are processed in their own handle, arrays in a cached post-process handle.  The
    numpy-NMS branch (`nms_configs.pyfunc`) is dead in the reference (key typo `enable_softnax` at :806) and is
    rejected by the driver."""
    p = _params_dict(params)
    _, width = parse_image_size(p["image_size"])
    widths = np.asarray(image_scales, dtype=np.float32)[:, None] * np.float32(width)
    out = _run(p, cls_outputs, box_outputs, image_scales, "per_class" if per_class_nms else "global", driver)
    boxes, scores, classes = out[0][..., :4], out[1], out[2]
    if classes.ndim == 3:
        classes = classes[..., 0]
    logits = out[4] if (len(out) > 4 and p["enable_softmax"]) else None
    return generate_detections_from_nms_output(boxes, classes, scores, image_ids, widths, flip, logits)


def transform_detections(detections):
    """[id, x1, y1, x2, y2, score, class] -> [id, x, y, w, h, score, class] (postprocess.py:874-887)."""
    d = np.asarray(detections)
    return np.stack([d[:, :, 0], d[:, :, 1], d[:, :, 2], d[:, :, 3] - d[:, :, 1], d[:, :, 4] - d[:, :, 2],
                     d[:, :, 5], d[:, :, 6]], axis=-1)


def unpack_detections(params, detections, probab=None, entropy=None):
    """Split the serve() tuple the way the reference's callers do (validate_model.py:159-202, infer_model.py:585-636):
    box columns 4: are the aleatoric and / or MC (epistemic) box std, class columns 1: the MC std of the logits;
    NaNs in the uncertainty columns become 0 (`np.nan_to_num`)."""
    boxes, scores, classes, valid = detections[:4]
    logits = detections[4] if len(detections) > 4 else None
    mc_box = bool(params.get("mc_boxheadrate") or params.get("mc_dropoutrate")) and bool(params.get("mc_dropout"))
    mc_cls = bool(params.get("mc_classheadrate") or params.get("mc_dropoutrate")) and bool(params.get("mc_dropout"))
    la = bool(params.get("loss_attenuation"))
    albox = mcbox = mcclass = None
    if mc_box and not la:
        mcbox = np.nan_to_num(boxes[:, :, 4:])
    elif mc_box and la:
        albox = np.nan_to_num(boxes[:, :, 4:8])
        mcbox = np.nan_to_num(boxes[:, :, 8:])
    elif la:
        albox = np.nan_to_num(boxes[:, :, 4:])
    if mc_cls and classes.ndim == 3:
        mcclass = np.nan_to_num(classes[:, :, 1:])
        classes = classes[:, :, 0]
    return dict(boxes=boxes[:, :, :4], scores=scores, classes=classes, valid_len=valid, logits=logits,
                probab=probab, entropy=entropy, albox=albox, mcbox=mcbox, mcclass=mcclass)
