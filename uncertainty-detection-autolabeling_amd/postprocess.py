"""Legacy detection row formats on top of the driver's output tuple.

Mirrors `generate_detections_from_nms_output` / `generate_detections` / `transform_detections`
of the reference (src/postprocess.py:743-887): pure re-packing of <=100 rows per image that the
reference does with tf.stack on the host side of the path; the arithmetic (network, decode, NMS)
stays on the GPU behind `ServingDriver`.
"""
import numpy as np

from .hparams_config import parse_image_size


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


def generate_detections(driver, params, cls_outputs, box_outputs, image_scales, image_ids, flip=False,
                        per_class_nms=True):
    """The legacy interface over raw head outputs (postprocess.py:788-871): post-process on the GPU
    (`driver.postprocess`), then pack rows.  The numpy-NMS branch (`nms_configs.pyfunc`) is dead in
    the reference (key typo at :806) and is rejected by the driver."""
    _, width = parse_image_size(params["image_size"])
    widths = np.asarray(image_scales, dtype=np.float32)[:, None] * np.float32(width)
    out = driver.postprocess(cls_outputs, box_outputs, image_scales,
                             post_mode="per_class" if per_class_nms else "global")
    boxes, scores, classes = out[0][..., :4], out[1], out[2]
    if classes.ndim == 3:
        classes = classes[..., 0]
    logits = out[4] if (len(out) > 4 and params["enable_softmax"]) else None
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
