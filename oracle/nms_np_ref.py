"""ORACLE (test infrastructure): the numpy NMS family of the reference (row a18).

Restates src/nms_np.py:30-278.  This is the one part of the hot path whose
reference imports in this container (numpy only), so the restatement is PINNED
against golden vectors produced by the real module
(tests/golden/make_nms_np_golden.py -> tests/golden/nms_np_golden.npz).

Conventions that differ from the TF path (postprocess.nms):
  * boxes are [x1, y1, x2, y2]; areas and intersections use the +1 pixel convention
    (nms_np.py:52,67-68,109,121-122,160,176-177)
  * soft-NMS rescoring: weight = exp(-iou^2 / sigma) (gaussian) or 1 - iou above the
    threshold (linear); a box is dropped when its score falls below score_thresh (:166-192)
  * sorting is `argsort()[::-1]` — unstable under ties, so fixtures use distinct scores
"""
import numpy as np

DUMMY_SCORE = -1e5


def _areas(d):
    return (d[:, 2] - d[:, 0] + 1) * (d[:, 3] - d[:, 1] + 1)


def _iou_one_to_many(box, area, others, areas):
    w = np.maximum(0.0, np.minimum(box[2], others[:, 2]) - np.maximum(box[0], others[:, 0]) + 1)
    h = np.maximum(0.0, np.minimum(box[3], others[:, 3]) - np.maximum(box[1], others[:, 1]) + 1)
    inter = w * h
    return inter / (area + areas - inter)


def hard_nms(dets, iou_thresh=None):
    thr = iou_thresh or 0.5
    areas = _areas(dets)
    order = dets[:, 4].argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        rest = order[1:]
        iou = _iou_one_to_many(dets[i], areas[i], dets[rest], areas[rest])
        order = rest[iou <= thr]
    return dets[keep]


def diou_nms(dets, iou_thresh=None):
    thr = iou_thresh or 0.5
    areas = _areas(dets)
    cx, cy = (dets[:, 0] + dets[:, 2]) / 2, (dets[:, 1] + dets[:, 3]) / 2
    order = dets[:, 4].argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        rest = order[1:]
        iou = _iou_one_to_many(dets[i], areas[i], dets[rest], areas[rest])
        ex1, ex2 = np.minimum(dets[i, 0], dets[rest, 0]), np.maximum(dets[i, 2], dets[rest, 2])
        ey1, ey2 = np.minimum(dets[i, 1], dets[rest, 1]), np.maximum(dets[i, 3], dets[rest, 3])
        diag = (ex2 - ex1) ** 2 + (ey2 - ey1) ** 2
        dist = (cx[i] - cx[rest]) ** 2 + (cy[i] - cy[rest]) ** 2
        order = rest[(iou - dist / (diag + 1e-10)) <= thr]
    return dets[keep]


def soft_nms(dets, nms_configs):
    method = nms_configs["method"]
    sigma = nms_configs["sigma"] or 0.5
    thr = nms_configs["iou_thresh"] or 0.3
    score_thr = nms_configs["score_thresh"] or 0.001
    work = np.concatenate([dets, _areas(dets)[:, None]], axis=1)   # x1 y1 x2 y2 score area
    out = []
    while work.size > 0:
        m = int(np.argmax(work[:, 4]))
        work[[0, m]] = work[[m, 0]]
        out.append(work[0, :5].copy())
        iou = _iou_one_to_many(work[0], work[0, 5], work[1:], work[1:, 5])
        if method == "linear":
            wgt = np.where(iou > thr, 1.0 - iou, 1.0)
        elif method == "gaussian":
            wgt = np.exp(-(iou * iou) / sigma)
        else:
            wgt = np.where(iou > thr, 0.0, 1.0)
        work[1:, 4] *= wgt
        work = work[1:][work[1:, 4] >= score_thr]
    return np.vstack(out)


def nms(dets, nms_configs):
    method = (nms_configs or {})["method"]
    if method == "hard" or not method:
        return hard_nms(dets, nms_configs["iou_thresh"])
    if method == "diou":
        return diou_nms(dets, nms_configs["iou_thresh"])
    if method in ("linear", "gaussian"):
        return soft_nms(dets, nms_configs)
    raise ValueError("Unknown NMS method: {}".format(method))


def per_class_nms(boxes, scores, classes, image_id, image_scale, num_classes,
                  max_boxes_to_draw, nms_configs):
    """boxes [K,4] (y1,x1,y2,x2) -> float32 [max_boxes, 7] rows
    [image_id, x1, y1, x2, y2, score, class]; dummy rows score -1e5 (:223-278)."""
    xyxy = boxes[:, [1, 0, 3, 2]]
    parts = []
    for c in range(num_classes):
        sel = np.where(classes == c)[0]
        if sel.size == 0:
            continue
        top = nms(np.column_stack((xyxy[sel], scores[sel])), nms_configs)
        parts.append(np.column_stack((np.repeat(image_id, len(top)), top,
                                      np.repeat(c + 1, len(top)))))

    def dummy(k):
        d = np.zeros((k, 7), dtype=np.float32)
        d[:, 0] = image_id[0]
        d[:, 5] = DUMMY_SCORE
        return d

    if parts:
        det = np.vstack(parts)
        det = np.array(det[np.argsort(-det[:, -2])[:max_boxes_to_draw]], dtype=np.float32)
        det = np.vstack([det, dummy(max(max_boxes_to_draw - len(det), 0))])
    else:
        det = dummy(max_boxes_to_draw)
    det[:, 1:5] *= image_scale
    return det
