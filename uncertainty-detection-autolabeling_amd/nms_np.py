"""The reference's numpy NMS family (src/nms_np.py) on the GPU, same function names and arguments (row a18).

`postprocess.generate_detections(..., nms_configs.pyfunc)` is the only caller in the reference and it is dead there
(key typo at postprocess.py:806); the functions are still public API of the module, so they exist here too.
"""
import ctypes as C

import numpy as np

from . import capi

MAX_DETECTIONS_PER_IMAGE = 100
_METHOD = {"hard": 0, None: 0, "": 0, "diou": 1, "gaussian": 2, "linear": 3}


def _run(dets, method, iou_thresh, sigma, score_thresh, device=0):
    lib = capi.load()
    d = np.ascontiguousarray(dets, np.float64)
    out = np.empty_like(d)
    n_out = C.c_int32(0)
    rc = lib.uda_nms_np(device, d.ctypes.data, d.shape[0], method, float(iou_thresh), float(sigma), float(score_thresh),
                        out.ctypes.data, C.byref(n_out))
    if rc:
        raise capi.UdaError(lib.uda_last_error(None).decode())
    return out[:n_out.value]


def hard_nms(dets, iou_thresh=None):
    return _run(dets, 0, iou_thresh or 0.5, 0.5, 0.0)


def diou_nms(dets, iou_thresh=None):
    return _run(dets, 1, iou_thresh or 0.5, 0.5, 0.0)


def soft_nms(dets, nms_configs):
    method = nms_configs["method"]
    if method not in ("gaussian", "linear"):
        raise ValueError("soft_nms method must be gaussian or linear, got {}".format(method))
    return _run(dets, _METHOD[method], nms_configs["iou_thresh"] or 0.3, nms_configs["sigma"] or 0.5,
                nms_configs["score_thresh"] or 0.001)


def nms(dets, nms_configs):
    method = (nms_configs or {})["method"]
    if method == "hard" or not method:
        return hard_nms(dets, nms_configs["iou_thresh"])
    if method == "diou":
        return diou_nms(dets, nms_configs["iou_thresh"])
    if method in ("linear", "gaussian"):
        return soft_nms(dets, nms_configs)
    raise ValueError("Unknown NMS method: {}".format(method))


def per_class_nms(boxes, scores, classes, image_id, image_scale, num_classes, max_boxes_to_draw=None, nms_configs=None,
                  device=0):
    lib = capi.load()
    max_boxes = max_boxes_to_draw or MAX_DETECTIONS_PER_IMAGE
    cfg = nms_configs or {}
    method = cfg.get("method")
    if method not in _METHOD:
        raise ValueError("Unknown NMS method: {}".format(method))
    m = _METHOD[method]
    if m <= 1:
        thr, sigma, sthr = cfg.get("iou_thresh") or 0.5, 0.5, 0.0
    else:
        thr, sigma, sthr = cfg.get("iou_thresh") or 0.3, cfg.get("sigma") or 0.5, cfg.get("score_thresh") or 0.001
    b = np.ascontiguousarray(boxes, np.float32)
    s = np.ascontiguousarray(scores, np.float32)
    c = np.ascontiguousarray(classes, np.int32)
    out = np.empty((max_boxes, 7), np.float32)
    rc = lib.uda_per_class_nms_np(device, b.ctypes.data, s.ctypes.data, c.ctypes.data, b.shape[0],
                                  float(np.asarray(image_id).reshape(-1)[0]), float(np.asarray(image_scale).reshape(-1)[0]),
                                  int(num_classes), int(max_boxes), m, float(thr), float(sigma), float(sthr), out.ctypes.data)
    if rc:
        raise capi.UdaError(lib.uda_last_error(None).decode())
    return out
