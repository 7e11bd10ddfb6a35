"""ORACLE (test infrastructure): what the reference's callers compute from the serve() tuple.

  stable_softmax                     utils_class.py:36-41
  entropy / column unpacking         validate_model.py:159-202 (same code in infer_model.py:585-636)

Plain numpy in float32, row by row as the reference writes it.
"""
import numpy as np


def stable_softmax(logits):
    out = []
    for x in np.asarray(logits, np.float32):
        e = np.exp(x - max(x))
        out.append(e / np.sum(e))
    return np.asarray(out, np.float32)


def probab_entropy(logits):
    """logits [M, C] -> (probab [M, C], entropy [M])."""
    p = stable_softmax(logits)
    ent = -np.sum(p * np.nan_to_num(np.log2(np.maximum(p, 10 ** -7))), axis=1)
    return p, ent.astype(np.float32)


def unpack(params, boxes, classes):
    """(boxes4, classes_id, albox, mcbox, mcclass) with the reference's branch structure."""
    mc_box = params["mc_boxheadrate"] or params["mc_dropoutrate"]
    mc_cls = params["mc_classheadrate"] or params["mc_dropoutrate"]
    la = params["loss_attenuation"]
    if mc_box and not la:
        mcbox, albox = np.nan_to_num(boxes[:, :, 4:]), None
    elif mc_box and la:
        albox, mcbox = np.nan_to_num(boxes[:, :, 4:8]), np.nan_to_num(boxes[:, :, 8:])
    elif (not mc_box) and la:
        albox, mcbox = np.nan_to_num(boxes[:, :, 4:]), None
    else:
        albox = mcbox = None
    if mc_cls:
        mcclass = np.nan_to_num(classes[:, :, 1:])
        classes = classes[:, :, 0]
    else:
        mcclass = None
    if mcbox is not None or albox is not None:
        boxes = boxes[:, :, :4]
    return boxes, classes, albox, mcbox, mcclass
