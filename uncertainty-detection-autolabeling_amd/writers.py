"""The on-disk record every downstream experiment parses: `prediction_data.txt`, one python-dict
literal per detection above the score threshold, as `Infer` writes it (reference src/infer_model.py:836-960,
`add_array_dict` utils_extra.py:67-81) and as active learning / SSL / thresholding read it back with
`ast.literal_eval(line.replace("inf", "2e308"))` (active_learning_loop.py:532, SSL_stac.py:345).

Values are plain python floats / lists (the reference's numpy-1 scalars print the same way), arrays rounded to 4
decimals with NaNs zeroed exactly like `add_array_dict`.
"""
import numpy as np


def add_array_dict(data_dict, source_array, target_key, select_index):
    """utils_extra.add_array_dict: rounded (4 decimals), NaN-free copy of row `select_index` under `target_key`."""
    source_array = np.asarray(source_array)
    if source_array.size > 0:
        vals = np.nan_to_num(np.around(source_array[select_index].astype("float32"), 4))
        data_dict[target_key] = [float(v) for v in np.ravel(vals)] if np.size(vals) > 1 else float(vals)
    return data_dict


def prediction_records(unpacked, image_names, min_score, calibrated=None):
    """Records of `ServingDriver.serve_unpacked` output for a batch, in the reference's key order.

    calibrated: optional dict name -> [N, M, ...] arrays written next to the uncalibrated columns, e.g.
    {"iso_all_albox": ..., "ts_all_mcbox": ...} (`BoxCalibrator.calibrate_boxuncert`)."""
    recs = []
    calibrated = calibrated or {}
    boxes, scores, classes = unpacked["boxes"], unpacked["scores"], unpacked["classes"]
    for i, name in enumerate(image_names):
        base = {"image_name": name + ".jpg", "score_thresh": float(min_score),
                "top_5scores": [float(s) for s in scores[i][:5]]}
        for sel in np.where(scores[i] > min_score)[0]:
            d = dict(base)
            d["det_score"] = float(scores[i][sel])
            d["bbox"] = [float(v) for v in boxes[i][sel]]
            d["class"] = float(classes[i][sel])
            if unpacked.get("logits") is not None:
                add_array_dict(d, unpacked["logits"][i], "logits", sel)
                add_array_dict(d, unpacked["entropy"][i], "entropy", sel)
                d["probab"] = [float(v) for v in unpacked["probab"][i][sel]]
            for key, col in (("uncalib_mcclass", "mcclass"), ("uncalib_albox", "albox"), ("uncalib_mcbox", "mcbox")):
                if unpacked.get(col) is not None:
                    add_array_dict(d, unpacked[col][i], key, sel)
                    for cname, arr in calibrated.items():
                        if cname.endswith("_" + col):
                            add_array_dict(d, arr[i], cname, sel)
            recs.append(d)
    return recs


def write_prediction_data(path, records):
    """Append one `str(dict)` line per record (infer_model.py:956-960)."""
    with open(path, "a") as f:
        for r in records:
            f.write(str(r) + "\n")
