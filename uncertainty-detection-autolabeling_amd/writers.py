"""The on-disk records every downstream experiment parses.

  prediction_data.txt   one python-dict literal per detection above the score threshold, as `Infer.iterate_infer`
                        writes it (reference src/infer_model.py:836-960, `add_array_dict` utils_extra.py:67-81)
  validate_results.txt  one dict per detection matched to a ground-truth box, as `Validate.launch_val` writes it
                        (src/validate_model.py:524-681), plus the runtime summary of validationstep_runtime.txt
                        (:685-704)

Both are read back with `ast.literal_eval(line.replace("inf", "2e308"))` (active_learning_loop.py:532,
SSL_stac.py:345, uncertainty_analysis.py).  The reference prints numpy-1 float32 scalars, whose repr is the shortest
decimal that round-trips in float32 ('0.1234', not '0.12340000271797180'); the values are written here as the python
float with that shortest repr, so a line is text-identical to the reference's for the same numbers.  Key order follows
the reference statement by statement (it differs between the two files: albox before mcbox in prediction_data,
mcbox before albox in validate_results).
"""
import numpy as np

CLASS_CALIB_KEYS_INFER = ("ts_all", "ts_percls", "iso_all", "iso_percls")          # infer_model.py:868-893, 903-915
CLASS_CALIB_KEYS_VAL = ("iso_all", "ts_all", "ts_percls", "iso_percls")            # validate_model.py:547-590, 597-618
BOX_CALIB_KEYS_INFER = ("iso_all", "ts_all", "ts_percoo", "iso_percoo", "iso_perclscoo", "rel_iso_perclscoo")
BOX_CALIB_KEYS_VAL = BOX_CALIB_KEYS_INFER                                          # validate_model.py:625-655


def _f32(v):
    """python float that prints like numpy 1's repr of the float32 value v."""
    return float(str(np.float32(v)))


def _f32_list(a):
    return [_f32(v) for v in np.ravel(np.asarray(a))]


def add_array_dict(data_dict, source_array, target_key, select_index):
    """utils_extra.add_array_dict (:67-81): rounded (4 decimals), NaN-free copy of row `select_index` under `target_key`."""
    source_array = np.asarray(source_array)
    if source_array.size > 0:
        vals = np.nan_to_num(np.around(source_array[select_index].astype("float32"), 4))
        data_dict[target_key] = _f32_list(vals) if np.size(source_array[select_index]) > 1 else _f32(vals)
    return data_dict


def prediction_records(unpacked, image_names, min_score, calibrated=None, calibrate_classification=True,
                       calibrate_regression=True):
    """Records of `ServingDriver.serve_unpacked` output for a batch, in the reference's key order
    (infer_model.py:836-960).

    calibrated: optional dict name -> [N, M, ...] arrays written behind the uncalibrated column they refine:
    "<method>_probab" / "<method>_entropy" / "<method>_mcclass" (`ClassCalibrator`), "<method>_albox" /
    "<method>_mcbox" (`BoxCalibrator`); a missing name is skipped like the reference's empty array."""
    recs = []
    cal = calibrated or {}
    boxes, scores, classes = unpacked["boxes"], unpacked["scores"], unpacked["classes"]

    def add_cal(d, names, suffix, i, sel):
        for m in names:
            key = "%s_%s" % (m, suffix)
            if key in cal:
                add_array_dict(d, np.asarray(cal[key])[i], key, sel)

    for i, name in enumerate(image_names):
        # the reference mutates ONE dict per image: a key written for an earlier detection stays in the later lines
        d = {"image_name": name + ".jpg", "score_thresh": float(min_score), "top_5scores": _f32_list(scores[i][:5])}
        for sel in np.where(scores[i] > min_score)[0]:
            d["det_score"] = _f32(scores[i][sel])
            d["bbox"] = _f32_list(boxes[i][sel])
            d["class"] = _f32(classes[i][sel])
            if unpacked.get("logits") is not None:
                add_array_dict(d, unpacked["logits"][i], "logits", sel)
                add_array_dict(d, unpacked["entropy"][i], "entropy", sel)
                d["probab"] = _f32_list(unpacked["probab"][i][sel])
                if calibrate_classification:
                    add_cal(d, CLASS_CALIB_KEYS_INFER, "probab", i, sel)
                    add_cal(d, CLASS_CALIB_KEYS_INFER, "entropy", i, sel)
            if unpacked.get("mcclass") is not None:
                add_array_dict(d, unpacked["mcclass"][i], "uncalib_mcclass", sel)
                if calibrate_classification:
                    add_cal(d, CLASS_CALIB_KEYS_INFER, "mcclass", i, sel)
            if unpacked.get("albox") is not None:
                add_array_dict(d, unpacked["albox"][i], "uncalib_albox", sel)
                if calibrate_regression:
                    add_cal(d, BOX_CALIB_KEYS_INFER, "albox", i, sel)
            if unpacked.get("mcbox") is not None:
                add_array_dict(d, unpacked["mcbox"][i], "uncalib_mcbox", sel)
                if calibrate_regression:
                    add_cal(d, BOX_CALIB_KEYS_INFER, "mcbox", i, sel)
            recs.append(dict(d))
    return recs


def write_prediction_data(path, records):
    """Append one `str(dict)` line per record (infer_model.py:956-960)."""
    with open(path, "a") as f:
        for r in records:
            f.write(str(r) + "\n")


def predict_to_file(driver, batches, names, path, min_score, box_calibrator=None, class_calibrator=None,
                    box_methods=(), class_methods=()):
    """The inference flow of `Infer.iterate_infer` (infer_model.py:554-960) as ONE loop over batches of images (arrays, or
    lists of images of different raw sizes): serve (feed of the next batch hidden under this one) -> unpack + softmax /
    entropy (device) -> calibrated box / class uncertainties (device, `calibration.BoxCalibrator` / `ClassCalibrator`)
    -> `prediction_data.txt` lines appended to `path`.  `names`: one list of image names per batch.  Returns the number
    of records written."""
    from . import postprocess as pp
    names = list(names)
    state = {"i": 0, "written": 0}

    def per_batch(det):
        n = det[0].shape[0]
        probs = ent = None
        if driver.params["enable_softmax"]:
            probs, ent = driver.class_probs(n)
        un = pp.unpack_detections(driver.params, det, probs, ent)
        cal = {}
        if box_calibrator is not None:
            for m in box_methods:
                for which in ("albox", "mcbox"):
                    if un.get(which) is not None:
                        cal["%s_%s" % (m, which)] = box_calibrator.calibrate_boxuncert(n, which, m)
        if class_calibrator is not None and un.get("logits") is not None:
            for m in class_methods:
                r = class_calibrator.perform_class_calib(n, m)
                cal[m + "_entropy"], cal[m + "_probab"] = r[0], r[1]
                if len(r) > 2:
                    cal[m + "_mcclass"] = r[2]
        recs = prediction_records(un, names[state["i"]], min_score, cal)
        write_prediction_data(path, recs)
        state["i"] += 1
        state["written"] += len(recs)
        return len(recs)

    for _ in driver.serve_stream(batches, while_resident=per_batch):
        pass
    return state["written"]


def validate_records(filtered, params, calibrated=None):
    """Records of `validate_results.txt` (validate_model.py:524-681) from the per-detection arrays the validation
    harness keeps after ground-truth assignment (CPU numpy analysis outside the hot path).

    filtered: dict with `names` [K] str, `scores` [K], `boxes` [K,4], `gt_boxes` [K,4], `occlusions` [K],
    `truncations` [K], `classes` [K], `gt_classes` [K] and, as the configuration produces them, `logits` [K,C],
    `probab` [K,C], `entropy` [K], `mcclass` [K,C], `mcbox` [K,4], `albox` [K,4].
    calibrated: optional dict "<method>_<probab|entropy|mcclass|mcbox|albox>" -> [K, ...]."""
    cal = calibrated or {}
    recs = []
    K = len(filtered["boxes"])

    def add_cal(d, names, suffix, i):
        for m in names:
            key = "%s_%s" % (m, suffix)
            if key in cal:
                add_array_dict(d, np.asarray(cal[key]), key, i)

    for i in range(K):
        d = {"image_name": filtered["names"][i], "score": _f32(filtered["scores"][i]), "bbox": _f32_list(filtered["boxes"][i]),
             "gt_bbox": _f32_list(filtered["gt_boxes"][i]), "gt_occl": _plain(filtered["occlusions"][i]),
             "gt_trunc": _plain(filtered["truncations"][i]), "class": _plain(filtered["classes"][i]),
             "gt_class": _plain(filtered["gt_classes"][i])}
        if params["enable_softmax"]:
            d["logits"] = _f32_list(filtered["logits"][i])
            d["probab"] = _f32_list(filtered["probab"][i])
            d["entropy"] = _f32(filtered["entropy"][i])
            if params.get("calibrate_classification"):
                add_cal(d, CLASS_CALIB_KEYS_VAL, "probab", i)
                add_cal(d, CLASS_CALIB_KEYS_VAL, "entropy", i)
        if params.get("mc_classheadrate") or params.get("mc_dropoutrate"):
            d["uncalib_mcclass"] = _f32_list(filtered["mcclass"][i])
            if params.get("calibrate_classification"):
                add_cal(d, CLASS_CALIB_KEYS_VAL, "mcclass", i)
        if params.get("mc_boxheadrate") or params.get("mc_dropoutrate"):
            d["uncalib_mcbox"] = _f32_list(filtered["mcbox"][i])
            if params.get("calibrate_regression"):
                add_cal(d, BOX_CALIB_KEYS_VAL, "mcbox", i)
        if params.get("loss_attenuation"):
            d["uncalib_albox"] = _f32_list(filtered["albox"][i])
            if params.get("calibrate_regression"):
                add_cal(d, BOX_CALIB_KEYS_VAL, "albox", i)
        recs.append(d)
    return recs


def _plain(v):
    """ints stay ints, float32 scalars print like numpy 1 (ground-truth fields come from the label files)."""
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        return _f32(v) if isinstance(v, np.float32) else float(v)
    return v


def write_validate_results(path, records):
    """One `str(dict)` line per record, file rewritten (validate_model.py:526, :681)."""
    with open(path, "w") as f:
        for r in records:
            f.write(str(r) + "\n")


def summarize_runtimes(seconds):
    """The three lines `Validate.launch_val` leaves in validationstep_runtime.txt (validate_model.py:685-704): per-image
    serve() wall clock, calls of 1 s or more dropped, then outliers above q3 + 50 IQR dropped; mean / std / median in ms."""
    t = np.asarray(seconds, np.float32)
    t = t[t < 1]
    q3 = np.percentile(t, 75)
    iqr = np.percentile(t, 75) - np.percentile(t, 25)
    kept = [x for x in t if x <= q3 + 50 * iqr]
    return ["Mean time in ms: {:.3f}\n".format(np.mean(kept) * 1000), "STD time in ms: {:.3f}\n".format(np.std(kept) * 1000),
            "Median time in ms: {:.3f}\n".format(np.median(kept) * 1000)]
