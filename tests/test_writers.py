"""the two text records downstream experiments parse, against lines WRITTEN OUT BY HAND in the
reference's format (what `str(dict)` of numpy-1 float32 scalars / lists gives: shortest float32 repr, arrays rounded to
4 decimals by add_array_dict, the statement order of src/infer_model.py:836-960 and src/validate_model.py:524-681) -
not a round trip through the writer itself."""
import ast

import numpy as np

from uda_amd import writers


def _unpacked():
    f = np.float32
    return dict(
        boxes=np.array([[[10.5, 20.25, 110.125, 220.0], [1.0, 2.0, 3.0, 4.0]]], f),
        scores=np.array([[0.8765, 0.1234]], f), classes=np.array([[3.0, 1.0]], f), valid_len=np.array([2], np.int32),
        logits=np.array([[[1.23456, -0.5, 0.0], [0.1, 0.2, 0.3]]], f),
        probab=np.array([[[0.7, 0.2, 0.1], [0.3, 0.3, 0.4]]], f),
        entropy=np.array([[1.156789, 1.5]], f),
        albox=np.array([[[0.51234, np.nan, 2.0, 3.33333], [0.1, 0.1, 0.1, 0.1]]], f),
        mcbox=np.array([[[1.0, 1.5, 0.25, 0.125], [0.2, 0.2, 0.2, 0.2]]], f),
        mcclass=np.array([[[0.01, 0.02222, 0.03], [0.0, 0.0, 0.0]]], f))


def test_prediction_data_line_is_text_identical_to_the_reference_format(tmp_path):
    un = _unpacked()
    cal = {"ts_all_probab": np.array([[[0.6, 0.3, 0.1], [0.0, 0.0, 1.0]]], np.float32),
           "iso_all_entropy": np.array([[1.25, 0.5]], np.float32),
           "iso_all_albox": un["albox"] * 2, "ts_all_albox": un["albox"] / 2,
           "rel_iso_perclscoo_mcbox": un["mcbox"] * 4, "ts_percls_mcclass": un["mcclass"] * 10}
    recs = writers.prediction_records(un, ["000123"], 0.45, calibrated=cal)
    assert len(recs) == 1                                   # only the 0.8765 detection passes the threshold
    path = tmp_path / "prediction_data.txt"
    writers.write_prediction_data(str(path), recs)
    want = ("{'image_name': '000123.jpg', 'score_thresh': 0.45, 'top_5scores': [0.8765, 0.1234], 'det_score': 0.8765, "
            "'bbox': [10.5, 20.25, 110.125, 220.0], 'class': 3.0, 'logits': [1.2346, -0.5, 0.0], 'entropy': 1.1568, "
            "'probab': [0.7, 0.2, 0.1], 'ts_all_probab': [0.6, 0.3, 0.1], 'iso_all_entropy': 1.25, "
            "'uncalib_mcclass': [0.01, 0.0222, 0.03], 'ts_percls_mcclass': [0.1, 0.2222, 0.3], "
            "'uncalib_albox': [0.5123, 0.0, 2.0, 3.3333], 'iso_all_albox': [1.0247, 0.0, 4.0, 6.6667], "
            "'ts_all_albox': [0.2562, 0.0, 1.0, 1.6667], 'uncalib_mcbox': [1.0, 1.5, 0.25, 0.125], "
            "'rel_iso_perclscoo_mcbox': [4.0, 6.0, 1.0, 0.5]}\n")
    assert open(path).read() == want
    back = ast.literal_eval(want.replace("inf", "2e308"))   # the readers' parse (active_learning_loop.py:532)
    assert back["bbox"] == [10.5, 20.25, 110.125, 220.0] and back["class"] == 3.0
    # append mode, and the reference's one-dict-per-image behaviour: keys of an earlier detection persist
    un2 = _unpacked()
    un2["scores"][0, 1] = 0.5
    un2["mcbox"] = None
    recs2 = writers.prediction_records(un2, ["000124"], 0.45)
    assert len(recs2) == 2 and list(recs2[1]) == list(recs2[0]) and "uncalib_mcbox" not in recs2[0]
    writers.write_prediction_data(str(path), recs2)
    assert len(open(path).read().splitlines()) == 3


def test_infinite_and_tiny_values_print_like_numpy_float32():
    un = _unpacked()
    un["albox"][0, 0] = [np.inf, 1e-10, 123456.789, 0.00004]
    r = writers.prediction_records(un, ["x"], 0.45)[0]
    assert str(r["uncalib_albox"]) == "[3.4028235e+38, 0.0, 123456.8, 0.0]"      # nan_to_num(inf) = float32 max; np.around works in float32
    assert str(writers._f32(np.float32(1e-10))) == "1e-10" and str(writers._f32(np.float32(0.1))) == "0.1"


def test_validate_results_lines_follow_the_reference_statement_order(tmp_path):
    f = np.float32
    params = dict(enable_softmax=True, calibrate_classification=True, calibrate_regression=True, mc_classheadrate=0.0,
                  mc_boxheadrate=0.0, mc_dropoutrate=0.05, loss_attenuation=True)
    filt = dict(names=["000007.png", "000007.png"], scores=np.array([0.9, 0.75], f),
                boxes=np.array([[1, 2, 3, 4], [10.5, 20.5, 30.5, 40.5]], f), gt_boxes=np.array([[1, 2, 3, 5], [10, 20, 30, 40]], f),
                occlusions=[0, 2], truncations=[0.0, 0.35], classes=np.array([1, 4]), gt_classes=np.array([1, 3]),
                logits=np.array([[2.0, -1.0], [0.5, 0.25]], f), probab=np.array([[0.95, 0.05], [0.5625, 0.4375]], f),
                entropy=np.array([0.2864, 0.9887], f), mcclass=np.array([[0.1, 0.2], [0.3, 0.4]], f),
                mcbox=np.array([[0.5, 0.5, 0.5, 0.5], [1, 2, 3, 4]], f), albox=np.array([[1.5, 1.5, 1.5, 1.5], [2, 2, 2, 2]], f))
    cal = {"iso_all_probab": filt["probab"], "ts_all_entropy": np.array([0.3, 1.0], f), "ts_percoo_mcbox": filt["mcbox"] / 2,
           "iso_all_albox": filt["albox"] * 2, "iso_percls_mcclass": filt["mcclass"]}
    recs = writers.validate_records(filt, params, calibrated=cal)
    path = tmp_path / "validate_results.txt"
    writers.write_validate_results(str(path), recs)
    want0 = ("{'image_name': '000007.png', 'score': 0.9, 'bbox': [1.0, 2.0, 3.0, 4.0], 'gt_bbox': [1.0, 2.0, 3.0, 5.0], 'gt_occl': 0, "
             "'gt_trunc': 0.0, 'class': 1, 'gt_class': 1, 'logits': [2.0, -1.0], 'probab': [0.95, 0.05], 'entropy': 0.2864, "
             "'iso_all_probab': [0.95, 0.05], 'ts_all_entropy': 0.3, 'uncalib_mcclass': [0.1, 0.2], 'iso_percls_mcclass': [0.1, 0.2], "
             "'uncalib_mcbox': [0.5, 0.5, 0.5, 0.5], 'ts_percoo_mcbox': [0.25, 0.25, 0.25, 0.25], "
             "'uncalib_albox': [1.5, 1.5, 1.5, 1.5], 'iso_all_albox': [3.0, 3.0, 3.0, 3.0]}")
    lines = open(path).read().splitlines()
    assert len(lines) == 2 and lines[0] == want0
    second = ast.literal_eval(lines[1])
    assert second["gt_occl"] == 2 and second["gt_trunc"] == 0.35 and second["class"] == 4 and second["ts_percoo_mcbox"] == [0.5, 1.0, 1.5, 2.0]
    # mcbox before albox in THIS file, albox before mcbox in prediction_data.txt
    keys = list(second)
    assert keys.index("uncalib_mcbox") < keys.index("uncalib_albox")
    # a deterministic model writes neither MC key
    recs_det = writers.validate_records(filt, dict(params, mc_dropoutrate=0.0, loss_attenuation=False, enable_softmax=False))
    assert list(recs_det[0]) == ["image_name", "score", "bbox", "gt_bbox", "gt_occl", "gt_trunc", "class", "gt_class"]
    writers.write_validate_results(str(path), recs_det)          # "w": the file is rewritten
    assert len(open(path).read().splitlines()) == 2


def test_runtime_summary_follows_the_outlier_rules():
    times = [0.010] * 50 + [0.011] * 49 + [1.5, 0.9]        # 1.5 s: dropped (>= 1 s); 0.9 s: beyond q3 + 50 IQR
    out = writers.summarize_runtimes(times)
    assert out[0].startswith("Mean time in ms: 10.") and out[2] == "Median time in ms: 10.000\n" and out[1].startswith("STD time in ms: 0.")
