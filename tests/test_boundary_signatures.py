"""The drop-in boundary keeps the reference's call shapes (CPU: signatures only, no compute).

Reference: src/infer_lib.py:154-192 (create / __init__), :299-311 (SavedModelDriver), :416-440 (KerasDriver),
:194-204 (visualize); src/postprocess.py:472,719,788,874; src/utils_extra.py:119,142,201,220."""
import inspect

import numpy as np
import pytest

from uda_amd import efficientdet_keras, infer_lib, postprocess, utils_extra, utils_keras


def _positional(fn):
    return [p.name for p in inspect.signature(fn).parameters.values()
            if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]


def test_driver_constructors_bind_the_reference_positionals():
    assert _positional(infer_lib.ServingDriver.__init__)[1:] == ["model_name", "batch_size", "only_network", "model_params"]
    assert _positional(infer_lib.KerasDriver.__init__)[1:3] == ["ckpt_path", "debug"]
    assert _positional(infer_lib.SavedModelDriver.__init__)[1:2] == ["saved_model_dir_or_frozen_graph"]
    assert _positional(infer_lib.ServingDriver.create.__func__)[1:4] == ["model_dir", "debug", "saved_model_dir"]
    assert issubclass(infer_lib.KerasDriver, infer_lib.ServingDriver)
    assert issubclass(infer_lib.SavedModelDriver, infer_lib.ServingDriver)
    assert infer_lib.SavedModelDriver is not infer_lib.ServingDriver
    d = inspect.signature(infer_lib.ServingDriver.__init__).parameters
    assert d["batch_size"].default == 1 and d["only_network"].default is False and d["model_params"].default is None
    # the build's extras are keyword-only: they can never capture a positional argument of a reference call
    extras = [p for p in d.values() if p.kind == p.KEYWORD_ONLY]
    assert {p.name for p in extras} >= {"weights", "device", "chunk_images"}
    assert _positional(infer_lib.ServingDriver.visualize)[1:] == ["image", "boxes", "classes", "scores", "uncertainty"]
    assert _positional(infer_lib.ServingDriver.benchmark)[1:] == ["image_arrays", "bm_runs", "trace_filename"]


def test_create_dispatches_like_the_reference(monkeypatch):
    calls = []
    monkeypatch.setattr(infer_lib.KerasDriver, "__init__", lambda self, *a, **k: calls.append(("keras", a, k)))
    monkeypatch.setattr(infer_lib.SavedModelDriver, "__init__", lambda self, *a, **k: calls.append(("saved", a, k)))
    # inspector.py:161-169
    infer_lib.ServingDriver.create("_", False, None, "efficientdet-d0", 1, False, {"num_classes": 7})
    infer_lib.ServingDriver.create("_", True, "/tmp/saved_model", "efficientdet-d0", 1, False, {"num_classes": 7})
    assert calls[0] == ("keras", ("_", False, "efficientdet-d0", 1, False, {"num_classes": 7}), {})
    assert calls[1] == ("saved", ("/tmp/saved_model", "efficientdet-d0", 1, False, {"num_classes": 7}), {})
    with pytest.raises(ValueError):
        infer_lib.ServingDriver.create("_", False, "/tmp/model.tflite", "efficientdet-d0", 1, False, None)


def test_postprocess_and_mc_helpers_keep_the_reference_argument_order():
    assert _positional(postprocess.postprocess_global) == ["params", "cls_outputs", "box_outputs", "image_scales"]
    assert _positional(postprocess.postprocess_per_class) == ["params", "cls_outputs", "box_outputs", "image_scales"]
    assert _positional(postprocess.generate_detections)[:7] == ["params", "cls_outputs", "box_outputs", "image_scales",
                                                                "image_ids", "flip", "per_class_nms"]
    sig = inspect.signature(postprocess.generate_detections).parameters
    assert sig["flip"].default is False and sig["per_class_nms"].default is True
    assert _positional(postprocess.generate_detections_from_nms_output) == [
        "nms_boxes_bs", "nms_classes_bs", "nms_scores_bs", "image_ids", "original_image_widths", "flip", "nms_multi_class_bs"]
    assert _positional(utils_extra.mc_infer) == ["driver", "image", "T"]
    assert _positional(utils_extra.mc_eval) == ["mc_model", "images", "config"]
    assert _positional(utils_keras.restore_ckpt) == ["model", "ckpt_path_or_file", "ema_decay", "skip_mismatch", "exclude_layers"]
    assert _positional(efficientdet_keras.EfficientDetNet.__init__)[1:] == ["model_name", "config", "params", "name"]
    assert _positional(efficientdet_keras.EfficientDetModel.__call__)[1:] == ["inputs", "training", "pre_mode", "post_mode"]
    assert postprocess.CLASS_OFFSET == 1


def test_mc_helpers_on_plain_callables():
    """mc_infer / mc_eval / stack_mcpred / get_mcuncert with stand-in driver and model objects (host logic only)."""
    class Drv:
        def __init__(self):
            self.k = 0

        def serve(self, image):
            self.k += 1
            return (np.full((1, 100, 4), self.k, np.float32), np.full((1, 100), 0.5, np.float32))

    out = utils_extra.mc_infer(Drv(), np.zeros((1, 4, 4, 3), np.uint8), T=3)
    assert out[0].shape == (3, 1, 100, 4) and out[1].shape == (3, 1, 100) and out[0][:, 0, 0, 0].tolist() == [1, 2, 3]

    class Cfg:
        mc_classheadrate, mc_boxheadrate, mc_dropoutrate, mc_dropoutsamp = 0.1, 0.0, 0.0, 4

    class Net:
        def __init__(self):
            self.k = 0

        def __call__(self, images, training=False):
            self.k += 1
            return ([np.full((2, 1, 1, 9), self.k, np.float32)] * 5, [np.full((2, 1, 1, 36), -self.k, np.float32)] * 5)

    cls, box = utils_extra.mc_eval(Net(), None, Cfg())
    assert cls[0].shape == (4, 2, 1, 1, 9) and box[0].shape == (2, 1, 1, 36)      # box head: the last iteration, unstacked
    assert box[0][0, 0, 0, 0] == -4
    mean, std = utils_extra.get_mcuncert(cls)
    np.testing.assert_allclose(mean[0], 2.5)
    np.testing.assert_allclose(std[0], np.sqrt(1.25), rtol=1e-6)


def test_visualize_draws_boxes_on_the_host():
    from uda_amd.visualize import visualize_image
    img = np.zeros((64, 96, 3), np.uint8)
    out = visualize_image(img, np.array([[8, 10, 40, 60], [5, 5, 6, 6]], np.float32), np.array([1, 3]),
                          np.array([0.9, 0.001], np.float32), "kitti", uncertainty=np.full((2, 4), 2.0, np.float32))
    assert out.shape == img.shape and out.dtype == np.uint8
    assert out[8, 10:60].any() and not out[50:, 70:].any()           # the confident box is drawn, the 0.1 % one is not
