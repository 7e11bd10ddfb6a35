"""Oracle pinning: oracle/nms_np_ref.py against golden vectors produced by the REAL
reference module src/nms_np.py (tests/golden/make_nms_np_golden.py)."""
import os

import numpy as np
import pytest

from oracle import nms_np_ref as ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "nms_np_golden.npz"))
N_CASES, N_PC = [int(v) for v in G["n_cases"]]


@pytest.mark.parametrize("ci", range(N_CASES))
def test_nms_family_matches_reference(ci):
    dets = G["c%d_dets" % ci]
    np.testing.assert_array_equal(ref.hard_nms(dets.copy(), 0.5), G["c%d_hard" % ci])
    np.testing.assert_array_equal(ref.hard_nms(dets.copy(), 0.3), G["c%d_hard03" % ci])
    np.testing.assert_array_equal(ref.diou_nms(dets.copy(), 0.5), G["c%d_diou" % ci])
    cfgs = {"gauss": dict(method="gaussian", sigma=None, iou_thresh=None, score_thresh=None),
            "gauss2": dict(method="gaussian", sigma=0.25, iou_thresh=None, score_thresh=0.05),
            "linear": dict(method="linear", sigma=None, iou_thresh=0.3, score_thresh=0.01)}
    for tag, cfg in cfgs.items():
        got = ref.soft_nms(dets.copy(), cfg)
        want = G["c%d_%s" % (ci, tag)]
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-15)


@pytest.mark.parametrize("pi", range(N_PC))
def test_per_class_nms_matches_reference(pi):
    image_id, ncls, hard = [int(v) for v in G["p%d_meta" % pi]]
    cfg = dict(method="hard" if hard else "gaussian", sigma=None, iou_thresh=None, score_thresh=None)
    got = ref.per_class_nms(G["p%d_boxes" % pi].copy(), G["p%d_scores" % pi].copy(),
                            G["p%d_classes" % pi].copy(), np.array([image_id]),
                            np.array([1.25], dtype=np.float32), ncls, 100, cfg)
    want = G["p%d_det" % pi]
    assert got.dtype == np.float32 and got.shape == want.shape
    np.testing.assert_array_equal(got, want)
