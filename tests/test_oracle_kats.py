"""Hand-derived known-answer tests that pin the oracle's restatement of the TF semantics
(SURVEY §9).  TF itself cannot be imported here (parity unpinned against TF); these KATs
anchor every rule the oracle encodes to a value computed by hand / by closed form."""
import math

import numpy as np
import pytest
import torch

from common import FULL_MC, HEAD_MC, make_params, make_weights
from oracle import effdet_ref as E, post_ref as P, preprocess_ref as PP

F32 = np.float32


# ------------------------------------------------------------------ anchors (anchors.py:138-218)
def test_anchor_table_small_image():
    p = make_params(image_size=64)
    a = P.anchor_boxes(p)
    assert a.shape == ((64 + 16 + 4 + 1 + 1) * 9, 4) and a.dtype == np.float32
    # level 3: stride 8, first centre (4, 4); octave 0, aspect 1: half size 4*8/2 = 16
    np.testing.assert_allclose(a[0], [-12, -12, 20, 20], rtol=0, atol=0)
    # aspect 2.0: half_x = 32*sqrt(2)/2, half_y = 32/sqrt(2)/2
    hx, hy = 32 * math.sqrt(2) / 2, 32 / math.sqrt(2) / 2
    np.testing.assert_allclose(a[1], [4 - hy, 4 - hx, 4 + hy, 4 + hx], rtol=1e-7)
    # aspect 0.5 swaps them; octave 1 scales by 2^(1/3)
    np.testing.assert_allclose(a[2], [4 - hx, 4 - hy, 4 + hx, 4 + hy], rtol=1e-7)
    h1 = 32 * 2 ** (1 / 3) / 2
    np.testing.assert_allclose(a[3], [4 - h1, 4 - h1, 4 + h1, 4 + h1], rtol=1e-7)
    # second location of level 3 is one stride to the right (x fastest)
    np.testing.assert_allclose(a[9], [-12, -4, 20, 28])
    # level 7 (1x1, stride 64): centre (32, 32), half 4*64/2 = 128
    np.testing.assert_allclose(a[-9], [32 - 128, 32 - 128, 32 + 128, 32 + 128])


def test_anchor_count_kitti_resolution():
    p = make_params(image_size="1280x768")
    assert P.feat_sizes("1280x768", 7)[3:] == [(96, 160), (48, 80), (24, 40), (12, 20), (6, 10)]
    assert P.anchor_boxes(p).shape[0] == 184140


# ------------------------------------------------------------------ decode (anchors.py:41-75, utils_box.py:140-160)
def test_decode_plain_identity_and_scaling():
    anc = np.array([[10, 20, 30, 60]], np.float32)
    np.testing.assert_array_equal(P.decode_box_outputs(np.zeros((1, 4), np.float32), anc), anc)
    out = P.decode_box_outputs(np.array([[0.5, -0.25, math.log(2), 0]], np.float32), anc)
    # ha=20, wa=40, centre (20,40): yc = .5*20+20 = 30, xc = -.25*40+40 = 30, h = 40, w = 40
    np.testing.assert_allclose(out, [[10, 10, 50, 50]], rtol=1e-6)


def test_decode_uncert_lnorm_closed_form():
    anc = np.array([[0, 0, 10, 20]], np.float32)
    t = np.array([[0.1, -0.2, 0.3, -0.1]], np.float32)
    s = np.array([[0.5, 0.4, 0.3, 0.2]], np.float32)
    box, sig = P.decode_uncert(t, s, anc, "l-norm")
    ty, tx, th, tw = [float(F32(v)) for v in (0.1, -0.2, 0.3, -0.1)]
    vy, vx, vh, vw = [float(F32(v)) ** 2 for v in (0.5, 0.4, 0.3, 0.2)]
    ha, wa, ya, xa = 10.0, 20.0, 5.0, 10.0
    h, w = math.exp(th + vh / 2) * ha, math.exp(tw + vw / 2) * wa
    yc, xc = ty * ha + ya, tx * wa + xa
    var_h = (math.exp(vh) - 1) * math.exp(2 * th + vh) * ha ** 2
    var_w = (math.exp(vw) - 1) * math.exp(2 * tw + vw) * wa ** 2
    sy, sx = math.sqrt(vy * ha ** 2 + var_h / 4), math.sqrt(vx * wa ** 2 + var_w / 4)
    np.testing.assert_allclose(box[0], [yc - h / 2, xc - w / 2, yc + h / 2, xc + w / 2], rtol=1e-6)
    np.testing.assert_allclose(sig[0], [sy, sx, sy, sx], rtol=1e-6)
    # zero predicted sigma -> plain decode, zero uncertainty
    b0, s0 = P.decode_uncert(t, np.zeros_like(s), anc, "l-norm")
    np.testing.assert_allclose(b0, P.decode_box_outputs(t, anc), rtol=1e-6)
    assert np.all(s0 == 0)
    assert P.decode_uncert(t, s, anc, "n-flow")[0].tolist() == box.tolist()


# ------------------------------------------------------------------ NMSV5
B0, B1, B2 = [0, 0, 10, 10], [0, 0, 10, 9], [20, 20, 30, 30]      # iou(B0, B1) = 0.9


@pytest.mark.parametrize("impl", [P.nms_v5, P.nms_v5_py])
def test_nms_hard_toy(impl):
    boxes = np.array([B0, B1, B2], np.float32)
    idx, sc, valid = impl(boxes, np.array([.9, .8, .7], np.float32), 5, 0.5, float("-inf"), 0.0, True)
    assert valid == 2 and idx.tolist() == [0, 2, 0, 0, 0]
    np.testing.assert_array_equal(sc, np.array([.9, .7, 0, 0, 0], np.float32))
    idx, sc, valid = impl(boxes, np.array([.9, .8, .7], np.float32), 5, 0.95, float("-inf"), 0.0, False)
    assert valid == 3 and idx.tolist() == [0, 1, 2]          # iou .9 <= .95: nothing suppressed


@pytest.mark.parametrize("impl", [P.nms_v5, P.nms_v5_py])
def test_nms_soft_toy(impl):
    boxes = np.array([B0, B1, B2], np.float32)
    scores = np.array([.9, .8, .7], np.float32)
    idx, sc, valid = impl(boxes, scores, 3, 0.5, 0.001, 0.25, True)
    assert valid == 3 and idx.tolist() == [0, 2, 1]
    iou = F32(F32(90) / F32(F32(100) + F32(90) - F32(90)))
    w = F32(math.exp(float(F32(F32(F32(-2.0) * iou) * iou))))
    np.testing.assert_array_equal(sc, np.array([.9, .7, F32(.8) * w], np.float32))
    # the decayed score falls under the threshold -> dropped, output padded with index 0 / score 0
    idx, sc, valid = impl(boxes, np.array([.9, .004, .7], np.float32), 3, 0.5, 0.001, 0.25, True)
    assert valid == 2 and idx.tolist() == [0, 2, 0] and sc[2] == 0
    # max_output_size truncates
    idx, sc, valid = impl(boxes, scores, 1, 0.5, 0.001, 0.25, True)
    assert valid == 1 and idx.tolist() == [0]


@pytest.mark.parametrize("impl", [P.nms_v5, P.nms_v5_py])
def test_nms_ties_and_threshold(impl):
    boxes = np.array([[0, 0, 1, 1], [5, 5, 6, 6], [9, 9, 10, 10], [20, 20, 21, 21]], np.float32)
    scores = np.array([.5, .7, .7, .0005], np.float32)
    idx, sc, valid = impl(boxes, scores, 4, 0.5, 0.001, 0.25, True)
    assert valid == 3 and idx.tolist() == [1, 2, 0, 0]        # ties -> smaller index; 0.0005 never a candidate
    # degenerate (zero-area) boxes have IoU 0 with everything
    boxes = np.array([[0, 0, 10, 10], [5, 5, 5, 9], [0, 0, 10, 10]], np.float32)
    idx, sc, valid = impl(boxes, np.array([.9, .8, .7], np.float32), 3, 0.5, float("-inf"), 0.0, True)
    assert idx.tolist() == [0, 1, 0] and valid == 2


def test_nms_lazy_chain_order():
    """A candidate overlapping two selected boxes is rescored newest-selected first."""
    boxes = np.array([[0, 0, 10, 10], [0, 20, 10, 30], [0, 6, 10, 24]], np.float32)
    scores = np.array([.9, .8, .7], np.float32)
    idx, sc, valid = P.nms_v5(boxes, scores, 3, 0.5, 0.001, 0.25, True)
    assert idx.tolist() == [0, 1, 2]
    iou = lambda a, b: F32(P._lib().oracle_iou(boxes[a].ctypes.data, boxes[b].ctypes.data))
    wgt = lambda s: F32(math.exp(float(F32(F32(F32(-2.0) * s) * s))))
    want = F32(F32(F32(.7) * wgt(iou(2, 1))) * wgt(iou(2, 0)))
    assert sc[2] == want


def test_nms_c_and_python_twins_agree():
    rng = np.random.default_rng(3)
    for tied in (False, True):
        c = rng.uniform(0, 100, (300, 2))
        wh = rng.uniform(2, 40, (300, 2))
        boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        s = (rng.uniform(0, 1, 300) if not tied else 0.01 + rng.normal(0, 1e-4, 300)).astype(np.float32)
        for sigma, thr in ((0.25, 0.001), (0.0, float("-inf"))):
            a = P.nms_v5(boxes, s, 50, 0.5, thr, sigma, True)
            b = P.nms_v5_py(boxes, s, 50, 0.5, thr, sigma, True)
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x, y)


# ------------------------------------------------------------------ TF op semantics
def test_same_padding_is_bottom_right_heavy():
    x = torch.arange(16, dtype=torch.float32).reshape(1, 1, 4, 4)
    k = np.ones((3, 3, 1, 1), np.float32)
    y = E.conv2d(x, k, stride=2)[0, 0].numpy()
    xs = x[0, 0].numpy()
    # in=4,k=3,s=2: out=2, pad_total=1 -> 0 before, 1 after
    np.testing.assert_array_equal(y, [[xs[0:3, 0:3].sum(), xs[0:3, 2:4].sum()],
                                      [xs[2:4, 0:3].sum(), xs[2:4, 2:4].sum()]])
    d = E.depthwise(x, k, stride=1)[0, 0].numpy()          # stride 1: symmetric 1/1
    assert d[0, 0] == xs[0:2, 0:2].sum() and d[3, 3] == xs[2:4, 2:4].sum()


def test_max_pool_same_and_nearest_upsample():
    x = -torch.arange(25, dtype=torch.float32).reshape(1, 1, 5, 5)     # all negative: padding must never win
    y = E.max_pool_same(x, 3, 2)[0, 0].numpy()
    xs = x[0, 0].numpy()
    assert y.shape == (3, 3)                                            # 5 -> 3, pad 1 before / 1 after
    assert y[0, 0] == xs[0:2, 0:2].max() and y[1, 1] == xs[1:4, 1:4].max() and y[2, 2] == xs[3:5, 3:5].max()
    u = E.nearest_upsample(torch.arange(3, dtype=torch.float32).reshape(1, 1, 1, 3), 1, 5)
    assert u[0, 0, 0].tolist() == [0, 0, 1, 1, 2]                       # floor(dst * 3/5)
    u2 = E.nearest_upsample(torch.arange(4, dtype=torch.float32).reshape(1, 1, 2, 2), 4, 4)
    assert u2[0, 0].tolist() == [[0, 0, 1, 1], [0, 0, 1, 1], [2, 2, 3, 3], [2, 2, 3, 3]]


def test_batch_norm_and_swish_formulas():
    w = {"bn/gamma": np.array([2.0], np.float32), "bn/beta": np.array([0.5], np.float32),
         "bn/moving_mean": np.array([1.0], np.float32), "bn/moving_variance": np.array([3.0], np.float32)}
    x = torch.tensor([[[[4.0]]]])
    np.testing.assert_allclose(E.batch_norm(x, w, "bn").item(), 2.0 * (4.0 - 1.0) / math.sqrt(3.0 + 1e-3) + 0.5, rtol=1e-6)
    np.testing.assert_allclose(E.swish(torch.tensor([1.5])).item(), 1.5 / (1 + math.exp(-1.5)), rtol=1e-6)


def test_bilinear_resize_half_pixel_centres():
    img = np.array([[0, 10], [20, 30]], np.float32)[..., None]
    out = PP.resize_bilinear(img, 4, 4)[..., 0]
    # src coords -0.25, 0.25, 0.75, 1.25 -> weights 0, .25, .75, 1 between the two samples
    r0 = [0, 2.5, 7.5, 10]
    np.testing.assert_allclose(out[0], r0)
    np.testing.assert_allclose(out[1], [v + 5 for v in r0])
    np.testing.assert_allclose(out[3], [v + 20 for v in r0])


def test_preprocess_scale_pad_and_normalise():
    p = make_params(image_size="192x128")
    img = np.full((1, 100, 180, 3), 128, np.uint8)
    x, s = PP.preprocess(img, (128, 192), p["mean_rgb"], p["stddev_rgb"])
    scale = min(F32(128) / F32(100), F32(192) / F32(180))
    assert s[0] == F32(1) / scale
    sh, sw = int(F32(100) * scale), int(F32(180) * scale)
    assert (sh, sw) == (106, 192)
    want = (F32(128) - F32(p["mean_rgb"][0])) / F32(p["stddev_rgb"][0])
    np.testing.assert_allclose(x[0, :sh, :sw, 0], want, rtol=1e-6)
    assert np.all(x[0, sh:] == 0)
    # BDD-like: no resample needed, only bottom padding
    x2, s2 = PP.preprocess(np.zeros((1, 120, 192, 3), np.uint8), (128, 192), p["mean_rgb"], p["stddev_rgb"])
    assert s2[0] == 1.0 and np.all(x2[0, 120:] == 0) and np.all(x2[0, :120] != 0)


# ------------------------------------------------------------------ MC aggregation (utils_extra.py:220-244)
def test_mc_mean_std_is_population_std():
    x = np.random.default_rng(0).normal(size=(10, 5, 3)).astype(np.float32)
    m, s = P.seq_mean_std(x)
    np.testing.assert_allclose(m, x.astype(np.float64).mean(0), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(s, x.astype(np.float64).std(0), rtol=1e-5, atol=1e-6)   # ddof = 0


# ------------------------------------------------------------------ network self-consistency
def test_mc_forward_with_unit_masks_equals_deterministic():
    p = make_params(image_size="64x64", **FULL_MC)
    w = make_weights(p)
    x = np.random.default_rng(1).normal(size=(2, 64, 64, 3)).astype(np.float32)
    sites = E.dropout_sites(p)
    assert len(sites) == 31 + 2 * 3 * 5                                  # 31 backbone sites in B0
    ones = {n: np.ones((2, 3, c), np.float32) for n, c, _ in sites}
    cls_mc, box_mc = E.forward(w, p, x, ones)
    cls, box = E.forward_once(w, p, x)
    assert cls_mc[0].shape == (3,) + cls[0].shape and box_mc[0].shape[-1] == 72
    for t in range(3):
        for l in range(5):
            np.testing.assert_array_equal(cls_mc[l][t], cls[l])
            np.testing.assert_array_equal(box_mc[l][t], box[l])


def test_head_only_dropout_leaves_backbone_rate_zero():
    p = make_params(image_size="64x64", **HEAD_MC)
    sites = E.dropout_sites(p)
    assert all(r == 0 for n, _, r in sites if n.startswith("blocks_"))
    assert all(abs(r - 0.05) < 1e-9 for n, _, r in sites if not n.startswith("blocks_"))


def test_postprocess_output_layout():
    p = make_params(image_size="64x64", **FULL_MC)
    w = make_weights(p, cls_spread=20.0)
    x = np.random.default_rng(2).normal(size=(2, 64, 64, 3)).astype(np.float32)
    from oracle import philox_ref as R
    masks = R.make_masks(E.dropout_sites(p), 5, 2, 3)
    cls, box = E.forward(w, p, x, masks)
    boxes, scores, classes, valid, logits = P.postprocess_global(p, cls, box, np.array([1.5, 2.0], np.float32))
    assert boxes.shape == (2, 100, 12) and scores.shape == (2, 100) and classes.shape == (2, 100, 8)
    assert valid.dtype == np.int32 and logits.shape == (2, 100, 7)
    assert np.all(np.diff(scores[0, :valid[0]]) <= 0)                    # soft-NMS emits non-increasing scores
    assert np.all(classes[..., 0] >= 1) and np.all(classes[..., 0] <= 7)
    assert boxes[..., :4].min() >= 0 and boxes[0, :, [0, 2]].max() <= 64 * 1.5 + 1e-3
    # per-class mode drops the uncertainty columns and never exceeds 100 rows
    b2, s2, c2, v2 = P.postprocess_per_class(p, cls, box, np.array([1.5, 2.0], np.float32))
    assert b2.shape == (2, 100, 4) and c2.shape == (2, 100) and np.all(v2 <= 100)
    assert np.all(np.diff(s2[0]) <= 0)


def test_softmax_entropy_unpack_kats():
    """stable softmax / entropy in bits / column unpacking (validate_model.py:159-202)."""
    from oracle import unpack_ref as U
    p, h = U.probab_entropy(np.array([[0.0, 0.0], [100.0, 0.0], [1.0, 2.0, 3.0][:2]], np.float32))
    np.testing.assert_allclose(p[0], [0.5, 0.5])
    np.testing.assert_allclose(h[0], 1.0, rtol=1e-6)               # one bit
    assert p[1][0] == 1.0 and abs(h[1]) < 1e-30 and np.isfinite(h[1])   # clamp at 1e-7 keeps 0 * log2 finite
    np.testing.assert_allclose(p[2], np.exp([1.0, 2.0]) / np.exp([1.0, 2.0]).sum(), rtol=1e-6)
    params = dict(mc_boxheadrate=0.0, mc_dropoutrate=0.05, mc_classheadrate=0.0, loss_attenuation=True)
    boxes = np.arange(2 * 3 * 12, dtype=np.float32).reshape(2, 3, 12)
    boxes[0, 0, 5] = np.nan
    classes = np.arange(2 * 3 * 8, dtype=np.float32).reshape(2, 3, 8)
    b4, cid, al, mc, mcc = U.unpack(params, boxes, classes)
    assert b4.shape == (2, 3, 4) and al.shape == (2, 3, 4) and mc.shape == (2, 3, 4) and mcc.shape == (2, 3, 7)
    assert al[0, 0, 1] == 0.0 and cid.shape == (2, 3)
    params["loss_attenuation"] = False
    b4, cid, al, mc, mcc = U.unpack(params, boxes[:, :, :8], classes)
    assert al is None and mc.shape == (2, 3, 4)


def test_isotonic_table_matches_sklearn():
    """the thresholds-table restatement of IsotonicRegression(out_of_bounds='clip').predict against the
    real sklearn class (what the reference pickles in calibrate_regression.py:370-434)."""
    sk = pytest.importorskip("sklearn.isotonic")
    from oracle import calib_ref as CR
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 5, 400)
    y = np.sqrt(x) + rng.normal(0, 0.2, 400)
    iso = sk.IsotonicRegression(increasing=True, out_of_bounds="clip").fit(x, y)
    q = np.concatenate([rng.uniform(-1, 7, 500), x[:20], [iso.X_thresholds_[0], iso.X_thresholds_[-1]]]).astype(np.float32)
    got = CR.iso_predict((iso.X_thresholds_, iso.y_thresholds_), q)
    np.testing.assert_allclose(got, iso.predict(q), rtol=1e-6, atol=1e-7)
    assert got.dtype == np.float32


def test_class_calibration_restatement_matches_sklearn_models():
    """SURVEY 8f.2, class half (utils_class.py:109-187): the oracle's temperature / isotonic class calibration against the
    reference's recipe executed with the REAL sklearn IsotonicRegression objects it pickles
    (calibrate_classification.py:53-70: IsotonicRegression(y_min=0, y_max=1, out_of_bounds="clip"))."""
    sk = pytest.importorskip("sklearn.isotonic")
    from oracle import calib_ref as CR, unpack_ref as U
    rng = np.random.default_rng(11)
    C = 7
    logits = rng.normal(0, 3, (60, C)).astype(np.float32)
    probs = U.stable_softmax(logits)
    onehot = (rng.uniform(size=probs.shape) < probs).astype(np.float64)
    iso_all = sk.IsotonicRegression(y_min=0, y_max=1, out_of_bounds="clip").fit(probs.flatten(), onehot.flatten())
    iso_cls = [sk.IsotonicRegression(y_min=0, y_max=1, out_of_bounds="clip").fit(probs[:, i], onehot[:, i]) for i in range(C)]
    tab = lambda m: (m.X_thresholds_, m.y_thresholds_)
    models = dict(ts_all=1.7, ts_percls=np.linspace(0.8, 2.4, C), iso_all=tab(iso_all), iso_percls=[tab(m) for m in iso_cls])
    test = rng.normal(0, 3, (40, C)).astype(np.float32)
    # the reference's recipe, literally, with the sklearn objects
    want = {}
    want["ts_all"] = U.stable_softmax(test / np.float32(1.7))
    want["ts_percls"] = U.stable_softmax(test / models["ts_percls"].astype(np.float32))
    p = U.stable_softmax(test)
    post = iso_all.predict(p.flatten()).reshape(p.shape)
    want["iso_all"] = post / np.stack([np.sum(post, axis=-1)] * C, axis=-1)
    post = np.stack([iso_cls[i].predict(p[:, i]) for i in range(C)], axis=1)
    want["iso_percls"] = post / np.stack([np.sum(post, axis=-1)] * C, axis=-1)
    for method, w in want.items():
        ent, prob = CR.perform_class_calib(method, models, test)
        np.testing.assert_allclose(prob, w, rtol=2e-6, atol=1e-7, err_msg=method)
        went = -np.sum(w * np.nan_to_num(np.log2(np.maximum(w, 10 ** -7))), axis=1)
        np.testing.assert_allclose(ent, went, rtol=1e-5, atol=1e-6, err_msg=method)
        np.testing.assert_allclose(prob.sum(1), 1.0, rtol=1e-5)
    # sampling leg: the Philox normal stream is standard normal, and the sampled calibration returns mean / std over the draws
    z = CR.philox_normal(7, np.arange(200000, dtype=np.uint32), np.uint32(3), np.uint32(1), 0x5A)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01 and abs(((z ** 3).mean())) < 0.03
    unc = np.abs(rng.normal(0, 0.3, test.shape)).astype(np.float32)
    ent, prob, su = CR.perform_class_calib("ts_all", models, test, unc, draws=10, seed=5)
    assert prob.shape == test.shape and su.shape == test.shape and (su > 0).all() and ent.shape == (40,)
    np.testing.assert_allclose(prob.sum(1), 1.0, rtol=1e-5)
