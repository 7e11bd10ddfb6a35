"""GPU tests at BASELINE.json's full sizes (D0, 1280x768, T=10, batch 32): the oracle is too slow
for the whole batch, so the checks are size-independent properties plus an oracle spot check on
one image."""
import numpy as np
import pytest

from common import make_images, make_params, make_weights

pytestmark = pytest.mark.gpu

FULL = dict(image_size="1280x768", mc_dropout=True, mc_dropoutrate=0.05, mc_dropoutsamp=10, loss_attenuation=True)


def _driver(params, w, batch, **kw):
    from uda_amd.infer_lib import KerasDriver, ServingDriver
    return KerasDriver("_", False, params["name"], batch_size=batch, model_params=params, weights=w, **kw)


@pytest.fixture(scope="module")
def full_run():
    p = make_params(**FULL)
    w = make_weights(p, seed=0)
    imgs = make_images(32, 768, 1280, seed=2)
    d = _driver(p, w, 32, chunk_images=16)
    d.set_dropout_seed(5)
    det = d.serve(imgs)
    yield p, w, imgs, d, det
    d.close()


def test_output_layout_and_invariants(full_run):
    p, w, imgs, d, (boxes, scores, classes, valid, logits) = full_run
    assert boxes.shape == (32, 100, 12) and scores.shape == (32, 100) and classes.shape == (32, 100, 8)
    assert valid.dtype == np.int32 and logits.shape == (32, 100, 7)
    assert np.all(valid == 100)                              # random-init regime: every anchor is a candidate
    assert np.all(np.isfinite(boxes)) and np.all(np.isfinite(scores))
    assert np.all(np.diff(scores, axis=1) <= 0)              # soft-NMS emits non-increasing scores
    assert np.all(scores > 0.001) and np.all(scores < 1)
    b = boxes[..., :4]
    assert b.min() >= 0 and b[..., [0, 2]].max() <= 768 and b[..., [1, 3]].max() <= 1280   # clipped, scale 1
    assert np.all(b[..., 2] >= b[..., 0]) and np.all(b[..., 3] >= b[..., 1])
    assert np.all(boxes[..., 4:] >= 0)                       # aleatoric / epistemic std
    assert np.all(classes[..., 0] >= 1) and np.all(classes[..., 0] <= 7) and np.all(classes[..., 1:] >= 0)
    # MC dropout really varies the samples: epistemic std is non-zero almost everywhere
    assert (boxes[..., 8:] > 0).mean() > 0.99
    # chunk 16 of 32: the NMS grid of chunk 0 shares the device with the conv stack of chunk 1 and must still be
    # co-resident in time (no time-out -> no redo with two launches per epoch)
    assert d.nms_coop_fallbacks() == 0 and d.nms_prefix_fallbacks() == 0


def test_rerun_is_deterministic_and_seed_matters(full_run):
    p, w, imgs, d, det = full_run
    d.set_dropout_seed(5)
    again = d.serve(imgs)
    for a, b in zip(again, det):
        np.testing.assert_array_equal(a, b)
    d.set_dropout_seed(6)
    other = d.serve(imgs)
    assert not np.array_equal(other[0], det[0])


def test_chunking_and_batch_position_invariance(full_run):
    """Detections of an image do not depend on the chunk size or on where the image sits in the batch
    once the Philox image offset is given (the property image sharding relies on)."""
    p, w, imgs, d, det = full_run
    d2 = _driver(p, w, 8, chunk_images=3)
    d2.set_dropout_seed(5)
    d2.set_image_offset(16)
    part = d2.serve(imgs[16:24])
    d2.close()
    for a, b in zip(part, det):
        np.testing.assert_array_equal(a, b[16:24])


def test_postprocess_of_own_heads_reproduces_serve(full_run):
    p, w, imgs, d, det = full_run
    d.set_dropout_seed(5)
    d.serve(imgs[:4])
    cls, box = d.head_outputs(4)
    assert cls[0].shape == (10, 4, 96, 160, 63) and box[0].shape == (10, 4, 96, 160, 72)
    again = d.postprocess(cls, box, np.ones(4, np.float32))
    for a, b in zip(again, det):
        np.testing.assert_array_equal(a, b[:4])


def test_unit_masks_equal_the_deterministic_network(full_run):
    """MC path with every keep-scale = 1 must equal the non-MC network (checks the shared / per-sample
    split of the plan at full size)."""
    p, w, imgs, d, det = full_run
    ones = {name: np.ones((2, 10, ch), np.float32) for name, ch, _ in d.plan.sites}
    d.set_dropout_masks(ones)
    d.serve(imgs[:2])
    cls_mc, box_mc = d.head_outputs(2)
    d._injected = False
    p0 = make_params(image_size="1280x768", loss_attenuation=True)
    d0 = _driver(p0, w, 2)
    d0.serve(imgs[:2])
    cls, box = d0.head_outputs(2)
    d0.close()
    for l in range(5):
        for t in (0, 9):
            np.testing.assert_array_equal(cls_mc[l][t], cls[l])
            np.testing.assert_array_equal(box_mc[l][t], box[l])


def test_oracle_spot_check_one_image_full_resolution(full_run):
    """One image at 1280x768 with T=2 against the CPU oracle: heads within f32 tolerance, and the whole
    post-process (184 140 near-tied candidates per image) bit-exact on the oracle's head outputs."""
    from oracle import effdet_ref as E, philox_ref as R, post_ref as P, preprocess_ref as PP
    _, w, imgs, _, _ = full_run
    p = make_params(**dict(FULL, mc_dropoutsamp=2))
    d = _driver(p, w, 1)
    d.set_dropout_seed(9)
    d.serve(imgs[:1])
    cls, box = d.head_outputs(1)
    x, scales = PP.preprocess(imgs[:1], (768, 1280), p["mean_rgb"], p["stddev_rgb"])
    masks = R.make_masks(E.dropout_sites(p), 9, 1, 2)
    rcls, rbox = E.forward(w, p, x, masks)
    from common import check_heads
    from uda_amd import plan as plan_mod
    # float32-class schemes (the default: two fp16 pieces; three bf16 pieces; exact f32): the bars the 192 x 128 test
    # holds them to (test_gpu_round3: 2e-5 x max / 1e-5 relative RMS per channel group), at FULL size against the float32
    # CPU oracle - measured 2.4e-7 relative RMS.  Two bf16 pieces (UDA_PW_SCHEME=bf16x2) is the narrow scheme: 2e-4 / 1e-4.
    tight = plan_mod.pw_scheme() != "bf16x2"
    check_heads(cls, rcls, tol=2e-5 if tight else 2e-4, tol_rms=5e-6 if tight else 1e-4)
    check_heads(box, rbox, tol=2e-5 if tight else 2e-4, tol_rms=5e-6 if tight else 1e-4)
    want = P.postprocess_global(p, rcls, rbox, scales)
    got = d.postprocess(rcls, rbox, scales)
    for g, r in zip(got, want):
        np.testing.assert_array_equal(g, r)
    d.close()


def test_full_size_serve_equals_the_oracle_chain_end_to_end(capsys):
    """serve() as a whole - uint8 image, network x T, aggregate, decode, soft-NMS - against the oracle chain
    post_ref.postprocess_global(effdet_ref.forward(preprocess_ref(...))) on ONE image at 1280 x 768 (the reference runs batch 1,
    validate_model.py:476-522), T = 2, with spread scores (class-predict kernel x 5: a score range instead of
    184 140 near-ties).  Soft-NMS keeps one of several overlapping anchors, so the statement is margin-aware like the
    scheme-vs-scheme test below: every oracle detection whose selection margin (gap to the best overlapping runner-up on
    fully updated scores) exceeds twice the measured perturbation of those scores - its own and every earlier one - must
    come back as the SAME anchor with class equal and score / box / both sigmas within 1e-4; a detection that differs must
    be explained by such a margin."""
    from oracle import effdet_ref as E, philox_ref as R, post_ref as P, preprocess_ref as PP
    from uda_amd import plan as plan_mod
    p = make_params(**dict(FULL, mc_dropoutsamp=2))
    w = make_weights(p, seed=0, cls_spread=5.0)       # (x 20 saturates the best scores of a full-size map at 1 - a few ulps: ties again)
    imgs = make_images(1, 768, 1280, seed=7)
    d = _driver(p, w, 1)
    d.set_dropout_seed(9)
    det = d.serve(imgs)
    cand = d.candidates(1)                       # boxes [1,K,4], scores, classes, u_cls, u_al, u_ep of the device run
    d.close()
    x, scales = PP.preprocess(imgs, (768, 1280), p["mean_rgb"], p["stddev_rgb"])
    masks = R.make_masks(E.dropout_sites(p), 9, 1, 2)
    rcls, rbox = E.forward(w, p, x, masks)
    ref = P.pre_nms(p, rcls, rbox)
    want = P.postprocess_global(p, rcls, rbox, scales)
    sigma2, iou_thr, score_thr = P.nms_params(p)
    M = p["nms_configs"]["max_output_size"]
    keep_ref = P.nms_v5(ref["boxes"][0], ref["scores"][0], M, iou_thr, score_thr, sigma2, True)[0]
    keep_dev = P.nms_v5(cand["boxes"][0], cand["scores"][0], M, iou_thr, score_thr, sigma2, True)[0]      # (the device's own keep set, re-derived from its candidates)
    tol = 1e-4 if plan_mod.pw_scheme() != "bf16x2" else 1e-3
    # candidates: every one of the 184 140 within the tolerance
    assert (cand["classes"][0] != ref["classes"][0]).mean() < 1e-4
    np.testing.assert_allclose(cand["scores"][0], ref["scores"][0], rtol=tol, atol=1e-7)
    top = 40
    sel = keep_ref[:top]
    margin, pert, runner = _selection_margins(ref["boxes"][0], ref["scores"][0], cand["boxes"][0], cand["scores"][0], sel, -0.5 / sigma2)
    decided = margin > 2.0 * pert
    rows_dev = {int(a): r for r, a in enumerate(keep_dev)}
    same = n_checked = 0
    for k in range(top):
        a = int(sel[k])
        if a in rows_dev:
            same += 1
        if not decided[:k + 1].all():
            continue                             # this or an earlier selection was within reach of the perturbation
        assert a in rows_dev and rows_dev[a] == k, ("a decided detection moved", k, a, margin[:k + 1], pert[:k + 1])
        n_checked += 1
        g_b, r_b = det[0][0, k], want[0][0, k]
        assert det[2][0, k, 0] == want[2][0, k, 0]
        np.testing.assert_allclose(det[1][0, k], want[1][0, k], rtol=tol, atol=0)
        scale = max(r_b[2] - r_b[0], r_b[3] - r_b[1], 1.0)
        assert np.abs(g_b[:4] - r_b[:4]).max() <= tol * scale, (k, g_b, r_b)
        assert (np.abs(g_b[4:] - r_b[4:]) <= tol * np.maximum(r_b[4:], 1e-2 * scale)).all(), (k, g_b, r_b)
    with capsys.disabled():
        print("\n[full-size end-to-end vs oracle] top %d oracle detections: %d same anchors, %d decided by margins > 2 x perturbation "
              "(all within %.0e); margins median %.2e, perturbation median %.2e" % (top, same, n_checked, tol, float(np.median(margin)), float(np.median(pert))))
    # (random-init weights lose the image in the depth of the network: the best anchors sit in a plateau of bit-identical
    # scores whatever the input - tools/debug/score_probe.py: 90-140 distinct values among the top 1000 for any spread and any image -
    # so most selections are decided by the index tie-break on BOTH sides, margin 0: they count as "same", few as "decided")
    assert int(keep_dev[0]) == int(sel[0]) and n_checked >= 1 and same >= 30
    assert det[3][0] == want[3][0]


PRECISION_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
FULL = dict(image_size="1280x768", mc_dropout=True, mc_dropoutrate=0.05, mc_dropoutsamp=10, loss_attenuation=True)
p = make_params(**FULL)
w = make_weights(p, seed=0, cls_spread=20.0)          # spread-out scores: a few confident clusters, as a trained head gives
d = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
d.set_dropout_seed(5)
det = d.serve(make_images(2, 768, 1280, seed=2))
c = d.candidates(2)
cls, box = d.head_outputs(2)
np.savez(sys.argv[1], b=det[0], s=det[1], c=det[2], v=det[3], cb=c["boxes"], cs=c["scores"], cc=c["classes"], ual=c["u_al"],
         uep=c["u_ep"], ucls=c["u_cls"], h_cls=cls[0], h_box=box[0])
d.close()
print("saved")
"""


def _iou_1n(box, boxes):
    """NonMaxSuppressionV5's IoU (float32, corners normalised, empty boxes -> 0) of one box against many."""
    f = np.float32
    y0, x0 = np.minimum(boxes[:, 0], boxes[:, 2]), np.minimum(boxes[:, 1], boxes[:, 3])
    y1, x1 = np.maximum(boxes[:, 0], boxes[:, 2]), np.maximum(boxes[:, 1], boxes[:, 3])
    by0, bx0, by1, bx1 = min(box[0], box[2]), min(box[1], box[3]), max(box[0], box[2]), max(box[1], box[3])
    area = (y1 - y0) * (x1 - x0)
    barea = f((by1 - by0) * (bx1 - bx0))
    ih = np.maximum(np.minimum(y1, by1) - np.maximum(y0, by0), f(0))
    iw = np.maximum(np.minimum(x1, bx1) - np.maximum(x0, bx0), f(0))
    inter = ih * iw
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / (area + barea - inter)
    return np.where((area > 0) & (barea > 0), iou, f(0)).astype(f)


def _selection_margins(ex_boxes, ex_scores, te_boxes, te_scores, sel, scale):
    """Replays the exact run's selection sequence `sel` on BOTH candidate sets with fully updated scores (the score a
    candidate has when NonMaxSuppressionV5 compares it: its stale score times the weights of every box selected so far,
    oracle/post_ref.py:239-278).  Per epoch k returns
      margin[k]  exact run: score of the selected candidate minus the best score of any other live candidate that
                 OVERLAPS it (IoU > 0: every candidate whose selection before / after it would change its output score - a
                 superset of the same-object runner-ups with IoU > 0.5),
      pert[k]    the largest difference between the two runs' updated scores of any candidate at that epoch (measured, it
                 includes what perturbed boxes do to the decay weights),
      runner[k]  the index of that best overlapping runner-up (-1: none)."""
    f = np.float32
    cur_e, cur_t = ex_scores.astype(f).copy(), te_scores.astype(f).copy()
    live = np.ones(cur_e.shape, bool)
    margin, pert, runner = [], [], []
    for i in sel:
        iou_e = _iou_1n(ex_boxes[i], ex_boxes)
        iou_t = _iou_1n(te_boxes[i], te_boxes)
        live[i] = False
        over = live & (iou_e > 0)
        if over.any():
            j = int(np.argmax(np.where(over, cur_e, -np.inf)))
            margin.append(float(cur_e[i] - cur_e[j]))
            runner.append(j)
        else:
            margin.append(np.inf)
            runner.append(-1)
        pert.append(float(np.abs(cur_t - cur_e)[live | (np.arange(live.size) == i)].max()))
        cur_e = np.where(live, cur_e * np.exp(np.float64(scale) * iou_e.astype(np.float64) ** 2).astype(f), cur_e)
        cur_t = np.where(live, cur_t * np.exp(np.float64(scale) * iou_t.astype(np.float64) ** 2).astype(f), cur_t)
    return np.array(margin), np.array(pert), np.array(runner)


SCHEMES = {"bf16x2": (1e-3, 1e-4, 2e-4), "bf16x3": (1e-4, 2e-6, 1e-5), "f16x2": (1e-4, 2e-6, 1e-5)}   # tol, head rel. RMS, head max
# of the top 20 detections of an image (measured, round 4: f16x2 20 / 20 on both images, bf16x2 20 / 20, bf16x3 20 / 20 and 12 / 10 -
# one near-tie (margins of 1e-7 against perturbations of 2e-8) flips in the six-term run of image 1 and takes its neighbours along)
MIN_QUALIFY = {"bf16x2": 8, "bf16x3": 8, "f16x2": 12}      # same anchor AND decayed by the same earlier selections
MIN_SAME = {"bf16x2": 10, "bf16x3": 10, "f16x2": 15}       # same anchors kept


def test_split_products_stay_within_north_star_of_exact_f32_at_full_size(tmp_path, capsys):
    """Every split scheme of the 1x1 contractions against the exact f32-input MFMA path (UDA_PW_SCHEME=f32) on the SAME
    full-size batch: what reaches the caller - scores, boxes, aleatoric / epistemic sigma of every one of the 184 140
    candidates per image, and the final detections - stays within north_star's 1e-3 for two bf16 pieces (three cross
    terms), and ten times tighter (1e-4; heads 2e-6 relative RMS / 1e-5 max) for the float32-class schemes: three bf16
    pieces (six terms) and two fp16 pieces (the shipped default).  The switch is read once per process, so each scheme
    runs in its own interpreter.
    Detections (postprocess.py:392-413): soft-NMS is discontinuous in the candidates - it keeps ONE of several overlapping
    anchors whose scores differ in the 5th digit - so the detection-level statement is margin-aware: for every detection of
    the exact run that the other run represents by the SAME ANCHOR, decayed by the same earlier selections, score, box and
    both sigmas agree within the tolerance; and a detection the other run represents by a different anchor must be explained
    by a selection margin (gap to the best overlapping runner-up, computed on fully updated scores) that twice the MEASURED
    perturbation of those scores can cross - its own or an earlier one."""
    import os
    import subprocess
    import sys
    from common import ROOT
    from oracle import post_ref as P
    runs = {}
    for scheme in ["f32"] + list(SCHEMES):
        out = str(tmp_path / ("t_%s.npz" % scheme))
        e = dict(os.environ, UDA_PW_SCHEME=scheme)
        e.pop("UDA_PW_TERMS", None)
        r = subprocess.run([sys.executable, "-c", PRECISION_WORKER % {"root": ROOT}, out], cwd=ROOT, env=e, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "saved" in r.stdout, (scheme, r.stdout[-1500:], r.stderr[-1500:])
        runs[scheme] = dict(np.load(out))
    exact = runs["f32"]
    p = make_params(**FULL)
    soft_sigma, iou_thr, score_thr = P.nms_params(p)
    M = p["nms_configs"]["max_output_size"] or 100
    # the exact run's keep list per image, from the oracle's NonMaxSuppressionV5 on the device's own candidates; it must
    # reproduce the device's detections (ties the anchor indices to what the caller got)
    keep = {}
    for name, run in runs.items():
        keep[name] = []
        for n in range(2):
            idx, sc, valid = P.nms_v5(run["cb"][n], run["cs"][n], M, iou_thr, score_thr, soft_sigma, True)
            np.testing.assert_array_equal(sc, run["s"][n])
            assert valid == run["v"][n]
            keep[name].append(idx)
    report, counts = [], []
    for scheme, (tol, head_rms, head_max) in SCHEMES.items():
        got = runs[scheme]
        # --- head outputs, per channel group (box deltas | sigma share a tensor: judge each group on its own scale)
        for key, groups in (("h_cls", [(0, 63)]), ("h_box", [(0, 36), (36, 72)])):
            for lo, hi in groups:
                g, r = got[key][..., lo:hi].astype(np.float64), exact[key][..., lo:hi].astype(np.float64)
                rel_rms = np.sqrt(np.mean((g - r) ** 2)) / np.sqrt(np.mean(r * r))
                assert rel_rms <= head_rms, (scheme, key, lo, rel_rms)
                assert np.abs(g - r).max() <= head_max * np.abs(r).max(), (scheme, key, lo)
        # --- every candidate (index-aligned: the argmax path keeps all anchors)
        same_cls = got["cc"] == exact["cc"]
        assert same_cls.mean() > 0.999                                  # an argmax may flip only between near-tied classes
        assert np.abs(got["cs"] - exact["cs"]).max() <= tol * exact["cs"].max()
        box_scale = np.maximum(exact["cb"][..., 2] - exact["cb"][..., 0], exact["cb"][..., 3] - exact["cb"][..., 1])[..., None]
        assert (np.abs(got["cb"] - exact["cb"]) <= tol * np.maximum(box_scale, 1.0)).all()
        for key in ("ual", "uep"):
            d = np.abs(got[key] - exact[key])
            assert (d <= tol * np.maximum(exact[key], 1e-2 * np.maximum(box_scale, 1.0))).mean() > 0.999, (scheme, key)
            assert np.sqrt(np.mean(d ** 2)) <= tol * np.sqrt(np.mean(exact[key] ** 2)), (scheme, key)
        # --- final detections, margin-aware
        np.testing.assert_array_equal(got["v"], exact["v"])
        for n in range(2):
            top = 20
            eb_all, gb_all = exact["cb"][n], got["cb"][n]
            sel = keep["f32"][n][:top]
            margin, pert, runner = _selection_margins(eb_all, exact["cs"][n], gb_all, got["cs"][n], sel, -0.5 / soft_sigma)
            decided = margin > 2.0 * pert
            rows_other = {int(a): r for r, a in enumerate(keep[scheme][n])}
            # overlapping predecessors of a kept anchor in a run: the selected boxes that decayed its score
            def preds(order, pos, boxes):
                a = int(order[pos])
                ov = _iou_1n(boxes[a], boxes[order[:pos]]) > 0 if pos else np.zeros(0, bool)
                return frozenset(int(x) for x in order[:pos][ov])
            n_q = 0
            for k in range(top):
                a = int(sel[k])
                if a not in rows_other:
                    continue
                r = rows_other[a]
                if preds(keep["f32"][n], k, eb_all) != preds(keep[scheme][n], r, gb_all):
                    continue
                # same anchor, decayed by the same earlier selections: everything the caller gets for it must agree
                n_q += 1
                eb, gb = exact["b"][n, k], got["b"][n, r]
                assert got["c"][n, r, 0] == exact["c"][n, k, 0]
                np.testing.assert_allclose(got["s"][n, r], exact["s"][n, k], rtol=tol, atol=0, err_msg=str((scheme, n, k)))
                scale = max(eb[2] - eb[0], eb[3] - eb[1], 1.0)
                assert np.abs(gb[:4] - eb[:4]).max() <= tol * scale, (scheme, n, k)
                assert (np.abs(gb[4:] - eb[4:]) <= tol * np.maximum(eb[4:], 1e-2 * scale)).all(), (scheme, n, k)
            same = sum(int(a) in rows_other for a in sel)
            lost = [k for k in range(top) if int(sel[k]) not in rows_other]
            # a detection that is NOT the same anchor in the other run must be explained by a margin the perturbation can
            # cross: its own, or that of an earlier selection (any of them can change what decays it)
            for k in lost:
                assert (~decided[:k + 1]).any(), (scheme, n, k, margin[:k + 1], pert[:k + 1])
            report.append("%s image %d: of the top %d detections %d are the same anchors, %d of them decayed by the same earlier "
                          "selections (all within %.0e on score / box / sigma); selection margins: median %.2e, measured perturbation "
                          "of the updated scores: median %.2e, %d margins above 2 x perturbation" % (
                              scheme, n, top, same, n_q, tol, float(np.median(margin)), float(np.median(pert)), int(decided.sum())))
            counts.append((scheme, n, n_q, same))
            assert int(keep[scheme][n][0]) == int(sel[0])            # the best detection of an image is the same anchor
    with capsys.disabled():
        print()
        for line in report:
            print("[margin-aware detection parity] " + line)
    # the statement must not be vacuous: a good share of the confident detections is decided by margins above the perturbation
    for scheme, n, q, same in counts:
        assert q >= MIN_QUALIFY[scheme], (scheme, n, q)
        assert same >= MIN_SAME[scheme], (scheme, n, same)
