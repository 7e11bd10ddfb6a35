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
    check_heads(cls, rcls)          # max-norm 2e-4 per level AND relative RMS 1e-4 per channel group (deltas | sigmas)
    check_heads(box, rbox)
    want = P.postprocess_global(p, rcls, rbox, scales)
    got = d.postprocess(rcls, rbox, scales)
    for g, r in zip(got, want):
        np.testing.assert_array_equal(g, r)
    d.close()


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


def _iou(a, b):
    y0, x0 = np.maximum(a[:, None, 0], b[None, :, 0]), np.maximum(a[:, None, 1], b[None, :, 1])
    y1, x1 = np.minimum(a[:, None, 2], b[None, :, 2]), np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(y1 - y0, 0, None) * np.clip(x1 - x0, 0, None)
    area = lambda z: (z[:, 2] - z[:, 0]) * (z[:, 3] - z[:, 1])
    return inter / (area(a)[:, None] + area(b)[None, :] - inter + 1e-9)


def test_three_term_products_stay_within_1e3_of_exact_f32_at_full_size(tmp_path):
    """The shipped contraction (split-bf16, 3 cross terms) against the exact f32-input MFMA path (UDA_PW_TERMS=0) and the
    six-term one on the SAME full-size batch: what reaches the caller - scores, boxes, aleatoric / epistemic sigma of
    every one of the 184 140 candidates per image, and the final detections - stays within north_star's 1e-3.
    The switch is read once per process, so each mode runs in its own interpreter."""
    import os
    import subprocess
    import sys
    from common import ROOT
    runs = {}
    for terms in ("3", "0", "6"):
        out = str(tmp_path / ("t%s.npz" % terms))
        e = dict(os.environ, UDA_PW_TERMS=terms)
        r = subprocess.run([sys.executable, "-c", PRECISION_WORKER % {"root": ROOT}, out], cwd=ROOT, env=e, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "saved" in r.stdout, (terms, r.stdout[-1500:], r.stderr[-1500:])
        runs[terms] = dict(np.load(out))
    exact = runs["0"]
    for terms in ("3", "6"):
        got = runs[terms]
        # three terms (shipped): north_star's 1e-3 on what reaches the caller, the 1e-4 / 2e-4 head bars; six terms
        # (float32-equivalent, every fusion kept): ten to fifty times tighter - measured at this size (tools/precision_probe.py):
        # heads 2e-8..2e-7 relative RMS, scores 1e-6, boxes 3e-6 of the box size
        tol = 1e-3 if terms == "3" else 1e-4
        head_rms, head_max = (1e-4, 2e-4) if terms == "3" else (2e-6, 1e-5)
        # --- head outputs, per channel group (box deltas | sigma share a tensor: judge each group on its own scale)
        for key, groups in (("h_cls", [(0, 63)]), ("h_box", [(0, 36), (36, 72)])):
            for lo, hi in groups:
                g, r = got[key][..., lo:hi].astype(np.float64), exact[key][..., lo:hi].astype(np.float64)
                rel_rms = np.sqrt(np.mean((g - r) ** 2)) / np.sqrt(np.mean(r * r))
                assert rel_rms <= head_rms, (terms, key, lo, rel_rms)
                assert np.abs(g - r).max() <= head_max * np.abs(r).max(), (terms, key, lo)
        # --- every candidate (index-aligned: the argmax path keeps all anchors)
        same_cls = got["cc"] == exact["cc"]
        assert same_cls.mean() > 0.999                                  # an argmax may flip only between near-tied classes
        assert np.abs(got["cs"] - exact["cs"]).max() <= tol * exact["cs"].max()
        box_scale = np.maximum(exact["cb"][..., 2] - exact["cb"][..., 0], exact["cb"][..., 3] - exact["cb"][..., 1])[..., None]
        assert (np.abs(got["cb"] - exact["cb"]) <= tol * np.maximum(box_scale, 1.0)).all()
        for key in ("ual", "uep"):
            d = np.abs(got[key] - exact[key])
            assert (d <= tol * np.maximum(exact[key], 1e-2 * np.maximum(box_scale, 1.0))).mean() > 0.999, (terms, key)
            assert np.sqrt(np.mean(d ** 2)) <= tol * np.sqrt(np.mean(exact[key] ** 2)), (terms, key)
        # --- final detections.  They are a deterministic function of the candidates (the post-process is bit-exact against the
        # oracle on identical head outputs), and soft-NMS is discontinuous in them: it keeps ONE of several overlapping
        # anchors whose scores differ in the 5th digit, so a 1e-7 perturbation of a candidate score can swap the anchor that
        # represents an object and with it the decay of its neighbours - even the six-term (float32-equivalent) run
        # differs from the exact one in which anchors it keeps.  What must hold: the same objects are found (a kept box of
        # the other run overlaps every confident detection), the best detection of an image is the same anchor, and
        # wherever the same anchor was kept its score / box / sigma agree within the tolerance.
        np.testing.assert_array_equal(got["v"], exact["v"])
        for n in range(2):
            k = 20
            iou = _iou(exact["b"][n, :k, :4], got["b"][n, :, :4])
            j = iou.argmax(1)
            assert (iou.max(1) > 0.3).mean() >= 0.8, (terms, n, iou.max(1))
            assert iou[0].max() > 0.98, (terms, n)
            ok = iou.max(1) > 0.98
            rows = np.nonzero(ok)[0]
            np.testing.assert_array_equal(got["c"][n, j[rows], 0], exact["c"][n, rows, 0])
            scale = np.maximum(exact["b"][n, rows, 2] - exact["b"][n, rows, 0], exact["b"][n, rows, 3] - exact["b"][n, rows, 1])[:, None]
            assert (np.abs(got["b"][n, j[rows], :4] - exact["b"][n, rows, :4]) <= tol * np.maximum(scale, 1.0)).all()
            sig = np.abs(got["b"][n, j[rows], 4:] - exact["b"][n, rows, 4:])
            assert (sig <= tol * np.maximum(exact["b"][n, rows, 4:], 1e-2 * scale)).mean() > 0.99
            # undecayed scores (the first detection of an image has no earlier selection to decay it)
            np.testing.assert_allclose(got["s"][n, 0], exact["s"][n, 0], rtol=tol)
