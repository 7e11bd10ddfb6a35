"""GPU parity tests for the paths BASELINE.json's configs reach that round 1 never executed
("configs_untested"):

  * the aggregate kernel's one-class-at-a-time branch (T=30, C=7: (T*C + 4T) * 256 B of LDS parking
    does not fit 48 KB -> park_all = false) and the LDS-parking kernel where the register kernel
    would otherwise run (UDA_AGG_REG=0 at T=10/C=7 and T=20/C=10);
  * `uncert_adjust_method="falsedec"` (reference src/utils_box.py:247-266);
  * EfficientDet-D2 at its full 1024x1024 resolution (configs[4]): heads against the oracle on one
    image, post-process bit-exact on the oracle's heads in global and per-class (max_nms_inputs=5000,
    reference src/eval.py:75) mode, and size-independent properties at the per-GPU share of
    configs[4] (2 images, T=30).

All calls go through the C ABI (ServingDriver -> ctypes -> libuda_hip.so)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from common import FULL_MC, LOSS_ATT, MC_NO_ATT, ROOT, make_images, make_params, make_weights

pytestmark = pytest.mark.gpu


def _driver(params, w, batch, **kw):
    from uda_amd.infer_lib import KerasDriver, ServingDriver
    return KerasDriver("_", False, params["name"], batch_size=batch, model_params=params, weights=w, **kw)


def _oracle_heads(p, w, imgs, hw, seed):
    from oracle import preprocess_ref as PP
    from common import oracle_heads
    x, scales = PP.preprocess(imgs, hw, p["mean_rgb"], p["stddev_rgb"])
    rcls, rbox = oracle_heads(p, w, x, seed)       # (more than 4 MC samples: 4 real network passes, the rest derived)
    return rcls, rbox, scales


def _assert_tuple_equal(got, want, what=""):
    assert len(got) == len(want), (what, len(got), len(want))
    for k, (g, r) in enumerate(zip(got, want)):
        assert g.shape == r.shape and g.dtype == r.dtype, (what, k, g.shape, r.shape, g.dtype, r.dtype)
        np.testing.assert_array_equal(g, r, err_msg="%s output %d" % (what, k))


# ------------------------------------------------------------------ aggregate kernel branches
@pytest.mark.parametrize("name,over,spread", [
    ("t30_c7", dict(FULL_MC, mc_dropoutsamp=30), 1.0),                   # configs[4]'s T and C: park_all = false
    ("t30_c7_spread", dict(FULL_MC, mc_dropoutsamp=30), 20.0),
    ("t30_c7_noatt", dict(MC_NO_ATT, mc_dropoutsamp=30), 20.0),          # plain decode per sample
    ("t40_c10", dict(FULL_MC, mc_dropoutsamp=40, num_classes=10), 1.0),
    ("t96_c3", dict(FULL_MC, mc_dropoutsamp=96, num_classes=3), 1.0)])   # the largest T the handle accepts
def test_postprocess_one_class_at_a_time_branch(name, over, spread):
    """T*C too large to park every class in LDS: the kernel re-reads one class at a time
    (csrc/kernels_post.hip aggregate_kernel, park_all == false)."""
    from oracle import post_ref as P
    p = make_params(**over)
    w = make_weights(p, seed=71, cls_spread=spread)
    rcls, rbox, scales = _oracle_heads(p, w, make_images(2, 100, 180, seed=72), (128, 192), 5)
    want = P.postprocess_global(p, rcls, rbox, scales)
    d = _driver(p, w, 2)
    _assert_tuple_equal(d.postprocess(rcls, rbox, scales), want, name)
    d.close()


AGG_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import FULL_MC, MC_NO_ATT, make_images, make_params, make_weights, oracle_heads
from uda_amd.infer_lib import KerasDriver, ServingDriver
from oracle import post_ref as P, preprocess_ref as PP
for over, spread in ((dict(FULL_MC, mc_dropoutsamp=10), 1.0), (dict(FULL_MC, mc_dropoutsamp=20, num_classes=10), 1.0),
                     (dict(FULL_MC, mc_dropoutsamp=10), 20.0), (dict(MC_NO_ATT, mc_dropoutsamp=10), 20.0),
                     (dict(FULL_MC, mc_dropoutsamp=30), 1.0), (dict(FULL_MC, mc_dropoutsamp=10, num_classes=10), 20.0),
                     (dict(MC_NO_ATT, mc_dropoutsamp=20), 1.0)):
    p = make_params(**over)
    w = make_weights(p, seed=11, cls_spread=spread)
    x, scales = PP.preprocess(make_images(2, 100, 180, seed=12), (128, 192), p["mean_rgb"], p["stddev_rgb"])
    rcls, rbox = oracle_heads(p, w, x, 21)
    want = P.postprocess_global(p, rcls, rbox, scales)
    d = KerasDriver("_", False, p["name"], batch_size=2, model_params=p, weights=w)
    got = d.postprocess(rcls, rbox, scales)
    assert len(got) == len(want)
    for g, r in zip(got, want):
        assert g.shape == r.shape and np.array_equal(g, r, equal_nan=True), (over, g.shape)
    wantc = P.postprocess_per_class(p, rcls, rbox, scales)
    gotc = d.postprocess(rcls, rbox, scales, post_mode="per_class")
    for g, r in zip(gotc, wantc):
        assert g.shape == r.shape and np.array_equal(g, r), ("per_class", over)
    d.close()
print("aggregate ok")
"""


@pytest.mark.parametrize("env", [dict(UDA_AGG_REG="0"), dict()], ids=["lds-parking", "registers"])
def test_aggregate_kernel_variants_bit_exact(env):
    """T=10/C=7, T=20/C=10, T=30/C=7 (BASELINE configs[1] / [2] / [4]) and neighbours through the LDS-parking aggregate kernel
    (UDA_AGG_REG=0: all classes parked, or one class at a time for T=30) and through the default choice (register kernels:
    all logits in registers for T=10/C=7, two streaming passes for the other common (T, C)); the switch is read once per
    process."""
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", AGG_WORKER % {"root": ROOT}], cwd=ROOT, env=e, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "aggregate ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


# ------------------------------------------------------------------ falsedec
@pytest.mark.parametrize("name,over", [("lossatt", LOSS_ATT), ("full_mc", FULL_MC),
                                       ("full_mc_t10", dict(FULL_MC, mc_dropoutsamp=10)),
                                       ("full_mc_t30", dict(FULL_MC, mc_dropoutsamp=30)),
                                       ("box_only", dict(mc_dropout=True, mc_boxheadrate=0.1, mc_dropoutsamp=3,
                                                         loss_attenuation=True))])
def test_falsedec_decode_bit_exact(name, over):
    """a13, method "falsedec" (src/utils_box.py:247-266): boxes decoded plainly, the 'uncertainty'
    decoded like a box - in float64 on the device, compared bit for bit."""
    from oracle import post_ref as P
    p = make_params(uncert_adjust_method="falsedec", **over)
    w = make_weights(p, seed=81, cls_spread=20.0)
    rcls, rbox, scales = _oracle_heads(p, w, make_images(2, 100, 180, seed=82), (128, 192), 9)
    want = P.postprocess_global(p, rcls, rbox, scales)
    d = _driver(p, w, 2)
    got = d.postprocess(rcls, rbox, scales)
    _assert_tuple_equal(got, want, name)
    # the aleatoric columns differ from the l-norm ones: the branch is really taken
    p2 = make_params(**over)
    d2 = _driver(p2, w, 2)
    other = d2.postprocess(rcls, rbox, scales)
    assert not np.array_equal(other[0][..., 4:8], got[0][..., 4:8])
    d.close()
    d2.close()


# ------------------------------------------------------------------ D2 at 1024 x 1024 (configs[4])
D2 = dict(model="efficientdet-d2", image_size="1024x1024", mc_dropout=True, mc_dropoutrate=0.05, loss_attenuation=True)
EVAL_NMS = dict(nms_configs=dict(method="gaussian", iou_thresh=None, score_thresh=0.0, sigma=None, pyfunc=False,
                                 max_nms_inputs=5000, max_output_size=100))


@pytest.fixture(scope="module")
def d2_oracle():
    """One 1024x1024 image, T=2, through the CPU oracle (about a minute on the box's cores)."""
    p = make_params(mc_dropoutsamp=2, **D2)
    w = make_weights(p, seed=5)
    imgs = make_images(1, 1024, 1024, seed=5)
    rcls, rbox, scales = _oracle_heads(p, w, imgs, (1024, 1024), 13)
    return p, w, imgs, rcls, rbox, scales


def _check_heads_grouped(got, want, groups, tol_max=2e-4, tol_rms=1e-4):
    """max-norm per level AND relative RMS per channel group (box deltas and sigma channels share a tensor:
    a max-norm over the whole tensor would let the small-magnitude group be wrong by a lot)."""
    for lvl, (g, r) in enumerate(zip(got, want)):
        assert g.shape == r.shape, (lvl, g.shape, r.shape)
        assert np.abs(g - r).max() <= tol_max * np.abs(r).max() + 1e-6, lvl
        for lo, hi in groups:
            gg, rr = g[..., lo:hi].astype(np.float64), r[..., lo:hi].astype(np.float64)
            rms = np.sqrt(np.mean(rr * rr))
            err = np.sqrt(np.mean((gg - rr) ** 2))
            assert err <= tol_rms * rms + 1e-7, "level %d channels %d:%d rel rms %g" % (lvl, lo, hi, err / max(rms, 1e-30))


def test_d2_full_resolution_heads_match_oracle(d2_oracle):
    p, w, imgs, rcls, rbox, scales = d2_oracle
    d = _driver(p, w, 1)
    d.set_dropout_seed(13)
    det = d.serve(imgs)
    cls, box = d.head_outputs(1)
    assert cls[0].shape == (2, 1, 128, 128, 63) and box[0].shape == (2, 1, 128, 128, 72) and len(cls) == 5
    _check_heads_grouped(cls, rcls, [(0, 63)])
    _check_heads_grouped(box, rbox, [(0, 36), (36, 72)])
    assert det[0].shape == (1, 100, 12) and det[3][0] == 100
    d.close()


def test_d2_full_resolution_postprocess_bit_exact(d2_oracle):
    """196 416 anchors per image: global NMS over the whole set and the eval-time settings
    (top-5000 pre-selection, per-class NMS) on the oracle's head outputs."""
    from oracle import post_ref as P
    p, w, imgs, rcls, rbox, scales = d2_oracle
    d = _driver(p, w, 1)
    _assert_tuple_equal(d.postprocess(rcls, rbox, scales), P.postprocess_global(p, rcls, rbox, scales), "global")
    d.close()
    pe = make_params(mc_dropoutsamp=2, **dict(D2, **EVAL_NMS))
    d = _driver(pe, w, 1)
    _assert_tuple_equal(d.postprocess(rcls, rbox, scales, post_mode="per_class"),
                        P.postprocess_per_class(pe, rcls, rbox, scales), "per_class top-5000")
    _assert_tuple_equal(d.postprocess(rcls, rbox, scales), P.postprocess_global(pe, rcls, rbox, scales), "global top-5000")
    d.close()


def test_d2_t30_share_of_config5_properties():
    """The per-GPU share of configs[4]: 2 images, T=30, D2 at 1024x1024, global and per-class/top-5000.
    Too slow for the oracle: size-independent properties + post-process of the run's own heads
    against the oracle's post-process (bit-exact)."""
    from oracle import post_ref as P
    p = make_params(mc_dropoutsamp=30, **D2)
    w = make_weights(p, seed=5)
    imgs = make_images(2, 1024, 1024, seed=6)
    d = _driver(p, w, 2)
    d.set_dropout_seed(3)
    boxes, scores, classes, valid, logits = d.serve(imgs)
    assert boxes.shape == (2, 100, 12) and classes.shape == (2, 100, 8) and logits.shape == (2, 100, 7)
    assert np.all(valid == 100) and np.all(np.isfinite(boxes)) and np.all(np.diff(scores, axis=1) <= 0)
    b = boxes[..., :4]
    assert b.min() >= 0 and b.max() <= 1024 and np.all(b[..., 2] >= b[..., 0]) and np.all(b[..., 3] >= b[..., 1])
    assert (boxes[..., 8:] > 0).mean() > 0.99 and np.all(boxes[..., 4:] >= 0)
    d.set_dropout_seed(3)
    again = d.serve(imgs)
    for a, c in zip(again, (boxes, scores, classes, valid, logits)):
        np.testing.assert_array_equal(a, c)
    # chunking / batch position invariance with the Philox image offset
    d1 = _driver(p, w, 1)
    d1.set_dropout_seed(3)
    d1.set_image_offset(1)
    one = d1.serve(imgs[1:])
    d1.close()
    for a, c in zip(one, (boxes, scores, classes, valid, logits)):
        np.testing.assert_array_equal(a, c[1:])
    # the oracle's post-process on the device's own T=30 heads of one image (one-class-at-a-time aggregate at
    # full size, 196 416 candidates)
    cls, box = d.head_outputs(2)
    cls1, box1 = [c[:, :1] for c in cls], [x[:, :1] for x in box]
    want = P.postprocess_global(p, cls1, box1, np.ones(1, np.float32))
    for g, r in zip((boxes, scores, classes, valid, logits), want):
        np.testing.assert_array_equal(g[:1], r)
    d.close()
    pe = make_params(mc_dropoutsamp=30, **dict(D2, **EVAL_NMS))
    d = _driver(pe, w, 2)
    got = d.postprocess(cls, box, np.ones(2, np.float32), post_mode="per_class")
    wantc = P.postprocess_per_class(pe, cls1, box1, np.ones(1, np.float32))
    for g, r in zip(got, wantc):
        np.testing.assert_array_equal(g[:1], r)
    d.close()


@pytest.mark.parametrize("name,over,nsamp", [("lossatt", LOSS_ATT, 100), ("full_mc", FULL_MC, 30),
                                            ("full_mc_t10", dict(FULL_MC, mc_dropoutsamp=10), 16)])
def test_sample_decode_matches_oracle(name, over, nsamp):
    """a13, method "sample" (src/utils_box.py:162-184): moments over `decode_nsamples` decoded Normal draws, float64 on the
    device; TFP's stream is replaced by the build's Philox normal stream shared with the oracle.  The draws pass through
    double-precision log / cos / sin / exp of two different maths libraries, so the candidates are compared at 1e-5
    relative (they agree to ~1e-12 before the cast to float32) and the final tuple must match on all but tie-level rows."""
    from oracle import post_ref as P
    p = make_params(uncert_adjust_method="sample", decode_nsamples=nsamp, **over)
    w = make_weights(p, seed=91, cls_spread=20.0)
    rcls, rbox, scales = _oracle_heads(p, w, make_images(2, 100, 180, seed=92), (128, 192), 9)
    d = _driver(p, w, 2)
    d.set_dropout_seed(1234)
    got = d.postprocess(rcls, rbox, scales)
    cand = d.candidates(2)
    ref = P.pre_nms(p, rcls, rbox, decode_seed=1234)
    np.testing.assert_allclose(cand["boxes"], ref["boxes"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(cand["u_al"], ref["u_al"], rtol=1e-4, atol=1e-5)
    if ref["u_ep"] is not None:
        np.testing.assert_allclose(cand["u_ep"], ref["u_ep"], rtol=1e-3, atol=1e-4)
    np.testing.assert_array_equal(cand["scores"], ref["scores"])
    want = P.postprocess_global(p, rcls, rbox, scales, decode_seed=1234)
    np.testing.assert_array_equal(got[3], want[3])
    cid = lambda c: c if c.ndim == 2 else c[..., 0]
    same = (cid(got[2]) == cid(want[2])) & (np.abs(got[0][..., :4] - want[0][..., :4]).max(-1) < 1e-2)
    assert same.mean() > 0.97
    # the sampled moments converge on the closed form (l-norm) as the reference's comparison of the methods expects
    p2 = make_params(**over)
    d2 = _driver(p2, w, 2)
    d2.postprocess(rcls, rbox, scales)
    c2 = d2.candidates(2)
    rel = np.abs(cand["boxes"] - c2["boxes"]).mean() / np.abs(c2["boxes"]).mean()
    assert rel < 0.05, rel
    # a different seed draws different samples
    d.set_dropout_seed(99)
    d.postprocess(rcls, rbox, scales)
    assert not np.array_equal(d.candidates(2)["boxes"], cand["boxes"])
    d.close()
    d2.close()


TOPK_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import FULL_MC, PLAIN, make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
from oracle import effdet_ref as E, philox_ref as R, post_ref as P, preprocess_ref as PP
TOPK = dict(nms_configs=dict(method="gaussian", iou_thresh=None, score_thresh=0.0, sigma=None, pyfunc=False, max_nms_inputs=500,
                             max_output_size=100))
for over, quant in ((dict(PLAIN, **TOPK), True), (dict(FULL_MC, **TOPK), False)):
    p = make_params(**over)
    w = make_weights(p, seed=23)
    x, scales = PP.preprocess(make_images(2, 128, 192, seed=24), (128, 192), p["mean_rgb"], p["stddev_rgb"])
    sites = E.dropout_sites(p)
    masks = R.make_masks(sites, 0, 2, p["mc_dropoutsamp"]) if sites else None
    rcls, rbox = E.forward(w, p, x, masks)
    if quant:
        rcls = [np.round(c * 4) / 4 for c in rcls]              # thousands of exact ties: the index rule decides
    want = P.pre_nms(p, rcls, rbox)
    d = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
    d.postprocess(rcls, rbox, scales)
    got = d.candidates(2)
    for key in ("classes", "scores", "boxes"):
        assert np.array_equal(got[key], want[key]), (key, quant)
    for mode in ("global", "per_class"):
        g = d.postprocess(rcls, rbox, scales, post_mode=mode)
        r = (P.postprocess_global if mode == "global" else P.postprocess_per_class)(p, rcls, rbox, scales)
        for a, b in zip(g, r):
            assert np.array_equal(a, b), (mode, quant)
    d.close()
print("topk ok")
"""


@pytest.mark.parametrize("multi", ["2", "0"], ids=["multi-block", "one-block-per-image"])
def test_topk_selection_paths_bit_exact(multi):
    """a10: both top-k implementations - the device-wide radix select on 53-bit composite keys (forced for a short list with
    UDA_TOPK_MULTI=2; taken by itself from 65 536 values per image, i.e. by the D2 tests above) and the one-block-per-image
    kernel - give the oracle's candidates, exact ties included (value descending, ties -> lower flat index)."""
    e = dict(os.environ, UDA_TOPK_MULTI=multi)
    r = subprocess.run([sys.executable, "-c", TOPK_WORKER % {"root": ROOT}], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "topk ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
