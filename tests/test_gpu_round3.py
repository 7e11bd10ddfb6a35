"""Round-3 GPU tests (VERDICT r02 "Next round" 3a, 3b, 3e, 4, 5), all through the C ABI:

  * BASELINE configs[2] and configs[3] at their REAL shapes: size-independent properties on the whole batch plus the
    oracle post-process on the GPU's own head outputs;
  * f3 on the device path end to end: serve_unpacked -> Box/ClassCalibrator -> writers -> parse the lines back;
  * f4: batches whose raw sizes differ per image (KITTI's four sizes, and the 375x1220 of dataset_data.py:105),
    bit-exact preprocessing against the oracle per image; image files decoded on the host;
  * feed hiding: serve_stream (upload of batch i+1 under the network of batch i) returns what serve returns;
  * the "single-launch NMS wanted but not launched" counter.
"""
import ast
import os
import subprocess
import sys

import numpy as np
import pytest

from common import FULL_MC, LOSS_ATT, ROOT, make_images, make_params, make_weights

pytestmark = pytest.mark.gpu


def _driver(params, w, batch, **kw):
    from uda_amd.infer_lib import KerasDriver
    return KerasDriver("_", False, params["name"], batch_size=batch, model_params=params, weights=w, **kw)


def _eq(got, want, what=""):
    assert len(got) == len(want), (what, len(got), len(want))
    for k, (g, r) in enumerate(zip(got, want)):
        assert g.shape == r.shape and g.dtype == r.dtype, (what, k, g.shape, r.shape)
        np.testing.assert_array_equal(g, r, err_msg="%s output %d" % (what, k))


# ------------------------------------------------------------------ configs[2] at its real shape
def test_config2_real_shape_properties_and_oracle_postprocess():
    """BASELINE configs[2], the per-GPU share: 32 raw 1280x720 images (scale 1, 48 zero rows padded), C = 10, T = 20,
    full MC: output layout, determinism, image-offset / chunk invariance (what the 8-GPU image sharding relies on),
    and the oracle's post-process on this handle's own head outputs of one image, bit for bit."""
    from oracle import post_ref as P
    p = make_params(image_size="1280x768", num_classes=10, **dict(FULL_MC, mc_dropoutsamp=20))
    w = make_weights(p, seed=3)
    imgs = make_images(32, 720, 1280, seed=3)
    d = _driver(p, w, 32, chunk_images=32)
    d.set_dropout_seed(11)
    boxes, scores, classes, valid, logits = det = d.serve(imgs)
    assert boxes.shape == (32, 100, 12) and classes.shape == (32, 100, 11) and logits.shape == (32, 100, 10)
    assert np.all(valid == 100) and np.all(np.isfinite(boxes)) and np.all(np.diff(scores, axis=1) <= 0)
    assert boxes[..., [0, 2]].max() <= 768 and boxes[..., [1, 3]].max() <= 1280 and boxes[..., :4].min() >= 0
    assert np.all(classes[..., 0] >= 1) and np.all(classes[..., 0] <= 10) and (boxes[..., 8:] > 0).mean() > 0.99
    _, scales = d.preprocessed_scales(32)
    np.testing.assert_array_equal(scales, np.ones(32, np.float32))
    assert d.nms_coop_fallbacks() == 0 and d.nms_coop_not_launched() == 0 and d.nms_prefix_fallbacks() == 0
    again = d.serve(imgs)
    _eq(again, det, "rerun")
    # shard of 4 images at offset 20 in a handle of its own, chunked 3 + 1
    d2 = _driver(p, w, 4, chunk_images=3)
    d2.set_dropout_seed(11)
    d2.set_image_offset(20)
    part = d2.serve(imgs[20:24])
    d2.close()
    _eq(part, tuple(a[20:24] for a in det), "shard at offset 20")
    # own heads of image 0 -> oracle post-process == GPU post-process == serve
    d.serve(imgs[:1])
    cls, box = d.head_outputs(1)
    assert cls[0].shape == (20, 1, 96, 160, 90) and box[4].shape == (20, 1, 6, 10, 72)
    want = P.postprocess_global(p, cls, box, np.ones(1, np.float32))
    _eq(d.postprocess(cls, box, np.ones(1, np.float32)), want, "oracle post-process on own heads")
    _eq(tuple(a[:1] for a in det), want, "serve")
    d.close()


# ------------------------------------------------------------------ configs[3] at its real shape
def test_config3_real_shape_ensemble_vs_oracle_on_members_heads():
    """BASELINE configs[3], one GPU's share: 5 independently initialised members x 8 images at 1280x768, aggregated like
    MC samples.  The oracle post-process on the members' own GPU head outputs must reproduce the ensemble's detections
    bit for bit (2 of the 8 images: the oracle's NMS over 184 140 near-tied candidates takes seconds per image)."""
    from oracle import post_ref as P
    from uda_amd.infer_lib import EnsembleDriver
    p = make_params(image_size="1280x768", **LOSS_ATT)
    ws = [make_weights(p, seed=40 + m) for m in range(5)]
    imgs = make_images(8, 768, 1280, seed=4)
    ens = EnsembleDriver(ws, p["name"], batch_size=8, model_params=p, chunk_images=8)
    got = ens.serve(imgs)
    assert got[0].shape == (8, 100, 12) and got[2].shape == (8, 100, 8) and got[4].shape == (8, 100, 7)
    assert np.all(got[3] == 100) and np.all(np.isfinite(got[0])) and (got[0][..., 8:] > 0).mean() > 0.99
    _eq(ens.serve(imgs), got, "rerun")
    heads = [m.head_outputs(8) for m in ens.members]
    pm = dict(p, mc_dropout=True, mc_dropoutrate=1e-9, mc_dropoutsamp=5)
    for lo in (0, 6):
        cls_g = [np.stack([heads[m][0][l][lo:lo + 1] for m in range(5)]) for l in range(5)]
        box_g = [np.stack([heads[m][1][l][lo:lo + 1] for m in range(5)]) for l in range(5)]
        want = P.postprocess_global(pm, cls_g, box_g, np.ones(1, np.float32))
        _eq(tuple(a[lo:lo + 1] for a in got), want, "image %d" % lo)
    # members differ (independent weight sets) and a member alone equals a plain driver with its weights
    assert not np.array_equal(heads[0][0][0], heads[1][0][0])
    solo = _driver(p, ws[3], 2)
    solo.serve(imgs[:2])
    c3, b3 = solo.head_outputs(2)
    np.testing.assert_array_equal(c3[2], heads[3][0][2][:2])
    np.testing.assert_array_equal(b3[0], heads[3][1][0][:2])
    solo.close()
    ens.close()


# ------------------------------------------------------------------ f3: serve -> calibrate -> write -> parse
def test_serve_calibrate_write_parse_chain(tmp_path):
    """SURVEY 8f.1-3 as ONE flow on GPU output (infer_model.py:585-636,836-960; validate_model.py:159-202,524-681):
    serve_unpacked -> BoxCalibrator / ClassCalibrator -> prediction_data.txt / validate_results.txt -> parse every line
    back as the consumers do (ast.literal_eval, active_learning_loop.py:532) -> every field equals the oracle's
    unpack / calibrate restatements of the same serve() tuple rounded to 4 decimals."""
    from oracle import calib_ref as CR, unpack_ref as U
    from uda_amd import writers as W
    from uda_amd.calibration import BoxCalibrator, ClassCalibrator, IsoTable
    p = make_params(**FULL_MC)
    C = p["num_classes"]
    d = _driver(p, make_weights(p, cls_spread=20.0), 3)
    d.set_dropout_seed(5)
    imgs = make_images(3, 128, 192, seed=8)
    un = d.serve_unpacked(imgs)
    d.set_dropout_seed(5)
    det = d.serve(imgs)                                   # the same tuple, for the oracle side
    rng = np.random.default_rng(3)

    def table(hi, k=10):
        return np.sort(rng.uniform(0, hi, k)) + np.arange(k) * 1e-3, np.sort(rng.uniform(0, hi * 1.5, k))
    box_models = dict(ts_all=1.7, iso_percoo=[table(40) for _ in range(4)], iso_perclscoo=[table(40) for _ in range(4 * C)])
    cls_models = dict(ts_all=1.9, iso_percls=[table(1, 14) for _ in range(C)])
    as_dev = lambda m: {k: (v if k.startswith("ts") else [IsoTable(*t) for t in v]) for k, v in m.items()}
    bc, cc = BoxCalibrator(d, as_dev(box_models)), ClassCalibrator(d, as_dev(cls_models), draws=10, seed=77)
    calibrated = {}
    for m in box_models:
        calibrated[m + "_albox"] = bc.calibrate_boxuncert(3, "albox", m)
        calibrated[m + "_mcbox"] = bc.calibrate_boxuncert(3, "mcbox", m)
    for m in cls_models:
        ent, prob, unc = cc.perform_class_calib(3, m)
        calibrated[m + "_probab"], calibrated[m + "_entropy"], calibrated[m + "_mcclass"] = prob, ent, unc
    names = ["000%d" % i for i in range(3)]
    thr = float(np.sort(det[1].ravel())[-13])           # a score threshold that keeps a dozen detections
    recs = W.prediction_records(un, names, thr, calibrated)
    path = str(tmp_path / "prediction_data.txt")
    W.write_prediction_data(path, recs)
    lines = [ast.literal_eval(l.replace("inf", "2e308")) for l in open(path)]
    assert len(lines) == int((det[1] > thr).sum()) > 3

    # ---- the oracle's view of the same tuple
    boxes4, cls_id, albox, mcbox, mcclass = U.unpack(p, det[0], det[2])
    r4 = lambda a: np.nan_to_num(np.around(np.asarray(a, np.float32), 4)).astype(np.float32)
    k = 0
    for i in range(3):
        probab, entropy = U.probab_entropy(det[4][i])
        ref_box = {m: {"albox": CR.calibrate_boxuncert(m, box_models, C, det[0][i][:, 4:8], det[2][i][:, 0], det[0][i][:, :4]),
                       "mcbox": CR.calibrate_boxuncert(m, box_models, C, det[0][i][:, 8:12], det[2][i][:, 0], det[0][i][:, :4])}
                   for m in box_models}
        for sel in np.where(det[1][i] > thr)[0]:
            L = lines[k]; k += 1
            assert L["image_name"] == names[i] + ".jpg" and L["score_thresh"] == thr
            f32 = lambda v: np.asarray(v, np.float32)
            np.testing.assert_array_equal(f32(L["det_score"]), det[1][i][sel])
            np.testing.assert_array_equal(f32(L["bbox"]), boxes4[i][sel])
            assert L["class"] == cls_id[i][sel]
            np.testing.assert_array_equal(f32(L["logits"]), r4(det[4][i][sel]))
            np.testing.assert_allclose(f32(L["probab"]), probab[sel], rtol=2e-6, atol=1e-7)
            np.testing.assert_allclose(f32(L["entropy"]), r4(entropy[sel]), atol=1.01e-4)
            np.testing.assert_array_equal(f32(L["uncalib_albox"]), r4(albox[i][sel]))
            np.testing.assert_array_equal(f32(L["uncalib_mcbox"]), r4(mcbox[i][sel]))
            np.testing.assert_array_equal(f32(L["uncalib_mcclass"]), r4(mcclass[i][sel]))
            for m in box_models:
                for which in ("albox", "mcbox"):
                    np.testing.assert_allclose(f32(L["%s_%s" % (m, which)]), r4(ref_box[m][which][sel]), rtol=3e-6, atol=1.01e-4)
    assert k == len(lines)
    # class calibration: the restatement works on all rows of the batch at once (row index = Philox counter)
    logits_all, unc_all = det[4].reshape(-1, C), det[2][..., 1:].reshape(-1, C)
    for m in cls_models:
        ent, prob, unc = CR.perform_class_calib(m, cls_models, logits_all, unc_all, draws=10, seed=77)
        k = 0
        for i in range(3):
            for sel in np.where(det[1][i] > thr)[0]:
                L = lines[k]; k += 1
                row = i * d.M + sel
                np.testing.assert_allclose(np.asarray(L[m + "_probab"], np.float32), r4(prob[row]), atol=1.5e-4)
                np.testing.assert_allclose(np.float32(L[m + "_entropy"]), r4(ent[row]), atol=3e-4)
                np.testing.assert_allclose(np.asarray(L[m + "_mcclass"], np.float32), r4(unc[row]), atol=1.5e-4)

    # ---- the same flow as one call over two batches (feed hidden, calibration while each batch is resident): same lines
    path2 = str(tmp_path / "prediction_data_stream.txt")
    d.set_dropout_seed(5)
    nrec = W.predict_to_file(d, [imgs, imgs[:2]], [names, ["a", "b"]], path2, thr, bc, cc, tuple(box_models), tuple(cls_models))
    lines2 = open(path2).read().splitlines()
    assert nrec == len(lines2) and lines2[:len(lines)] == open(path).read().splitlines()
    assert len(lines2) > len(lines) and ast.literal_eval(lines2[len(lines)])["image_name"] == "a.jpg"

    # ---- validate_results.txt: detections "matched" to synthetic ground truth (the assignment itself is host analysis)
    keep = [(i, s) for i in range(3) for s in np.where(det[1][i] > thr)[0]][:7]
    filt = dict(names=[names[i] for i, _ in keep], scores=np.array([det[1][i][s] for i, s in keep]),
                boxes=np.array([boxes4[i][s] for i, s in keep]), gt_boxes=np.array([boxes4[i][s] + 1 for i, s in keep]),
                occlusions=[0] * len(keep), truncations=[np.float32(0.25)] * len(keep),
                classes=[int(cls_id[i][s]) for i, s in keep], gt_classes=[1] * len(keep),
                logits=np.array([det[4][i][s] for i, s in keep]), probab=np.array([un["probab"][i][s] for i, s in keep]),
                entropy=np.array([un["entropy"][i][s] for i, s in keep]), mcclass=np.array([mcclass[i][s] for i, s in keep]),
                mcbox=np.array([mcbox[i][s] for i, s in keep]), albox=np.array([albox[i][s] for i, s in keep]))
    vcal = {"ts_all_albox": np.array([calibrated["ts_all_albox"][i][s] for i, s in keep])}
    vp = dict(p, calibrate_regression=True, calibrate_classification=False)
    vpath = str(tmp_path / "validate_results.txt")
    W.write_validate_results(vpath, W.validate_records(filt, vp, vcal))
    vlines = [ast.literal_eval(l.replace("inf", "2e308")) for l in open(vpath)]
    assert len(vlines) == len(keep)
    for L, (i, s) in zip(vlines, keep):
        np.testing.assert_array_equal(np.asarray(L["bbox"], np.float32), boxes4[i][s])
        np.testing.assert_array_equal(np.asarray(L["uncalib_mcbox"], np.float32), mcbox[i][s])
        np.testing.assert_array_equal(np.asarray(L["uncalib_albox"], np.float32), albox[i][s])
        np.testing.assert_allclose(np.asarray(L["ts_all_albox"], np.float32), r4(albox[i][s] / np.float32(1.7)), atol=1.01e-4)
        assert list(L)[:8] == ["image_name", "score", "bbox", "gt_bbox", "gt_occl", "gt_trunc", "class", "gt_class"]
    d.close()


# ------------------------------------------------------------------ f4: heterogeneous raw sizes, image files
KITTI_SIZES = [(375, 1242), (370, 1224), (374, 1238), (376, 1241)]


def _ragged_images(sizes, seed):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]


def test_ragged_kitti_batch_preprocess_bit_exact_and_serve(tmp_path):
    """One batch of KITTI's four raw sizes + the 375x1220 of dataset_data.py:105 at the 1280x768 network size: every image
    gets its own scale (dataloader.py:123-152); preprocessing bit-exact against the oracle image by image, detections
    equal to serving each image alone (same Philox rows through the image offset)."""
    from oracle import preprocess_ref as PP
    sizes = KITTI_SIZES + [(375, 1220)]
    imgs = _ragged_images(sizes, 21)
    p = make_params(image_size="1280x768", **dict(FULL_MC, mc_dropoutsamp=2))
    w = make_weights(p, seed=2, cls_spread=20.0)
    d = _driver(p, w, 5)
    d.set_dropout_seed(4)
    det = d.serve(imgs)
    got, scales = d.preprocessed()
    for i, im in enumerate(imgs):
        want, s = PP.preprocess(im[None], (768, 1280), p["mean_rgb"], p["stddev_rgb"])
        np.testing.assert_array_equal(got[i], want[0], err_msg="image %d %s" % (i, sizes[i]))
        np.testing.assert_array_equal(scales[i], s[0])
    assert len(set(scales.tolist())) >= 4                       # really per-image scales (1242 -> 0.97, 1224 -> 0.956, ...)
    one = _driver(p, w, 1)
    one.set_dropout_seed(4)
    for i in (0, 1, 4):
        one.set_image_offset(i)
        _eq(one.serve(imgs[i]), tuple(a[i:i + 1] for a in det), "image %d alone" % i)
    one.close()
    # a list of equally sized images still takes the uniform path, and the same driver can alternate
    same = _ragged_images([(375, 1242)] * 3, 5)
    _eq(d.serve(same), d.serve(np.stack(same)), "list vs array")
    # image files decoded on the host (lossless PNG), served as one batch
    from PIL import Image
    from uda_amd.infer_lib import read_images
    paths = []
    for i, im in enumerate(imgs[:4]):
        paths.append(str(tmp_path / ("%06d.png" % i)))
        Image.fromarray(im).save(paths[-1])
    back = read_images(paths)
    for a, b in zip(back, imgs):
        np.testing.assert_array_equal(a, b)
    d.set_image_offset(0)
    _eq(d.serve_files(paths), tuple(a[:4] for a in det), "serve_files")
    with pytest.raises(ValueError):
        d.serve([imgs[0], imgs[1][..., :2]])
    d.close()


def test_ragged_small_sizes_up_and_down_scaling_bit_exact():
    """Mixed up- and down-scaling inside one batch (incl. an image that needs no resize and one that is limited by its
    height), at a small network size the oracle finishes at once."""
    from oracle import preprocess_ref as PP
    sizes = [(128, 192), (100, 180), (64, 64), (300, 250), (37, 190), (128, 100)]
    imgs = _ragged_images(sizes, 7)
    p = make_params(**LOSS_ATT)
    d = _driver(p, make_weights(p), 6)
    d.serve(imgs)
    got, scales = d.preprocessed()
    for i, im in enumerate(imgs):
        want, s = PP.preprocess(im[None], (128, 192), p["mean_rgb"], p["stddev_rgb"])
        np.testing.assert_array_equal(got[i], want[0], err_msg=str(sizes[i]))
        np.testing.assert_array_equal(scales[i], s[0])
    d.close()


# ------------------------------------------------------------------ feed hiding
def test_serve_stream_prefetch_equals_serve():
    """serve_stream uploads batch i+1 (second input slot, copy stream, pinned staging) under the network of batch i: same
    detections as serve() batch by batch, for uniform batches, ragged batches and a short last batch."""
    p = make_params(**dict(FULL_MC, mc_dropoutsamp=2))
    w = make_weights(p, cls_spread=20.0)
    batches = [make_images(4, 128, 192, seed=1), _ragged_images([(100, 180), (128, 150), (90, 192)], 2),
               make_images(4, 100, 180, seed=3), make_images(2, 128, 192, seed=4), make_images(4, 128, 192, seed=5)]
    d = _driver(p, w, 4)
    d.set_dropout_seed(9)
    want = [d.serve(b) for b in batches]
    d.set_dropout_seed(9)
    got = list(d.serve_stream(batches))
    assert len(got) == len(want)
    for i, (g, r) in enumerate(zip(got, want)):
        _eq(g, r, "batch %d" % i)
    assert list(d.serve_stream([])) == []
    # explicit calls, as bench.py uses them
    d.stage_images(batches[0])
    d.run_resident(sync=False)
    d.prefetch_images(batches[2])
    _eq(d._collect(4), want[0], "resident")
    assert d.swap_prefetched() == 4
    d.run_resident(sync=True)
    _eq(d._collect(4), want[2], "prefetched")
    d.close()


# ------------------------------------------------------------------ counters
COUNTER_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import LOSS_ATT, make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
p = make_params(image_size="512x384", **LOSS_ATT)           # 36 828 anchors: above the single-block limit (8192)
d = KerasDriver("_", False, p["name"], 2, False, p, weights=make_weights(p))
det = d.serve(make_images(2, 384, 512))
print("counters", d.nms_coop_not_launched(), d.nms_coop_fallbacks())
np.savez(sys.argv[1], *det)
d.close()
"""


def test_single_launch_nms_not_launched_is_counted(tmp_path):
    """A capacity query that answers 0 (UDA_NMS_COOP_CAP=0, the hook of the failure seen in round 2) silently moved the NMS
    to the slower versions; now `uda_nms_coop_not_launched` says so, and the results do not change."""
    outs = {}
    for cap in ("0", None):
        e = dict(os.environ)
        if cap is not None:
            e["UDA_NMS_COOP_CAP"] = cap
        out = str(tmp_path / ("cap%s.npz" % cap))
        r = subprocess.run([sys.executable, "-c", COUNTER_WORKER % {"root": ROOT}, out], cwd=ROOT, env=e, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
        outs[cap] = ([l for l in r.stdout.splitlines() if l.startswith("counters")][-1].split()[1:], dict(np.load(out)))
    assert int(outs["0"][0][0]) >= 1 and outs[None][0] == ["0", "0"]
    for k in outs[None][1]:
        np.testing.assert_array_equal(outs["0"][1][k], outs[None][1][k])


# ------------------------------------------------------------------ six-term (float32-equivalent) products keep the fusion
SIX_WORKER = r"""
import json, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import FULL_MC, HEAD_MC, make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
from oracle import effdet_ref as E, philox_ref as R, preprocess_ref as PP
out = {}
for name, kw, hw in (("d0", dict(image_size="192x128", **FULL_MC), (128, 192)),
                     ("d2", dict(model="efficientdet-d2", image_size="128x128", **FULL_MC), (128, 128))):
    p = make_params(**kw)
    w = make_weights(p, seed=31)
    imgs = make_images(2, hw[0] - 28, hw[1] - 12, seed=32)
    d = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
    d.set_dropout_seed(7)
    d.serve(imgs)
    cls, box = d.head_outputs(2)
    n_mbx = sum(1 for o in d.plan.ops if o["kind"] == 7)
    x, _ = PP.preprocess(imgs, hw, p["mean_rgb"], p["stddev_rgb"])
    rcls, rbox = E.forward(w, p, x, R.make_masks(E.dropout_sites(p), 7, 2, 3))
    worst_max = worst_rms = 0.0
    for l in range(5):
        for g, r, groups in ((cls[l], rcls[l], [(0, cls[l].shape[-1])]), (box[l], rbox[l], [(0, 36), (36, 72)])):
            for lo, hi in groups:
                gg, rr = g[..., lo:hi].astype(np.float64), r[..., lo:hi].astype(np.float64)
                worst_max = max(worst_max, np.abs(gg - rr).max() / np.abs(rr).max())
                worst_rms = max(worst_rms, np.sqrt(np.mean((gg - rr) ** 2)) / np.sqrt(np.mean(rr * rr)))
    out[name] = dict(max=worst_max, rms=worst_rms, n_mbx=n_mbx)
    d.close()
print("RESULT " + json.dumps(out))
"""


def _six_run(env):
    e = dict(os.environ)
    e.pop("UDA_PW_TERMS", None)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", SIX_WORKER % {"root": ROOT}], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    import json
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_six_term_products_keep_the_fused_kernels_and_are_float32_equivalent():
    """The float32-class split schemes (the reference computes in float32, utils.py:595-609): UDA_PW_SCHEME=bf16x3 (three bf16
    pieces per operand, six cross terms) used to switch the fused MBConv kernels off; mbxb_kernel / mbxd_kernel<..., PARTS = 3>
    keep every block fused.  UDA_PW_SCHEME=f16x2 (two fp16 pieces, three cross terms: the shipped default) must meet the
    SAME bars.  Heads of D0 (all 15 fused blocks incl. the deep ones) and D2 against the float32 CPU oracle: an order of
    magnitude tighter than two bf16 pieces, and no worse than the unfused six-term path."""
    six = _six_run(dict(UDA_PW_SCHEME="bf16x3"))
    half = _six_run(dict(UDA_PW_SCHEME="f16x2"))
    unfused = _six_run(dict(UDA_PW_SCHEME="bf16x3", UDA_FUSE_MBX6="0"))
    three = _six_run(dict(UDA_PW_SCHEME="bf16x2"))
    print("head error vs the float32 oracle (max / rel. rms): bf16x3 %s  f16x2 %s  bf16x2 %s" % (six, half, three))
    for m in ("d0", "d2"):
        assert six[m]["n_mbx"] >= 14 and unfused[m]["n_mbx"] == 0, (m, six[m], unfused[m])
        assert six[m]["n_mbx"] == three[m]["n_mbx"] == half[m]["n_mbx"]
        for name, run in (("bf16x3", six), ("f16x2", half)):
            assert run[m]["max"] <= 2e-5 and run[m]["rms"] <= 1e-5, (name, m, run[m])  # vs 2e-4 / 1e-4 for two bf16 pieces
            assert run[m]["rms"] <= 2.0 * unfused[m]["rms"] + 1e-7, (name, m, run[m], unfused[m])
            assert run[m]["rms"] < 0.5 * three[m]["rms"], (name, m, run[m], three[m])
