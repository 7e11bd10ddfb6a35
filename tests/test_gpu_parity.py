"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs.

Bars:
  * Philox masks, preprocessing, anchor table, NMS keep-sets, and the WHOLE post-process given
    identical head outputs: bit-exact (integer / index work and the pinned-down float order).
  * network head outputs: float32, |err| <= 2e-4 * max|ref| per level (different summation
    order in the MFMA GEMM vs the oracle's torch convs).
  * end-to-end serve: detections compared after matching by candidate index, 1e-3 relative.
"""
import numpy as np
import pytest

from common import (BOX_ONLY_MC, FULL_MC, HEAD_MC, LOSS_ATT, MC_NO_ATT, PLAIN, make_images, make_params,
                    make_weights)

pytestmark = pytest.mark.gpu


def _driver(params, w, batch, **kw):
    from uda_amd.infer_lib import KerasDriver, ServingDriver
    return KerasDriver("_", False, params["name"], batch_size=batch, only_network=kw.pop("only_network", False),
                         model_params=params, weights=w, **kw)


def _oracle_net(params, w, x, seed):
    from oracle import effdet_ref as E, philox_ref as R
    sites = E.dropout_sites(params)
    T = params["mc_dropoutsamp"] if params["mc_dropout"] else 1
    masks = R.make_masks(sites, seed, x.shape[0], T) if sites else None
    return E.forward(w, params, x, masks), masks


def test_library_loads_and_reports_errors():
    from uda_amd import capi
    lib = capi.load()
    for name in capi.EXPORTS:
        assert hasattr(lib, name)


@pytest.mark.parametrize("mode", ["full", "head"])
def test_philox_masks_bit_exact(mode):
    from oracle import effdet_ref as E, philox_ref as R
    p = make_params(**(FULL_MC if mode == "full" else HEAD_MC))
    w = make_weights(p)
    d = _driver(p, w, 3)
    d.set_dropout_seed(0x1234567890ABCDEF)
    d.serve(make_images(3, 128, 192))
    got = d.dropout_masks(3)
    want = R.make_masks(E.dropout_sites(p), 0x1234567890ABCDEF, 3, p["mc_dropoutsamp"])
    assert set(got) == set(want)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    d.close()


@pytest.mark.parametrize("hw", [(128, 192), (100, 180), (61, 77), (150, 400), (300, 200)])
def test_preprocess_bit_exact(hw):
    from oracle import preprocess_ref as PP
    p = make_params()
    w = make_weights(p)
    d = _driver(p, w, 2)
    imgs = make_images(2, hw[0], hw[1], seed=hw[0])
    d.serve(imgs)
    d._last_n = 2
    got, scales = d.preprocessed()
    want, wscales = PP.preprocess(imgs, d.image_size, p["mean_rgb"], p["stddev_rgb"])
    np.testing.assert_array_equal(scales, wscales)
    np.testing.assert_array_equal(got, want)
    d.close()


from common import check_heads as _check_heads          # noqa: E402  (shared with test_gpu_fullsize / test_gpu_configs)


@pytest.mark.parametrize("name,over", [("plain", PLAIN), ("lossatt", LOSS_ATT), ("full_mc", FULL_MC),
                                       ("head_mc", HEAD_MC), ("mc_noatt", MC_NO_ATT), ("box_only", BOX_ONLY_MC)])
def test_network_heads_match_oracle(name, over):
    from oracle import preprocess_ref as PP
    p = make_params(**over)
    w = make_weights(p, seed=3)
    d = _driver(p, w, 2, only_network=True)
    x, _ = PP.preprocess(make_images(2, 128, 192, seed=5), d.image_size, p["mean_rgb"], p["stddev_rgb"])
    d.set_dropout_seed(77)
    cls, box = d.predict(x)
    (rcls, rbox), _ = _oracle_net(p, w, x, 77)
    _check_heads(cls, rcls)
    _check_heads(box, rbox)
    d.close()


def test_network_chunking_is_invisible():
    """chunk_images only changes how many images go through the op list at once."""
    from oracle import preprocess_ref as PP
    p = make_params(**FULL_MC)
    w = make_weights(p, seed=4)
    x, _ = PP.preprocess(make_images(3, 128, 192, seed=6), (128, 192), p["mean_rgb"], p["stddev_rgb"])
    outs = []
    for chunk in (1, 2, 3):
        d = _driver(p, w, 3, only_network=True, chunk_images=chunk)
        d.set_dropout_seed(5)
        outs.append(d.predict(x))
        d.close()
    for o in outs[1:]:
        for a, b in zip(o[0] + o[1], outs[0][0] + outs[0][1]):
            np.testing.assert_array_equal(a, b)


def test_d2_topology_matches_oracle():
    p = make_params(model="efficientdet-d2", image_size="128x128", **HEAD_MC)
    w = make_weights(p, seed=8)
    from oracle import preprocess_ref as PP
    d = _driver(p, w, 1, only_network=True)
    x, _ = PP.preprocess(make_images(1, 128, 128, seed=9), d.image_size, p["mean_rgb"], p["stddev_rgb"])
    d.set_dropout_seed(3)
    cls, box = d.predict(x)
    (rcls, rbox), _ = _oracle_net(p, w, x, 3)
    _check_heads(cls, rcls)
    _check_heads(box, rbox)
    d.close()


def _rand_boxes(rng, n, span=400.0, tied=False):
    c = rng.uniform(0, span, (n, 2))
    wh = rng.uniform(4, 120, (n, 2))
    b = np.stack([c[:, 0] - wh[:, 0] / 2, c[:, 1] - wh[:, 1] / 2, c[:, 0] + wh[:, 0] / 2,
                  c[:, 1] + wh[:, 1] / 2], 1).astype(np.float32)
    if tied:
        s = (0.01 + rng.normal(0, 1e-4, n)).astype(np.float32)
        s[rng.integers(0, n, n // 8)] = s[0]            # exact ties -> index tie-break
    else:
        s = rng.uniform(0, 1, n).astype(np.float32)
    return b, s


@pytest.mark.parametrize("n,tied,sigma,thr", [(0, False, 0.25, 0.001), (1, False, 0.25, 0.001),
                                              (50, False, 0.25, 0.001), (3000, False, 0.25, 0.001),
                                              (3000, True, 0.25, 0.001), (5000, True, 0.0, float("-inf")),
                                              (2500, False, 0.0, 0.3), (20000, True, 0.25, 0.001),
                                              (777, False, 0.5, 0.2), (3000, False, 0.3, 0.001),      # scale = -0.5 / 0.3: not a power of two
                                              (3000, True, 0.15, 0.001)])
def test_nms_kernel_bit_exact(n, tied, sigma, thr):
    from oracle import post_ref as P
    p = make_params()
    d = _driver(p, make_weights(p), 1)
    rng = np.random.default_rng(n + int(tied))
    n_img = 3
    boxes = np.zeros((n_img, max(n, 1), 4), np.float32)
    scores = np.zeros((n_img, max(n, 1)), np.float32)
    for i in range(n_img):
        if n:
            boxes[i], scores[i] = _rand_boxes(rng, n, tied=tied)
    if n == 0:
        scores[:] = -1.0
    idx, sc, valid = d.nms(boxes, scores, 100, 0.5, thr, sigma)
    for i in range(n_img):
        ridx, rsc, rvalid = P.nms_v5(boxes[i], scores[i], 100, 0.5, thr, sigma, True)
        assert valid[i] == rvalid
        np.testing.assert_array_equal(idx[i], ridx)
        np.testing.assert_array_equal(sc[i], rsc)
    d.close()


PREFIX_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import make_params, make_weights
from uda_amd.infer_lib import KerasDriver, ServingDriver
from oracle import post_ref as P
p = make_params()
d = KerasDriver("_", False, p["name"], batch_size=1, model_params=p, weights=make_weights(p))
def rand_boxes(rng, n, span, tied):
    c = rng.uniform(0, span, (n, 2)); wh = rng.uniform(4, 120, (n, 2))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    if tied:
        s = (0.01 + rng.normal(0, 1e-4, n)).astype(np.float32)
        s[rng.integers(0, n, n // 8)] = s[0]
    else:
        s = rng.uniform(0, 1, n).astype(np.float32)
    return b, s
for case in ("sparse", "dense", "all_tied", "identical_boxes", "threshold", "hard"):
    rng = np.random.default_rng(len(case))
    n_img, n = 3, 30000
    sigma, thr = 0.25, 0.001
    boxes = np.zeros((n_img, n, 4), np.float32); scores = np.zeros((n_img, n), np.float32)
    for i in range(n_img):
        boxes[i], scores[i] = rand_boxes(rng, n, 40000.0 if case == "sparse" else 400.0, case == "dense")
    expect = None
    if case == "sparse":
        expect = 0                      # boxes hardly overlap: the 100 winners are the top scores
    elif case == "all_tied":
        scores[:] = 0.37                # more exact ties than any prefix holds
        expect = n_img
    elif case == "identical_boxes":
        boxes[:] = boxes[:, :1]         # every selection decays every other score: the winners fall below the cut
        expect = n_img
    elif case == "threshold":
        sigma, thr = 0.0, 0.5           # hard NMS, half of the candidates above the threshold
    elif case == "hard":
        sigma, thr = 0.0, float("-inf")
    before = d.nms_prefix_fallbacks()
    idx, sc, valid = d.nms(boxes, scores, 100, 0.5, thr, sigma)
    fell = d.nms_prefix_fallbacks() - before
    for i in range(n_img):
        ridx, rsc, rvalid = P.nms_v5(boxes[i], scores[i], 100, 0.5, thr, sigma, True)
        assert valid[i] == rvalid, (case, i)
        assert (idx[i] == ridx).all() and (sc[i] == rsc).all(), (case, i)
    assert 0 <= fell <= n_img and (expect is None or fell == expect), (case, fell, expect)
    print(case, "redone", fell)
print("prefix ok")
d.close()
"""


def test_nms_score_prefix_and_fallback():
    """Where the co-resident grid cannot take a problem (here: switched off), candidate sets above 8192 run on their
    score prefix; when the device check rejects the prefix the problem is redone on the full set.  Either way:
    bit-exact against the oracle's NonMaxSuppressionV5; accepted (sparse boxes) and rejected (exact ties, decay below
    the cut) prefixes are both exercised."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, UDA_NMS_COOP="0")
    r = subprocess.run([sys.executable, "-c", PREFIX_WORKER % {"root": root}], cwd=root, env=e, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "prefix ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_postprocess_large_anchor_set_matches_oracle():
    """256x384 input: 18 414 candidates per image, so the global NMS leaves the single-block kernels (co-resident grid, or
    score prefix where that is off) inside the serve post-process; outputs must still equal the oracle's bit for bit."""
    from oracle import post_ref as P, preprocess_ref as PP
    p = make_params(image_size="384x256", **FULL_MC)
    x, scales = PP.preprocess(make_images(2, 200, 360, seed=32), (256, 384), p["mean_rgb"], p["stddev_rgb"])
    for seed, spread in ((31, 20.0), (33, 1.0)):     # spread-out scores / near-tied scores (plain random init)
        w = make_weights(p, seed=seed, cls_spread=spread)
        (rcls, rbox), _ = _oracle_net(p, w, x, 5)
        want = P.postprocess_global(p, rcls, rbox, scales)
        d = _driver(p, w, 2)
        for rep in range(3):      # a rejected prefix pauses the prefix path for the following runs: same outputs
            got = d.postprocess(rcls, rbox, scales)
            for k, (g, r) in enumerate(zip(got, want)):
                np.testing.assert_array_equal(g, r, err_msg="spread %g run %d output %d" % (spread, rep, k))
        print("spread %g: %d image(s) redone on the full set" % (spread, d.nms_prefix_fallbacks()))
        d.close()


@pytest.mark.parametrize("name,over", [("plain", PLAIN), ("lossatt", LOSS_ATT), ("full_mc", FULL_MC),
                                       ("head_mc", HEAD_MC), ("mc_noatt", MC_NO_ATT), ("box_only", BOX_ONLY_MC),
                                       ("full_mc_t10", dict(FULL_MC, mc_dropoutsamp=10)),      # register-resident aggregate
                                       ("mc_noatt_t10", dict(MC_NO_ATT, mc_dropoutsamp=10)),
                                       ("full_mc_t10_c3", dict(FULL_MC, mc_dropoutsamp=10, num_classes=3)),     # runtime class loop
                                       ("full_mc_t20_c10", dict(FULL_MC, mc_dropoutsamp=20, num_classes=10))])  # BDD-like
def test_postprocess_bit_exact_on_oracle_heads(name, over):
    """Same head outputs in -> the HIP post-process must reproduce the oracle's output tuple exactly."""
    from oracle import post_ref as P, preprocess_ref as PP
    p = make_params(**over)
    w = make_weights(p, seed=11, cls_spread=20.0 if name in ("plain", "full_mc") else 1.0)
    x, scales = PP.preprocess(make_images(2, 100, 180, seed=12), (128, 192), p["mean_rgb"], p["stddev_rgb"])
    from common import oracle_heads
    rcls, rbox = oracle_heads(p, w, x, 21)         # (T = 10 / 20: 4 real network passes, the other sample rows derived)
    want = P.postprocess_global(p, rcls, rbox, scales)
    d = _driver(p, w, 2)
    got = d.postprocess(rcls, rbox, scales)
    assert len(got) == len(want)
    for k, (g, r) in enumerate(zip(got, want)):
        assert g.shape == r.shape and g.dtype == r.dtype, (k, g.shape, r.shape, g.dtype, r.dtype)
        np.testing.assert_array_equal(g, r, err_msg="output %d" % k)
    d.close()


def test_serve_end_to_end_close_to_oracle():
    from oracle import post_ref as P, preprocess_ref as PP
    p = make_params(**FULL_MC)
    w = make_weights(p, seed=13, cls_spread=20.0)
    imgs = make_images(2, 100, 180, seed=14)
    d = _driver(p, w, 2)
    d.set_dropout_seed(99)
    got = d.serve(imgs)
    x, scales = PP.preprocess(imgs, (128, 192), p["mean_rgb"], p["stddev_rgb"])
    (rcls, rbox), _ = _oracle_net(p, w, x, 99)
    want = P.postprocess_global(p, rcls, rbox, scales)
    np.testing.assert_array_equal(got[3], want[3])
    # top detections: same boxes within 1e-3 of the box scale, same class, scores within 1e-3 relative
    for n in range(2):
        k = min(10, int(want[3][n]))
        np.testing.assert_allclose(got[1][n, :k], want[1][n, :k], rtol=1e-3)
        np.testing.assert_allclose(got[0][n, :k, :4], want[0][n, :k, :4], rtol=1e-3, atol=0.2)
        np.testing.assert_array_equal(got[2][n, :k, 0], want[2][n, :k, 0])
    d.close()


TOPK = dict(nms_configs=dict(method="gaussian", iou_thresh=None, score_thresh=0.0, sigma=None, pyfunc=False,
                             max_nms_inputs=500, max_output_size=100))
HARD = dict(nms_configs=dict(method="hard", iou_thresh=None, score_thresh=0.0, sigma=None, pyfunc=False,
                             max_nms_inputs=0, max_output_size=100))


@pytest.mark.parametrize("name,over,spread", [("full_mc", FULL_MC, 20.0), ("lossatt", LOSS_ATT, 1.0), ("plain", PLAIN, 20.0),
                                              ("topk_mc", dict(FULL_MC, **TOPK), 20.0), ("topk_plain", dict(PLAIN, **TOPK), 1.0),
                                              ("hard", dict(HEAD_MC, **HARD), 20.0)])
def test_per_class_postprocess_bit_exact(name, over, spread):
    """a17 (+a10 top-k): per-class NMS, concat, pad, top-100 on the oracle's head outputs."""
    from oracle import post_ref as P, preprocess_ref as PP
    p = make_params(**over)
    w = make_weights(p, seed=17, cls_spread=spread)
    x, scales = PP.preprocess(make_images(2, 100, 180, seed=18), (128, 192), p["mean_rgb"], p["stddev_rgb"])
    (rcls, rbox), _ = _oracle_net(p, w, x, 31)
    want = P.postprocess_per_class(p, rcls, rbox, scales)
    d = _driver(p, w, 2)
    got = d.postprocess(rcls, rbox, scales, post_mode="per_class")
    assert len(got) == 4
    for k, (g, r) in enumerate(zip(got, want)):
        assert g.shape == r.shape, (k, g.shape, r.shape)
        np.testing.assert_array_equal(g, r, err_msg="output %d" % k)
    d.close()


@pytest.mark.parametrize("name,over", [("topk_mc", dict(FULL_MC, **TOPK)), ("topk_lossatt", dict(LOSS_ATT, **TOPK)),
                                       ("topk_headmc", dict(HEAD_MC, **TOPK))])
def test_topk_global_postprocess_bit_exact(name, over):
    """a10: max_nms_inputs > 0 (the eval-time setting, eval.py:75) in global mode."""
    from oracle import post_ref as P, preprocess_ref as PP
    p = make_params(**over)
    w = make_weights(p, seed=19, cls_spread=20.0)
    x, scales = PP.preprocess(make_images(2, 100, 180, seed=20), (128, 192), p["mean_rgb"], p["stddev_rgb"])
    (rcls, rbox), _ = _oracle_net(p, w, x, 41)
    want = P.postprocess_global(p, rcls, rbox, scales)
    d = _driver(p, w, 2)
    got = d.postprocess(rcls, rbox, scales)
    assert len(got) == len(want)
    for k, (g, r) in enumerate(zip(got, want)):
        assert g.shape == r.shape, (k, g.shape, r.shape)
        np.testing.assert_array_equal(g, r, err_msg="output %d" % k)
    d.close()


def test_topk_kernel_orders_ties_by_index():
    """top-k pre-selection on logits with many exact ties (random-init regime)."""
    from oracle import post_ref as P, preprocess_ref as PP
    p = make_params(**dict(PLAIN, **TOPK))
    w = make_weights(p, seed=23)
    x, scales = PP.preprocess(make_images(1, 128, 192, seed=24), (128, 192), p["mean_rgb"], p["stddev_rgb"])
    (rcls, rbox), _ = _oracle_net(p, w, x, 0)
    rcls = [np.round(c * 4) / 4 for c in rcls]            # quantise: thousands of exact ties
    want = P.pre_nms(p, rcls, rbox)
    d = _driver(p, w, 1)
    d.postprocess(rcls, rbox, scales)
    got = d.candidates(1)
    np.testing.assert_array_equal(got["classes"], want["classes"])
    np.testing.assert_array_equal(got["scores"], want["scores"])
    np.testing.assert_array_equal(got["boxes"], want["boxes"])
    d.close()


def test_legacy_detection_rows():
    """a19: generate_detections / transform_detections row formats."""
    from oracle import post_ref as P, preprocess_ref as PP
    from uda_amd import postprocess as legacy
    p = make_params(**LOSS_ATT)
    w = make_weights(p, seed=29, cls_spread=20.0)
    x, scales = PP.preprocess(make_images(2, 100, 180, seed=30), (128, 192), p["mean_rgb"], p["stddev_rgb"])
    (rcls, rbox), _ = _oracle_net(p, w, x, 0)
    d = _driver(p, w, 2)
    rows = legacy.generate_detections(p, rcls, rbox, scales, np.array([7, 9]), per_class_nms=True, driver=d)
    b, s, c, v = P.postprocess_per_class(p, rcls, rbox, scales)
    assert rows.shape == (2, 100, 7)
    np.testing.assert_array_equal(rows[..., 0], np.array([[7.0], [9.0]], np.float32) * np.ones((2, 100), np.float32))
    np.testing.assert_array_equal(rows[..., 1:5], b[..., [1, 0, 3, 2]])
    np.testing.assert_array_equal(rows[..., 5], s)
    np.testing.assert_array_equal(rows[..., 6], c)
    xywh = legacy.transform_detections(rows)
    np.testing.assert_array_equal(xywh[..., 3], rows[..., 3] - rows[..., 1])
    rows_g = legacy.generate_detections(p, rcls, rbox, scales, np.array([7, 9]), per_class_nms=False)
    assert rows_g.shape == (2, 100, 7 + 7)
    d.close()


SHARD_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch.distributed as dist
from common import FULL_MC, make_images, make_params, make_weights
from uda_amd import dist as udist
from uda_amd.infer_lib import KerasDriver, ServingDriver
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
p = make_params(**FULL_MC); w = make_weights(p, seed=33, cls_spread=20.0)
imgs = make_images(3, 100, 180, seed=34)
drv = KerasDriver("_", False, p["name"], batch_size=3, model_params=p, weights=w)
drv.set_dropout_seed(11)
got = udist.serve_sharded(drv, imgs, rank, world)
np.savez(sys.argv[1] + ".rank%%d.npz" %% rank, *got)
dist.barrier(); dist.destroy_process_group(); drv.close()
'''


def test_image_sharded_serve_equals_unsharded(tmp_path):
    """e: two ranks (gloo, both on GPU 0) each serve their contiguous image shard and all-gather the
    detections; with the global image offset fed to the Philox stream the result is bit-identical
    to one process serving the whole batch."""
    import os, socket, subprocess, sys
    from common import ROOT
    p = make_params(**FULL_MC)
    w = make_weights(p, seed=33, cls_spread=20.0)
    imgs = make_images(3, 100, 180, seed=34)
    d = _driver(p, w, 3)
    d.set_dropout_seed(11)
    want = d.serve(imgs)
    d.close()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "det")
    procs = []
    world = 2
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", SHARD_WORKER % {"root": ROOT}, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for pr in procs:
        o = pr.communicate(timeout=300)[0]
        assert pr.returncode == 0, o
    for rank in range(world):
        z = np.load(out + ".rank%d.npz" % rank)
        got = [z["arr_%d" % i] for i in range(len(want))]
        for g, r in zip(got, want):
            np.testing.assert_array_equal(g, r)


def test_deep_ensemble_matches_oracle_aggregation():
    """BASELINE configs[3]: M independently initialised members, aggregated like MC samples."""
    from oracle import effdet_ref as E, post_ref as P, preprocess_ref as PP
    from uda_amd.infer_lib import EnsembleDriver
    p = make_params(**LOSS_ATT)
    ws = [make_weights(p, seed=40 + m, cls_spread=20.0) for m in range(3)]
    imgs = make_images(2, 100, 180, seed=44)
    ens = EnsembleDriver(ws, p["name"], batch_size=2, model_params=p)
    got = ens.serve(imgs)
    x, scales = PP.preprocess(imgs, (128, 192), p["mean_rgb"], p["stddev_rgb"])
    outs = [E.forward_once(w, p, x) for w in ws]
    pm = dict(p, mc_dropout=True, mc_dropoutrate=1e-9, mc_dropoutsamp=3)
    # bit-exact aggregation given the members' GPU heads
    heads = [m.head_outputs(2) for m in ens.members]
    cls_g = [np.stack([heads[m][0][l] for m in range(3)]) for l in range(5)]
    box_g = [np.stack([heads[m][1][l] for m in range(3)]) for l in range(5)]
    want_exact = P.postprocess_global(pm, cls_g, box_g, scales)
    assert len(got) == len(want_exact) and got[0].shape == (2, 100, 12) and got[2].shape == (2, 100, 8)
    for g, r in zip(got, want_exact):
        np.testing.assert_array_equal(g, r)
    # and the members' heads match the oracle networks
    for m in range(3):
        for l in range(5):
            for g, r in ((heads[m][0][l], outs[m][0][l]), (heads[m][1][l], outs[m][1][l])):
                assert np.abs(g - r).max() <= 2e-4 * np.abs(r).max() + 1e-6
    ens.close()


STRIPE_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch.distributed as dist
from common import LOSS_ATT, make_images, make_params, make_weights
from uda_amd import dist as udist
from uda_amd.infer_lib import KerasDriver, ServingDriver
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
p = make_params(**LOSS_ATT)
M = int(os.environ.get("UDA_TEST_MEMBERS", "3"))
imgs = make_images(3, 100, 180, seed=44)
mine = {m: KerasDriver("_", False, p["name"], batch_size=3, model_params=p, weights=make_weights(p, seed=40 + m, cls_spread=20.0))
        for m in range(M) if udist.member_owner(m, world) == rank}
pm = dict(p, mc_dropout=True, mc_dropoutrate=1e-9, mc_dropoutsamp=M)
post = KerasDriver("_", False, p["name"], batch_size=3, model_params=pm, weights=make_weights(p, seed=40, cls_spread=20.0), chunk_images=1)
got = udist.serve_ensemble_striped(mine, post, imgs, M, rank, world)
np.savez(sys.argv[1] + ".rank%%d.npz" %% rank, *got)
dist.barrier(); dist.destroy_process_group()
for d in list(mine.values()) + [post]:
    d.close()
'''


@pytest.mark.parametrize("members,world", [(3, 2), (2, 3)], ids=["3-members-2-ranks", "2-members-3-ranks"])
def test_ensemble_striped_over_ranks_equals_single_process(tmp_path, members, world):
    """BASELINE configs[3] across ranks: members striped over the ranks (gloo, all on GPU 0), heads re-sharded by
    image, aggregate + NMS per shard, all-gather: bit-identical to the single-process EnsembleDriver.  With fewer members
    than ranks (as `bench.py --gpus 8 --config 3`: 5 members, 8 ranks) the last rank owns no member and still receives,
    aggregates and gathers its image shard."""
    import os, socket, subprocess, sys
    from common import ROOT
    from uda_amd.infer_lib import EnsembleDriver
    p = make_params(**LOSS_ATT)
    ws = [make_weights(p, seed=40 + m, cls_spread=20.0) for m in range(members)]
    imgs = make_images(3, 100, 180, seed=44)
    ens = EnsembleDriver(ws, p["name"], batch_size=3, model_params=p)
    want = ens.serve(imgs)
    ens.close()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "ens")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1", UDA_TEST_MEMBERS=str(members))
        procs.append(subprocess.Popen([sys.executable, "-c", STRIPE_WORKER % {"root": ROOT}, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for pr in procs:
        o = pr.communicate(timeout=300)[0]
        assert pr.returncode == 0, o
    for rank in range(world):
        z = np.load(out + ".rank%d.npz" % rank)
        got = [z["arr_%d" % i] for i in range(len(want))]
        for g, r in zip(got, want):
            np.testing.assert_array_equal(g, r)


def test_bdd_like_padding_and_ten_classes():
    """BASELINE configs[2]-shaped input: 10 classes, raw height below the network height (scale 1, zero rows padded)."""
    from oracle import post_ref as P, preprocess_ref as PP
    p = make_params(num_classes=10, **FULL_MC)
    w = make_weights(p, seed=51, cls_spread=20.0)
    imgs = make_images(2, 120, 192, seed=52)
    d = _driver(p, w, 2)
    d.set_dropout_seed(3)
    det = d.serve(imgs)
    got, scales = d.preprocessed()
    x, wscales = PP.preprocess(imgs, (128, 192), p["mean_rgb"], p["stddev_rgb"])
    assert np.all(scales == 1.0) and np.array_equal(got, x) and np.all(got[:, 120:] == 0)
    cls, box = d.head_outputs(2)
    assert cls[0].shape[-1] == 90 and det[2].shape == (2, 100, 11) and det[4].shape == (2, 100, 10)
    (rcls, rbox), _ = _oracle_net(p, w, x, 3)
    _check_heads(cls, rcls)
    want = P.postprocess_global(p, cls, box, wscales)
    for g, r in zip(det, want):
        np.testing.assert_array_equal(g, r)
    d.close()


def test_d2_per_class_topk_config5_shape():
    """BASELINE configs[4]-shaped: D2, MC dropout, l-norm decode, per-class NMS with max_nms_inputs (eval settings)."""
    from oracle import post_ref as P, preprocess_ref as PP
    over = dict(FULL_MC, **TOPK)
    p = make_params(model="efficientdet-d2", image_size="128x128", **over)
    w = make_weights(p, seed=61, cls_spread=20.0)
    imgs = make_images(1, 128, 128, seed=62)
    d = _driver(p, w, 1)
    d.set_dropout_seed(8)
    det = d.serve(imgs, post_mode="per_class")
    cls, box = d.head_outputs(1)
    x, scales = PP.preprocess(imgs, (128, 128), p["mean_rgb"], p["stddev_rgb"])
    (rcls, rbox), _ = _oracle_net(p, w, x, 8)
    _check_heads(cls, rcls)
    want = P.postprocess_per_class(p, cls, box, scales)
    for g, r in zip(det, want):
        np.testing.assert_array_equal(g, r)
    d.close()


def test_class_probs_entropy_and_unpacking():
    """SURVEY 8f.1 on the device: softmax / entropy of the selected rows against the callers' numpy code."""
    from oracle import unpack_ref as U
    p = make_params(**FULL_MC)
    w = make_weights(p, cls_spread=20.0)
    d = _driver(p, w, 2)
    d.set_dropout_seed(11)
    out = d.serve_unpacked(make_images(2, 128, 192))
    det = d._collect(2)
    for n in range(2):
        want_p, want_h = U.probab_entropy(det[4][n])
        np.testing.assert_allclose(out["probab"][n], want_p, rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(out["entropy"][n], want_h, rtol=1e-5, atol=2e-6)
    b4, cid, al, mc, mcc = U.unpack(p, det[0], det[2])
    for k, v in (("boxes", b4), ("classes", cid), ("albox", al), ("mcbox", mc), ("mcclass", mcc)):
        np.testing.assert_array_equal(out[k], v, err_msg=k)
    d.close()


@pytest.mark.parametrize("env", [dict(UDA_PW_SCHEME="f32"), dict(UDA_PW_SCHEME="bf16x3"), dict(UDA_PW_SCHEME="bf16x2"),
                                 dict(UDA_PW_TERMS="6"),
                                 dict(UDA_FUSE_MBXD="0", UDA_FUSE_SEP="0", UDA_DEFER_DROPOUT="0"),
                                 dict(UDA_FUSE_MBX="0", UDA_POST_OVERLAP="0"), dict(UDA_FUSE_PROJ="0", UDA_PW_SHARED="0"),
                                 dict(UDA_F16_MIN_RMS="1e9")],
                         ids=["f32-mfma", "bf16x3", "bf16x2", "legacy-terms-switch", "no-deep-fusion", "unfused-serial-post",
                              "no-absorbed-projection", "fp16-unfit-ops-demoted-to-bf16x3"])
def test_fallback_paths_stay_parity_green(env):
    """Every switchable path (exact-f32 MFMA kernels, the bf16 split schemes beside the default fp16 one, each fusion off,
    fused MBConv ops whose weights do not suit fp16 pieces kept on three bf16 pieces) passes the smoke parity check.
    The switches are read once per process, so each configuration runs in its own interpreter."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    if "UDA_PW_TERMS" in env:
        e.pop("UDA_PW_SCHEME", None)          # the older switch only speaks when the newer one is silent
    e.update(env)
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=root, env=e,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


NMS_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import make_params, make_weights
from uda_amd.infer_lib import KerasDriver, ServingDriver
from oracle import post_ref as P
p = make_params()
d = KerasDriver("_", False, p["name"], batch_size=4, model_params=p, weights=make_weights(p))
rng = np.random.default_rng(7)
for n, tied, sigma, thr in ((70000, True, 0.25, 0.001), (70000, False, 0.25, 0.001), (40000, False, 0.0, 0.3),
                            (3000, True, 0.25, 0.001), (6000, False, 0.5, 0.2)):
    boxes = np.zeros((3, n, 4), np.float32); scores = np.zeros((3, n), np.float32)
    for i in range(3):
        c = rng.uniform(0, 600.0, (n, 2)); wh = rng.uniform(4, 120, (n, 2))
        boxes[i] = np.concatenate([c - wh / 2, c + wh / 2], 1)
        if tied:
            scores[i] = 0.01 + rng.normal(0, 1e-4, n)
            scores[i, rng.integers(0, n, n // 8)] = scores[i, 0]
        else:
            scores[i] = rng.uniform(0, 1, n)
    idx, sc, valid = d.nms(boxes, scores, 100, 0.5, thr, sigma)
    for i in range(3):
        ridx, rsc, rvalid = P.nms_v5(boxes[i], scores[i], 100, 0.5, thr, sigma, True)
        assert valid[i] == rvalid, (n, tied, i, valid[i], rvalid)
        assert (idx[i] == ridx).all() and (sc[i] == rsc).all(), (n, tied, i)
print("nms paths ok, redone", d.nms_prefix_fallbacks())
d.close()
"""


@pytest.mark.parametrize("env", [dict(), dict(UDA_NMS_PREFIX="0", UDA_NMS_COOP="0"), dict(UDA_NMS_COOP="0"),
                                 dict(UDA_NMS_REG="0", UDA_NMS_COOP="0")],
                         ids=["co-resident-grid", "two-launches-per-epoch", "score-prefix", "global-state-solo"])
def test_nms_paths_bit_exact(env):
    """Every NMS execution path - score prefix + register kernel (default), the cooperative single launch over several
    blocks per problem, the two-launches-per-epoch grid version, the earlier single-launch kernel - against the
    oracle's NonMaxSuppressionV5 on 70 000 / 40 000 / 6 000 / 3 000 candidates (near-tied and uniform scores, soft and hard).
    The switches are read once per process, so each configuration runs in its own interpreter."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", NMS_WORKER % {"root": root}], cwd=root, env=e, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "nms paths ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


OVERSUB_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import make_params, make_weights
from uda_amd.infer_lib import KerasDriver, ServingDriver
from oracle import post_ref as P
p = make_params()
n_img, n = 64, 150000          # 64 problems x 5 blocks = 320 blocks of 1024 threads, one per CU: more than the device holds
d = KerasDriver("_", False, p["name"], batch_size=n_img, model_params=p, weights=make_weights(p))
rng = np.random.default_rng(1)
c = rng.uniform(0, 600.0, (n_img, n, 2)); wh = rng.uniform(4, 120, (n_img, n, 2))
boxes = np.concatenate([c - wh / 2, c + wh / 2], 2).astype(np.float32)
scores = rng.uniform(0, 1, (n_img, n)).astype(np.float32)
idx, sc, valid = d.nms(boxes, scores, 100, 0.5, 0.001, 0.25)
for i in (0, 31, 63):
    ridx, rsc, rvalid = P.nms_v5(boxes[i], scores[i], 100, 0.5, 0.001, 0.25, True)
    assert valid[i] == rvalid and (idx[i] == ridx).all() and (sc[i] == rsc).all(), i
print("oversubscribed ok")
d.close()
"""


def test_nms_grid_larger_than_the_device_completes():
    """The single-launch NMS needs the blocks of a problem resident together.  With the capacity check overridden the
    grid is larger than the device: it must still complete (blocks are dispatched in order, finished problems free their
    CUs; the bounded spin is the safety net) and return the oracle's selections."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, UDA_NMS_COOP_CAP="100000")
    r = subprocess.run([sys.executable, "-c", OVERSUB_WORKER % {"root": root}], cwd=root, env=e, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "oversubscribed ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_box_uncertainty_calibration_matches_reference_restatement():
    """SURVEY 8f.2 on the device: every calibrate_boxuncert method on the selected rows against the numpy restatement."""
    from oracle import calib_ref as CR
    from uda_amd.calibration import BoxCalibrator, IsoTable
    p = make_params(**FULL_MC)
    w = make_weights(p, cls_spread=20.0)
    d = _driver(p, w, 2)
    d.set_dropout_seed(5)
    det = d.serve(make_images(2, 128, 192))
    rng = np.random.default_rng(9)

    def table():
        xs = np.sort(rng.uniform(0, 40, 12))
        xs += np.arange(12) * 1e-3
        return xs, np.sort(rng.uniform(0, 60, 12))
    C = p["num_classes"]
    models = dict(ts_all=1.7, ts_percoo=[1.1, 2.3, 0.7, 3.1], iso_all=table(), iso_percoo=[table() for _ in range(4)],
                  iso_perclscoo=[table() for _ in range(4 * C)],
                  rel_iso_perclscoo=[(np.sort(rng.uniform(0, 2, 9)) + np.arange(9) * 1e-3, np.sort(rng.uniform(0, 3, 9)))
                                     for _ in range(4 * C)])
    dev_models = {k: (v if k.startswith("ts") else ([IsoTable(*t) for t in v] if isinstance(v, list) else IsoTable(*v)))
                  for k, v in models.items()}
    cal = BoxCalibrator(d, dev_models)
    for which, cols in (("albox", slice(4, 8)), ("mcbox", slice(8, 12))):
        for method in models:
            got = cal.calibrate_boxuncert(2, which, method)
            for n in range(2):
                want = CR.calibrate_boxuncert(method, models, C, det[0][n][:, cols], det[2][n][:, 0], det[0][n][:, :4])
                np.testing.assert_allclose(got[n], want, rtol=2e-6, atol=1e-6, err_msg="%s %s" % (which, method))
    with pytest.raises(ValueError):
        cal.calibrate_boxuncert(2, "albox", "nonsense")
    d.close()


def test_class_calibration_matches_reference_restatement():
    """SURVEY 8f.2, class half, on the device: every `_perform_class_calib` method (utils_class.py:109-187) on the selected
    rows against the numpy restatement (pinned against sklearn in tests/test_oracle_kats.py) - without MC class
    uncertainty (mean logits) and with it (10 Philox-normal draws per logit, mean / std of the calibrated probabilities)."""
    from oracle import calib_ref as CR
    from uda_amd.calibration import ClassCalibrator, IsoTable
    from common import LOSS_ATT
    rng = np.random.default_rng(19)

    def table():
        xs = np.sort(rng.uniform(0, 1, 14)) + np.arange(14) * 1e-4
        return xs, np.sort(rng.uniform(0, 1, 14))
    for over, with_unc in ((LOSS_ATT, False), (FULL_MC, True)):
        p = make_params(**over)
        C = p["num_classes"]
        w = make_weights(p, cls_spread=20.0)
        d = _driver(p, w, 2)
        d.set_dropout_seed(5)
        det = d.serve(make_images(2, 128, 192))
        models = dict(ts_all=1.9, ts_percls=np.linspace(0.7, 2.2, C), iso_all=table(), iso_percls=[table() for _ in range(C)])
        dev = dict(ts_all=1.9, ts_percls=models["ts_percls"], iso_all=IsoTable(*models["iso_all"]),
                   iso_percls=[IsoTable(*t) for t in models["iso_percls"]])
        cal = ClassCalibrator(d, dev, calib_method="iso_percls", draws=10, seed=77)
        logits = det[4].reshape(-1, C)
        unc = det[2][..., 1:].reshape(-1, C) if with_unc else None
        for method in ("ts_all", "ts_percls", "iso_all", "iso_percls"):
            got = cal.perform_class_calib(2, method)
            want = CR.perform_class_calib(method, models, logits, unc, draws=10, seed=77)
            assert len(got) == len(want) == (3 if with_unc else 2)
            np.testing.assert_allclose(got[1].reshape(-1, C), want[1], rtol=5e-5, atol=2e-6, err_msg=method)
            np.testing.assert_allclose(got[0].reshape(-1), want[0], rtol=1e-4, atol=1e-5, err_msg=method)
            if with_unc:
                np.testing.assert_allclose(got[2].reshape(-1, C), want[2], rtol=1e-3, atol=2e-6, err_msg=method)
                assert (got[2] > 0).any()
        full = cal.calibrate_class(2)
        assert len(full) == (14 if with_unc else 9)             # the reference's return tuple (utils_class.py:247-272)
        sel = cal.perform_class_calib(2, "iso_percls")
        np.testing.assert_array_equal(full[1 if with_unc else 0], sel[0])
        with pytest.raises(ValueError):
            cal.perform_class_calib(2, "nonsense")
        d.close()
