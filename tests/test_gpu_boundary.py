"""GPU tests of the drop-in boundary with the reference's own call expressions (through the C ABI):
driver factory / constructors (src/infer_lib.py:154-163,299-311,416-440), the eval.py flow
(src/eval.py:98-123: EfficientDetNet -> mc_eval -> generate_detections -> transform_detections), mc_infer,
input validation of injected head outputs, and the single-launch NMS fallback counter."""
import os
import subprocess
import sys

import numpy as np
import pytest

from common import FULL_MC, HEAD_MC, LOSS_ATT, ROOT, make_images, make_params, make_weights

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_factory_and_constructors_with_reference_call_expressions(tmp_path):
    from uda_amd import infer_lib, weights as W
    model_params = make_params(**FULL_MC)
    w = W.init_weights(model_params, seed=0)          # what "_" resolves to (uda_seed defaults to 0)
    npz = str(tmp_path / "d0.npz")
    W.save_weights(npz, w)
    imgs = make_images(2, 100, 180, seed=3)
    # inspector.py:161-169
    driver = infer_lib.ServingDriver.create("_", False, None, "efficientdet-d0", 2, False, model_params)
    assert type(driver) is infer_lib.KerasDriver and driver.batch_size == 2 and driver.only_network is False
    assert driver.params["is_training_bn"] is False and driver.label_map is None
    driver.set_dropout_seed(4)
    want = driver.serve(imgs)
    driver.close()
    # infer_lib.py:299-311 - positional (path, model_name, batch_size, only_network, model_params)
    saved = infer_lib.SavedModelDriver(npz, "efficientdet-d0", 2, False, model_params)
    assert saved.model_name == "efficientdet-d0" and saved.batch_size == 2
    saved.set_dropout_seed(4)
    _same(saved.serve(imgs), want)
    saved.close()
    via = infer_lib.ServingDriver.create("_", False, npz, "efficientdet-d0", 2, False, model_params)
    assert type(via) is infer_lib.SavedModelDriver
    via.close()
    # infer_lib.py:416-440 - positional (ckpt_path, debug, model_name, batch_size, only_network, model_params)
    keras = infer_lib.KerasDriver(npz, False, "efficientdet-d0", 2, True, model_params)
    keras.set_dropout_seed(4)
    _same(keras.serve(imgs), want)                      # only_network: serve = host-visible pre + predict + post, same tuple
    out = keras.visualize(imgs[0], want[0][0][:, :4], want[2][0][:, 0], want[1][0], uncertainty=want[0][0][:, 4:8],
                          min_score_thresh=0.0)
    assert out.shape == imgs[0].shape and out.dtype == np.uint8 and (out != imgs[0]).any()
    t = keras.benchmark(np.zeros((2, 128, 192, 3), np.float32), bm_runs=2, trace_filename=str(tmp_path / "trace.json"))
    assert t > 0 and os.path.getsize(str(tmp_path / "trace.json")) > 10
    keras.close()


def test_injected_head_outputs_are_validated():
    """A wrong rank / sample axis / level shape must raise, never read past the host buffer."""
    from uda_amd.infer_lib import KerasDriver
    p = make_params(**FULL_MC)
    d = KerasDriver("_", False, p["name"], 2, False, p, weights=make_weights(p))
    hw = d.plan.level_hw
    cls = [np.zeros((3, 2, h, w, 63), np.float32) for h, w in hw]
    box = [np.zeros((3, 2, h, w, 72), np.float32) for h, w in hw]
    d.postprocess(cls, box, np.ones(2, np.float32))                                   # well-formed
    with pytest.raises(ValueError):
        d.postprocess([c[0] for c in cls], box, np.ones(2, np.float32))              # unstacked array for a stacked head
    with pytest.raises(ValueError):
        d.postprocess(cls, [b[:2] for b in box], np.ones(2, np.float32))             # T = 2 instead of 3
    with pytest.raises(ValueError):
        d.postprocess(cls[:4], box[:4], np.ones(2, np.float32))                      # a level missing
    with pytest.raises(ValueError):
        d.postprocess(cls, [b[..., :36] for b in box], np.ones(2, np.float32))       # loss-attenuation half missing
    with pytest.raises(ValueError):
        d.postprocess(cls, box, np.ones(3, np.float32))                              # scales of another batch size
    # the C entry point itself refuses a float count that does not match its layout
    c0 = np.zeros((2,) + hw[0] + (63,), np.float32)
    rc = d._lib.uda_set_head_outputs(d._h, 0, 2, c0.ctypes.data, c0.size, None, 0)
    assert rc != 0 and b"expects" in d._lib.uda_last_error(d._h)
    d.close()


def test_eval_flow_model_mc_eval_generate_detections():
    """src/eval.py:98-127 by import switch only: build the model object, restore "_", mc_eval, generate_detections
    (per-class NMS, the eval default) and transform_detections; rows equal the oracle's on the same head outputs, and
    the resident (no host hop) and uploaded routes agree bit for bit."""
    from oracle import post_ref as P
    from uda_amd import efficientdet_keras, hparams_config, postprocess, utils_keras
    from uda_amd.utils_extra import mc_eval
    p = make_params(**dict(FULL_MC, nms_configs=dict(method="gaussian", iou_thresh=None, score_thresh=0.0, sigma=None,
                                                      pyfunc=False, max_nms_inputs=500, max_output_size=100)))
    config = hparams_config.Config(p)
    config.is_training_bn = True                  # eval.py leaves the default: the model call itself is one pass
    model = efficientdet_keras.EfficientDetNet(config=config)
    model.build((None, 128, 192, 3))
    utils_keras.restore_ckpt(model, "_", config.moving_average_decay, skip_mismatch=False)
    images = np.random.default_rng(5).normal(size=(2, 128, 192, 3)).astype(np.float32)
    image_scales = np.array([1.0, 1.5], np.float32)
    source_ids = np.array([11, 12], np.float32)
    cls_outputs, box_outputs = mc_eval(model, images, config)
    detections = postprocess.generate_detections(config, cls_outputs, box_outputs, image_scales, source_ids)
    detections = postprocess.transform_detections(detections)
    assert detections.shape == (2, 100, 7) and np.all(detections[..., 0] == source_ids[:, None])
    # the same through host arrays (download, then the cached post-process-only handle)
    cls_np, box_np = [np.array(c) for c in cls_outputs], [np.array(b) for b in box_outputs]
    assert cls_np[0].shape == (3, 2, 16, 24, 63) and box_np[0].shape == (3, 2, 16, 24, 72)
    rows2 = postprocess.transform_detections(postprocess.generate_detections(config, cls_np, box_np, image_scales, source_ids))
    np.testing.assert_array_equal(rows2[..., :7], detections[..., :7])
    b, s, c, v = P.postprocess_per_class(p, cls_np, box_np, image_scales)
    np.testing.assert_array_equal(detections[..., 1], b[..., 1])
    np.testing.assert_array_equal(detections[..., 2], b[..., 0])
    np.testing.assert_array_equal(detections[..., 3], b[..., 3] - b[..., 1])
    np.testing.assert_array_equal(detections[..., 4], b[..., 2] - b[..., 0])
    np.testing.assert_array_equal(detections[..., 5], s)
    np.testing.assert_array_equal(detections[..., 6], c)
    # a single model call: one stochastic pass, unstacked, as `EfficientDetNet.call` with is_training_bn
    c1, b1 = model(images, training=False)
    assert np.array(c1[0]).shape == (2, 16, 24, 63) and np.array(b1[4]).shape == (2, 1, 2, 72)
    # ... and it is the oracle's network under the masks of that handle's first call (seed 0, T = 1)
    from oracle import effdet_ref as E, philox_ref as R
    p1 = dict(p, mc_dropoutsamp=1)
    w = next(iter(model._drivers.values())).weights
    rc, rb = E.forward(w, p1, images, R.make_masks(E.dropout_sites(p1), 0, 2, 1))
    for l in range(5):
        for g, r in ((np.array(c1[l]), np.asarray(rc[l])), (np.array(b1[l]), np.asarray(rb[l]))):
            r = r.reshape(g.shape)
            assert np.abs(g - r).max() <= 2e-4 * np.abs(r).max() + 1e-6
    # postprocess_global / postprocess_per_class module functions with the reference's argument order
    g = postprocess.postprocess_global(p, cls_np, box_np, image_scales)
    want = P.postprocess_global(p, cls_np, box_np, image_scales)
    _same(g, want)
    model.close()


def test_full_model_object_and_mc_infer():
    """EfficientDetModel.call(inputs, training, pre_mode, post_mode) (efficientdet_keras.py:1118-1146) and the
    serving-level MC twin mc_infer (utils_extra.py:119-139)."""
    from uda_amd import efficientdet_keras, hparams_config
    from uda_amd.infer_lib import KerasDriver
    from uda_amd.utils_extra import mc_infer
    p = make_params(**HEAD_MC)
    w = make_weights(p, seed=2, cls_spread=20.0)
    imgs = make_images(2, 100, 180, seed=8)
    model = efficientdet_keras.EfficientDetModel(config=hparams_config.Config(p), weights=w)
    d = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
    got = model(imgs, training=False)
    # both draw seeds 0, 1, ... per call: the first call of each handle uses seed 0
    want = d.serve(imgs)
    _same(got, want)
    pc = model(imgs, training=False, pre_mode="infer", post_mode="per_class")
    assert len(pc) == 4 and pc[0].shape == (2, 100, 4)
    with pytest.raises(ValueError):
        model(imgs, post_mode="tflite")
    d.set_dropout_seed(None)
    out = mc_infer(d, imgs, T=3)
    assert out[0].shape == (3, 2, 100, 12) and out[3].shape == (3, 2)
    assert not np.array_equal(out[0][0], out[0][1])         # every serve call draws new masks
    d.close()
    model.close()


COOP_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import FULL_MC, make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
from oracle import effdet_ref as E, philox_ref as R, post_ref as P, preprocess_ref as PP
expect_fallback = %(expect)d
p = make_params()
d = KerasDriver("_", False, p["name"], 4, False, p, weights=make_weights(p))
rng = np.random.default_rng(3)
n = 70000
boxes = np.zeros((3, n, 4), np.float32); scores = np.zeros((3, n), np.float32)
for i in range(3):
    c = rng.uniform(0, 600.0, (n, 2)); wh = rng.uniform(4, 120, (n, 2))
    boxes[i] = np.concatenate([c - wh / 2, c + wh / 2], 1)
    scores[i] = 0.01 + rng.normal(0, 1e-4, n)
for rep in range(2):            # after a time-out the handle stays on the two-launch version: same results
    idx, sc, valid = d.nms(boxes, scores, 100, 0.5, 0.001, 0.25)
    for i in range(3):
        ridx, rsc, rvalid = P.nms_v5(boxes[i], scores[i], 100, 0.5, 0.001, 0.25, True)
        assert valid[i] == rvalid and (idx[i] == ridx).all() and (sc[i] == rsc).all(), (rep, i)
fb_nms = d.nms_coop_fallbacks()
d.close()
# the same inside the serve post-process: 384x512 input = 36 828 candidates per image (2 blocks of 32 768 per problem)
p = make_params(image_size="512x384", **FULL_MC)
w = make_weights(p, seed=33)
x, scales = PP.preprocess(make_images(2, 300, 480, seed=32), (384, 512), p["mean_rgb"], p["stddev_rgb"])
masks = R.make_masks(E.dropout_sites(p), 5, 2, 3)
rcls, rbox = E.forward(w, p, x, masks)
want = P.postprocess_global(p, rcls, rbox, scales)
d = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
for rep in range(2):
    got = d.postprocess(rcls, rbox, scales)
    for g, r in zip(got, want):
        assert np.array_equal(g, r), rep
fb_post = d.nms_coop_fallbacks()
d.close()
print("coop fallbacks", fb_nms, fb_post)
assert (fb_nms >= 1 and fb_post >= 1) if expect_fallback else (fb_nms == 0 and fb_post == 0), (fb_nms, fb_post)
print("coop ok")
"""


@pytest.mark.parametrize("spin,expect", [("1", 1), (None, 0)], ids=["forced-time-out", "default"])
def test_single_launch_nms_time_out_path_is_observable(spin, expect):
    """UDA_NMS_COOP_SPIN=1 makes the first unanswered poll of an exchange slot a time-out: the kernel raises the error
    word, every block drains, the host redoes the post-process with two launches per epoch and counts it
    (uda_nms_coop_fallbacks); outputs equal the oracle's either way.  Without the knob the counter stays 0."""
    e = dict(os.environ)
    if spin is not None:
        e["UDA_NMS_COOP_SPIN"] = spin
    r = subprocess.run([sys.executable, "-c", COOP_WORKER % {"root": ROOT, "expect": expect}], cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "coop ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


NCCL_WORKER = r'''
import os, sys
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch                     # torch first: the HIP library then binds to the same runtime
import torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
from common import LOSS_ATT, make_images, make_params, make_weights
from uda_amd import dist as udist
from uda_amd.infer_lib import EnsembleDriver, ServingDriver
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", rank))
torch.cuda.set_device(local)
tdev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=tdev)
p = make_params(**LOSS_ATT)
M = 3
imgs = make_images(3, 100, 180, seed=44)
ws = [make_weights(p, seed=40 + m, cls_spread=20.0) for m in range(M)]
mine = {m: ServingDriver(p["name"], 3, False, p, weights=ws[m], device=local) for m in range(M) if udist.member_owner(m, world) == rank}
pm = dict(p, mc_dropout=True, mc_dropoutrate=1e-9, mc_dropoutsamp=M)
post = ServingDriver(p["name"], 3, False, pm, post_only=True, device=local, chunk_images=1)
got = udist.serve_ensemble_striped(mine, post, imgs, M, rank, world, device=tdev)
got2 = udist.serve_sharded(next(iter(mine.values())) if mine else ServingDriver(p["name"], 3, False, p, weights=ws[0], device=local), imgs, rank, world, device=tdev)
for d in list(mine.values()) + [post]:
    d.close()
if rank == 0:
    ens = EnsembleDriver(ws, p["name"], batch_size=3, model_params=p, device=local)
    want = ens.serve(imgs)
    ens.close()
    assert len(got) == len(want)
    for g, r in zip(got, want):
        assert np.array_equal(g, r), "device-resident ensemble exchange differs from the single-process ensemble"
    assert got2[0].shape[0] == 3
    print("nccl ensemble ok", world)
dist.barrier(); dist.destroy_process_group()
'''


def _run_nccl(world, tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER % {"root": ROOT})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "nccl ensemble ok %d" % world in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_ensemble_exchange_stays_on_the_device_under_rccl_world1(tmp_path):
    """e / x1: with an RCCL (backend "nccl") process group the member heads go from the members' device buffers into the
    aggregating handle's sample slots through torch views of the handles' own memory (no numpy hop); world size 1
    exercises the zero-copy plumbing and the device-to-device path on the one GPU of this box."""
    _run_nccl(1, tmp_path)


def test_ensemble_exchange_over_rccl_two_gpus(tmp_path):
    """The same with two ranks on two GPUs: batched point-to-point transfers over xGMI.  Skipped on a one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run_nccl(2, tmp_path)
