"""Round-5 GPU parity tests (-m gpu, all through the C ABI).

  * every `model_params` architecture switch the planner honours: head outputs against the CPU
    oracle run with the SAME switch - act_type relu / relu6 / hswish (stem, expand, depthwise, SE reduce, BiFPN nodes,
    head layers), conv_bn_act_pattern, conv_after_downsample, apply_bn_for_resampling=False, fpn_weight_method attn,
    and the whole serve tuple for one of them.
"""
import numpy as np
import pytest

from common import FULL_MC, HEAD_MC, LOSS_ATT, check_heads, make_images, make_params, make_weights

pytestmark = pytest.mark.gpu


def _driver(params, w, batch, **kw):
    from uda_amd.infer_lib import KerasDriver
    return KerasDriver("_", False, params["name"], batch_size=batch, only_network=kw.pop("only_network", False),
                       model_params=params, weights=w, **kw)


def _oracle_net(params, w, x, seed):
    from oracle import effdet_ref as E, philox_ref as R
    sites = E.dropout_sites(params)
    T = params["mc_dropoutsamp"] if params["mc_dropout"] else 1
    masks = R.make_masks(sites, seed, x.shape[0], T) if sites else None
    return E.forward(w, params, x, masks)


SWITCH_CASES = [
    ("relu", dict(act_type="relu", **FULL_MC)),
    ("relu6", dict(act_type="relu6", **HEAD_MC)),
    ("hswish", dict(act_type="hswish", **FULL_MC)),
    ("swish_native", dict(act_type="swish_native", **LOSS_ATT)),
    ("mish", dict(act_type="mish", **HEAD_MC)),
    ("conv_bn_act", dict(conv_bn_act_pattern=True, **FULL_MC)),
    ("conv_bn_act_relu6", dict(conv_bn_act_pattern=True, act_type="relu6", **LOSS_ATT)),
    ("conv_after_downsample", dict(conv_after_downsample=True, **HEAD_MC)),
    ("no_resample_bn", dict(apply_bn_for_resampling=False, **LOSS_ATT)),
    ("attn", dict(fpn_weight_method="attn", **HEAD_MC)),
    ("all_switches", dict(act_type="hswish", conv_bn_act_pattern=True, conv_after_downsample=True,
                          apply_bn_for_resampling=False, fpn_weight_method="sum", **FULL_MC)),
]


@pytest.mark.parametrize("name,over", SWITCH_CASES, ids=[c[0] for c in SWITCH_CASES])
def test_architecture_switches_match_the_oracle_run_with_the_same_switch(name, over):
    from oracle import preprocess_ref as PP
    p = make_params(uda_keep_buffers=True, **over)
    w = make_weights(p, seed=11)
    d = _driver(p, w, 2, only_network=True)
    x, _ = PP.preprocess(make_images(2, 128, 192, seed=13), d.image_size, p["mean_rgb"], p["stddev_rgb"])
    d.set_dropout_seed(91)
    cls, box = d.predict(x)
    rcls, rbox = _oracle_net(p, w, x, 91)
    check_heads(cls, rcls)
    check_heads(box, rbox)
    # The BiFPN of a randomly initialised network damps a perturbation of its inputs about 7x per node, so the heads alone
    # would not notice a wrong P6: every named activation of the device (arena recycling off: uda_keep_buffers) against the
    # oracle's taps of MC sample 0 - stem, every block's depthwise / gate / output, P6 / P7, every BiFPN node
    from oracle import effdet_ref as E, philox_ref as R
    sites = E.dropout_sites(p)
    T = p["mc_dropoutsamp"] if p["mc_dropout"] else 1
    masks = R.make_masks(sites, 91, 2, T) if sites else None
    taps = {}
    E.forward_once(w, p, x, masks, 0, taps)
    seen = 0
    for tap, ref in taps.items():
        if tap not in d.plan.buffer_names:
            continue
        if tap in ("blocks_0/dw", "blocks_0/se") and p["mc_dropout"] and p["mc_dropoutrate"]:
            continue                              # block 0's dropout site is deferred into its SE gate: other tensors by design
        got = d.read_buffer(tap, 2)
        if got.shape[0] == 2 * T and T > 1:
            got = got[0::T]                       # rows are [image][sample]: sample 0 of each image
        if tap.endswith("/se"):
            ref = ref.reshape(got.shape)
        assert got.shape == ref.shape, (tap, got.shape, ref.shape)
        err, scale = np.abs(got - ref).max(), np.abs(ref).max()
        assert err <= 1e-4 * scale + 1e-6, "%s: %g vs scale %g" % (tap, err, scale)
        seen += 1
    assert seen >= 40 and "p6_in" in d.plan.buffer_names and "cell0/fnode0/out" in d.plan.buffer_names
    # the switch is not a no-op: P6 of the default network on the same weights differs
    if name in ("relu", "hswish", "conv_after_downsample"):
        q = dict(p, act_type="swish", conv_after_downsample=False)
        d2 = _driver(q, w, 2, only_network=True)
        d2.set_dropout_seed(91)
        d2.predict(x)
        other = d2.read_buffer("p6_in", 2)
        d2.close()
        assert np.abs(other - d.read_buffer("p6_in", 2)).max() > 1e-2
    d.close()


def test_relu_network_serves_the_oracle_detections_on_a_u8_batch():
    """a non-swish act_type keeps the uint8 batch off the (swish) uint8 stem and the MBConv blocks unfused: the whole
    serve - preprocess, network, aggregate, decode, NMS - against the oracle chain."""
    from oracle import post_ref, preprocess_ref as PP
    p = make_params(act_type="relu", **FULL_MC)
    w = make_weights(p, seed=5, cls_spread=20.0)
    imgs = make_images(2, 128, 192, seed=3)       # scale 1: the uint8-stem route would apply under swish
    d = _driver(p, w, 2)
    d.set_dropout_seed(7)
    got = d.serve(imgs)
    cls, box = d.head_outputs(2)
    x, scales = PP.preprocess(imgs, (128, 192), p["mean_rgb"], p["stddev_rgb"])
    rcls, rbox = _oracle_net(p, w, x, 7)
    check_heads(cls, rcls)
    check_heads(box, rbox)
    want = post_ref.postprocess_global(p, rcls, rbox, scales)
    again = d.postprocess(rcls, rbox, scales)
    for g, r in zip(again, want):
        np.testing.assert_array_equal(g, r)
    assert np.array_equal(got[3], want[3])
    d.close()


def test_refused_switches_fail_before_anything_is_created():
    from uda_amd import plan as plan_mod
    for over in (dict(act_type="srelu"), dict(separable_conv=False), dict(fpn_weight_method="channel_attn"),
                 dict(data_format="channels_first"), dict(fpn_name="qufpn")):
        p = make_params(**over)
        with pytest.raises(ValueError):
            _driver(p, None, 1)
    # and the C side refuses an activation code it does not know / a non-swish fused MBConv op
    p = make_params()
    w = make_weights(p)
    pl = plan_mod.Plan(p, w, chunk_images=1, max_images=1)
    mbx = [o for o in pl.ops if o["kind"] == 7]
    assert mbx
    mbx[0]["act"] = 2
    from uda_amd import capi
    lib = capi.load()
    import ctypes as C
    m, bufs, ops, sites, blob, anchors = pl.to_c()
    h = C.c_void_p()
    rc = lib.uda_create(C.byref(m), bufs, len(bufs), ops, len(pl.ops), sites, C.c_void_p(blob.ctypes.data), blob.size,
                        C.c_void_p(anchors.ctypes.data), 0, C.byref(h))
    assert rc != 0 and b"swish kernel" in lib.uda_last_error(None)


# ------------------------------------------------------------------ several winners per grid-wide NMS step
WINNERS_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import make_params, make_weights
from uda_amd.infer_lib import KerasDriver
from oracle import post_ref as P
p = make_params()
d = KerasDriver("_", False, p["name"], batch_size=4, model_params=p, weights=make_weights(p))     # 4 x 100 outputs: scratch for 3 problems x 128
def boxes_scores(rng, n, case):
    span = 40000.0 if case == "sparse" else (1500.0 if case in ("scattered", "tied") else 400.0)
    c = rng.uniform(0, span, (n, 2)); wh = rng.uniform(4, 120, (n, 2))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    if case in ("tied", "dense_tied"):
        s = (0.01 + rng.normal(0, 1e-4, n)).astype(np.float32)
        s[rng.integers(0, n, n // 8)] = s[0]
    elif case == "clusters":        # confident objects with many anchors each: consecutive winners overlap
        s = rng.uniform(0.0, 0.05, n).astype(np.float32)
        for o in range(12):
            m = rng.integers(0, n, 40)
            b[m] = b[m[0]] + rng.normal(0, 3.0, (40, 4)).astype(np.float32)
            s[m] = rng.uniform(0.5, 0.95, 40).astype(np.float32)
    elif case == "big_boxes":       # every box covers most of the frame: nothing is ever certified beyond the first winner
        b = np.concatenate([c * 0.02, span - c * 0.02], 1).astype(np.float32)
        s = rng.uniform(0, 1, n).astype(np.float32)
    else:
        s = rng.uniform(0, 1, n).astype(np.float32)
    return b, s
for case, n, sigma, thr, m in (("sparse", 30000, 0.25, 0.001, 100), ("scattered", 30000, 0.25, 0.001, 100), ("tied", 30000, 0.25, 0.001, 100),
                               ("dense_tied", 20000, 0.25, 0.001, 100), ("clusters", 30000, 0.25, 0.001, 100), ("big_boxes", 12000, 0.25, 0.001, 60),
                               ("scattered", 30000, 0.0, float("-inf"), 100), ("dense_tied", 20000, 0.0, 0.005, 100), ("scattered", 9000, 0.3, 0.2, 7),
                               ("few_alive", 30000, 0.25, 0.9995, 100), ("sparse", 70000, 0.5, 0.001, 128)):
    rng = np.random.default_rng(len(case) + n)
    n_img = 3
    boxes = np.zeros((n_img, n, 4), np.float32); scores = np.zeros((n_img, n), np.float32)
    for i in range(n_img):
        boxes[i], scores[i] = boxes_scores(rng, n, case)
    idx, sc, valid = d.nms(boxes, scores, m, 0.5, thr, sigma)
    for i in range(n_img):
        ridx, rsc, rvalid = P.nms_v5(boxes[i], scores[i], m, 0.5, thr, sigma, True)
        assert valid[i] == rvalid, (case, i, valid[i], rvalid)
        assert (idx[i] == ridx).all(), (case, i, np.flatnonzero(idx[i] != ridx)[:5])
        assert (sc[i].view(np.uint32) == rsc.view(np.uint32)).all(), (case, i)
    print(case, n, "ok", valid.tolist())
assert d.nms_coop_fallbacks() == 0
print("winners ok")
d.close()
"""


@pytest.mark.parametrize("winners,ipt", [("1", None), ("2", None), ("4", "16"), ("8", "4"), (None, None), ("8", "32")])
def test_several_winners_per_nms_step_are_bit_exact(winners, ipt):
    """nms_coop_kernel settles up to UDA_NMS_WINNERS selections per pair of grid-wide exchanges (default: half the blocks of
    a problem, at most 8): keep-sets, order and float32 scores equal to the oracle's heap in every regime - scattered
    (most steps settle several), clustered / frame-filling boxes (the first overlap ends a step), exact ties, hard NMS,
    a score threshold that leaves fewer live candidates than winners asked for - and for every block shape."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.pop("UDA_NMS_WINNERS", None)
    e.pop("UDA_NMS_COOP_IPT", None)
    if winners:
        e["UDA_NMS_WINNERS"] = winners
    if ipt:
        e["UDA_NMS_COOP_IPT"] = ipt
    r = subprocess.run([sys.executable, "-c", WINNERS_WORKER % {"root": root}], cwd=root, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "winners ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ------------------------------------------------------------------ abandoned streams
def test_a_dropped_serve_stream_leaves_the_handle_usable():
    """serve_stream pipelines: when it yields batch k, batch k + 1 is already queued.  A consumer that breaks out, or drops
    the generator, must not leave that run open in the handle (every synchronous entry point refuses beside a run in
    flight): the generator drains it on the way out (`uda_drain`)."""
    p = make_params(**FULL_MC)
    w = make_weights(p, seed=3, cls_spread=20.0)
    d = _driver(p, w, 2)
    d.set_dropout_seed(11)
    batches = [make_images(2, 128, 192, seed=s) for s in (1, 2, 3, 4)]
    want = [d.serve(b) for b in batches]
    gen = d.serve_stream(batches)
    first = next(gen)
    gen.close()                                   # dropped after the first batch: batch 2 is in flight
    for g, r in zip(first, want[0]):
        np.testing.assert_array_equal(g, r)
    again = d.serve(batches[2])                   # would raise "a pipelined run is in flight" without the drain
    for g, r in zip(again, want[2]):
        np.testing.assert_array_equal(g, r)
    for i, det in enumerate(d.serve_stream(batches)):      # and a full stream afterwards
        for g, r in zip(det, want[i]):
            np.testing.assert_array_equal(g, r)
        if i == 1:
            break                                 # left by `break`: CPython finalises the generator at once (refcount), which drains
    d.stage_images(batches[1])
    t = d.run_async()                             # explicit tickets: drain() closes them too
    d.drain()
    with pytest.raises(Exception):
        d.collect(t)
    final = d.serve(batches[3])
    for g, r in zip(final, want[3]):
        np.testing.assert_array_equal(g, r)
    d.close()


# ------------------------------------------------------------------ fp16 range with checkpoint-like statistics
TRAINED_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import FULL_MC, make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
p = make_params(**FULL_MC)
w = dict(make_weights(p, seed=21, cls_spread=20.0))
rng = np.random.default_rng(77)
# what trained checkpoints look like and the initialisers do not: batch-norm scales spread over orders of magnitude
# (gamma / sqrt(var) log-uniform in [0.05, 30]), depthwise taps up to +-8, BiFPN activations small
# (per channel, in six batch norms and three depthwise kernels picked at random - a spread like that in EVERY layer of a random
# network compounds to float32 overflow, which no trained network does; the layer's geometric-mean scale is divided back out)
gam = sorted(k for k in w if k.endswith("/gamma"))
dws = sorted(k for k in w if k.endswith("depthwise_kernel") and "blocks_" in k)
for k in rng.choice(gam, 6, replace=False):
    f = np.exp(rng.uniform(np.log(0.05), np.log(30.0), w[k].shape))
    w[k] = (w[k] * f / np.exp(np.mean(np.log(f)))).astype(np.float32)
for k in rng.choice(dws, 3, replace=False):
    w[k] = (w[k] * rng.uniform(1.0, 8.0 / max(1e-6, float(np.abs(w[k]).max())), w[k].shape)).astype(np.float32)
d = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
d.set_dropout_seed(9)
imgs = make_images(2, 128, 192, seed=5)
det = d.serve(imgs)
cls, box = d.head_outputs(2)
out = {"n": np.int64(d.range_demotions())}
det2 = d.serve(imgs)
out["n2"] = np.int64(d.range_demotions())
for i, x in enumerate(cls + box):
    out["head_%%d" %% i] = x
out["valid"] = det[3]
out["same"] = np.bool_(all(np.array_equal(x, y) for x, y in zip(det, det2)))
out["finite"] = np.bool_(all(np.isfinite(x).all() for x in list(det) + cls + box))
np.savez(sys.argv[1], **out)
d.close()
print("saved")
"""


def test_checkpoint_like_statistics_serve_and_say_how_many_ops_were_repacked(tmp_path):
    """Batch-norm scales log-uniform over [0.05, 30] x the initialiser's, depthwise taps up to +-8: the default scheme either
    holds the float32 bar as it is or re-packs the ops whose operands left fp16's range - never fails, never returns
    infinities - and ends within the float32 bar of a handle that runs three bf16 pieces everywhere."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for scheme in ("f16x2", "bf16x3"):
        e = dict(os.environ, UDA_PW_SCHEME=scheme)
        e.pop("UDA_PW_TERMS", None)
        out = str(tmp_path / ("trained_%s.npz" % scheme))
        r = subprocess.run([sys.executable, "-c", TRAINED_WORKER % {"root": root}, out], cwd=root, env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "saved" in r.stdout, (scheme, r.stdout[-1500:], r.stderr[-2500:])
        res[scheme] = dict(np.load(out))
    got, ref = res["f16x2"], res["bf16x3"]
    print("ops re-packed under checkpoint-like statistics:", int(got["n"]))
    assert bool(ref["finite"]), "the three-piece handle itself overflows float32: the statistics are not checkpoint-like"
    assert bool(got["finite"]), "non-finite results after %d demotions" % int(got["n"])
    assert bool(got["same"]), "the second serve of the batch differs from the first (served again after %d demotions)" % int(got["n"])
    assert int(got["n2"]) == int(got["n"]) and int(ref["n"]) == 0
    assert int(got["n"]) <= 40, "some ops, not the network"
    np.testing.assert_array_equal(got["valid"], ref["valid"])
    for k in sorted(k for k in got if k.startswith("head_")):
        g, r = got[k].astype(np.float64), ref[k].astype(np.float64)
        assert np.sqrt(np.mean((g - r) ** 2)) <= 2e-5 * np.sqrt(np.mean(r * r)) + 1e-7, k


# ------------------------------------------------------------------ MC samples striped over ranks
SAMPLE_SHARD_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch.distributed as dist
from common import FULL_MC, BOX_ONLY_MC, make_images, make_params, make_weights
from uda_amd import dist as udist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
mode, n_img, T = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
p = make_params(**dict(FULL_MC if mode == "full" else BOX_ONLY_MC, mc_dropoutsamp=T))
w = make_weights(p, seed=33, cls_spread=20.0)
imgs = make_images(n_img, 100, 180, seed=34)
drv = udist.SampleShardedDriver(p["name"], n_img, p, w, rank, world)
drv.set_dropout_seed(11)
got = drv.serve(imgs)
np.savez(sys.argv[1] + ".rank%%d.npz" %% rank, *got)
dist.barrier(); dist.destroy_process_group(); drv.close()
'''


@pytest.mark.parametrize("mode,n_img,T,world", [("full", 1, 5, 2), ("full", 3, 4, 2), ("box_only", 2, 3, 3)])
def test_mc_sample_sharded_serve_equals_one_process(tmp_path, mode, n_img, T, world):
    """north_star "images (and optionally MC samples) shard": the T samples of every image striped over the ranks (gloo, all on
    GPU 0), head outputs re-sharded to the image's owner, aggregated in sample order - bit-identical to ONE process that
    serves the batch, for the reference's batch-1 protocol (one image, five samples on two ranks: rank 1 owns no image), for
    more images than ranks, and for a configuration in which only the box head carries the sample axis."""
    import os, socket, subprocess, sys
    from common import BOX_ONLY_MC, ROOT
    p = make_params(**dict(FULL_MC if mode == "full" else BOX_ONLY_MC, mc_dropoutsamp=T))
    w = make_weights(p, seed=33, cls_spread=20.0)
    imgs = make_images(n_img, 100, 180, seed=34)
    d = _driver(p, w, n_img)
    d.set_dropout_seed(11)
    want = d.serve(imgs)
    d.close()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "det")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", SAMPLE_SHARD_WORKER % {"root": ROOT}, out, mode, str(n_img), str(T)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for pr in procs:
        o = pr.communicate(timeout=300)[0]
        assert pr.returncode == 0, o[-3000:]
    for rank in range(world):
        z = np.load(out + ".rank%d.npz" % rank)
        got = [z["arr_%d" % i] for i in range(len(want))]
        for g, r in zip(got, want):
            np.testing.assert_array_equal(g, r)
