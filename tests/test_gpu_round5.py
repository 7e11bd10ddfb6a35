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
    p = make_params(**over)
    w = make_weights(p, seed=11)
    d = _driver(p, w, 2, only_network=True)
    x, _ = PP.preprocess(make_images(2, 128, 192, seed=13), d.image_size, p["mean_rgb"], p["stddev_rgb"])
    d.set_dropout_seed(91)
    cls, box = d.predict(x)
    rcls, rbox = _oracle_net(p, w, x, 91)
    check_heads(cls, rcls)
    check_heads(box, rbox)
    # the switch is not a no-op: the default network on the same weights (where the weight set allows it) differs
    if name in ("relu", "hswish", "conv_after_downsample"):
        q = dict(p, act_type="swish", conv_after_downsample=False)
        d2 = _driver(q, w, 2, only_network=True)
        d2.set_dropout_seed(91)
        cls2, _ = d2.predict(x)
        d2.close()
        assert max(np.abs(a - b).max() for a, b in zip(cls, cls2)) > 1e-3
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
    for over in (dict(act_type="mish"), dict(separable_conv=False), dict(fpn_weight_method="channel_attn"),
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
