"""The boundary promises the reference's `model_params` dict (infer_lib.py:118-140): every key of
hparams_config.default_detection_configs() is consumed by the planner, inert at inference in the reference itself, or
refused with the reference's kind of ValueError - never silently ignored (missing 1 / weak 3).

For every architecture switch that efficientdet_keras.py / efficientnet_model.py read at inference the test flips the
value and demands that the lowered op list (or the weight set it needs) CHANGES, or that the planner raises."""
import numpy as np
import pytest

from common import FULL_MC, make_params, make_weights
from uda_amd import capi, hparams_config as hp, plan as plan_mod, weights as W


def _plan(**over):
    p = make_params(**over)
    return plan_mod.Plan(p, make_weights(p), chunk_images=1, max_images=1), p


def _signature(pl):
    """What the executor would run: op kinds, activations, BN / bias presence, fusion weights, buffer shapes."""
    ops = [(o["kind"], o["act"], o.get("fuse_act", 0), o["bias_off"] >= 0, o["bn_scale_off"] >= 0, tuple(o["resample"]),
            tuple(np.float32(o["fuse_w"]).tolist()), tuple(o["ins"]), o["out"]) for o in pl.ops]
    bufs = [(b.H, b.W, b.C, b.per_sample) for b in pl.bufs]
    return ops, bufs, pl.blob_len


def test_every_default_key_has_a_stated_handling():
    keys = set(hp.default_detection_configs().as_dict())
    table = set(plan_mod.MODEL_PARAM_HANDLING)
    assert keys <= table, "keys without a stated handling: %s" % sorted(keys - table)
    assert table <= keys, "handling stated for keys the defaults do not have: %s" % sorted(table - keys)
    for k, v in plan_mod.MODEL_PARAM_HANDLING.items():
        assert v.split(":")[0].split(" ")[0] in ("consumed", "inert", "checked"), (k, v)


# every key efficientdet_keras.py / backbone/efficientnet_model.py read while BUILDING or CALLING the inference model
# (grep of `config.<key>` in efficientdet_keras.py + what FNode / ResampleFeatureMap / heads receive), with a value
# that differs from the default.  "changes": the plan must differ; an exception class: the planner must raise it.
SWITCHES = [
    ("act_type", "relu", "changes"), ("act_type", "relu6", "changes"), ("act_type", "hswish", "changes"),
    ("act_type", "mish", "changes"), ("act_type", "srelu", ValueError), ("act_type", "gelu", ValueError),
    ("separable_conv", False, ValueError),
    ("conv_bn_act_pattern", True, "changes"),
    ("conv_after_downsample", True, "changes"),
    ("apply_bn_for_resampling", False, "changes"),
    ("fpn_weight_method", "sum", "changes"), ("fpn_weight_method", "attn", "changes"),
    ("fpn_weight_method", "channel_attn", ValueError), ("fpn_weight_method", "channel_fastattn", ValueError),
    ("fpn_weight_method", "nonsense", ValueError),
    ("fpn_name", "qufpn", ValueError), ("fpn_name", "bifpn_dyn", "same"), ("fpn_name", "no_such_fpn", KeyError),
    ("data_format", "channels_first", ValueError),
    ("heads", ["object_detection", "segmentation"], ValueError),
    ("fpn_num_filters", 48, "changes"), ("fpn_cell_repeats", 2, "changes"), ("box_class_repeats", 2, "changes"),
    ("min_level", 4, "changes"), ("max_level", 6, "changes"), ("num_scales", 2, "changes"),
    ("aspect_ratios", [1.0, 2.0], "changes"), ("num_classes", 3, "changes"), ("image_size", "256x128", "changes"),
    ("backbone_name", "efficientnet-b1", "changes"), ("backbone_name", "efficientnet-lite0", KeyError),
    ("survival_prob", 0.8, "same"), ("is_training_bn", True, "same"), ("strategy", "tpu", "same"),
    ("grad_checkpoint", True, "same"), ("mixed_precision", True, "same"),
]


@pytest.mark.parametrize("key,value,expect", SWITCHES, ids=["%s=%s" % (k, v) for k, v, _ in SWITCHES])
def test_switch_is_honoured_or_refused(key, value, expect):
    base, _ = _plan()
    if isinstance(expect, type):
        with pytest.raises(expect):
            _plan(**{key: value})
        return
    got, _ = _plan(**{key: value})
    same = _signature(got) == _signature(base)
    assert same == (expect == "same"), "%s=%r: the lowered network %s" % (key, value, "did not change" if same else "changed")


def test_activation_reaches_every_layer_that_applies_relu_fn():
    """act_type is ONE function for the stem, expand, depthwise and SE reduce layers (efficientdet_keras.py:864-868 ->
    efficientnet_model.py relu_fn), the BiFPN nodes (:229-236) and the head layers (:458-459,638-639)."""
    for name, code in (("relu", capi.ACT_RELU), ("relu6", capi.ACT_RELU6), ("hswish", capi.ACT_HSWISH), ("mish", capi.ACT_MISH), ("silu", capi.ACT_SWISH),
                       ("swish_native", capi.ACT_SWISH)):
        pl, p = _plan(act_type=name, **FULL_MC)
        kinds = {}
        for o in pl.ops:
            kinds.setdefault(o["kind"], []).append(o)
        assert all(o["act"] == code for o in kinds[capi.OP_STEM] + kinds[capi.OP_SE])
        if code != capi.ACT_SWISH:
            assert capi.OP_MBX not in kinds, "the fused MBConv kernels are swish kernels"
            assert all(o["act"] == code for o in kinds[capi.OP_DW])
        seps = kinds[capi.OP_SEP]
        nodes = [o for o in seps if o["fuse_in"]]
        assert nodes and all(o["fuse_act"] == code and o["act"] == capi.ACT_NONE for o in nodes)
        heads = [o for o in seps if not o["fuse_in"]]
        predict = [o for o in heads if pl.bufs[o["out"]].kind in (2, 3)]
        assert predict and all(o["act"] == capi.ACT_NONE for o in predict)
        assert all(o["act"] == code for o in heads if o not in predict)
        # 1x1 convs: the expand convs carry the activation, projections / resample convs none
        acts = {o["act"] for o in kinds[capi.OP_PW]}
        assert acts <= {capi.ACT_NONE, code}


def test_conv_bn_act_pattern_moves_the_activation_behind_the_bn_and_drops_the_bias():
    pl, p = _plan(conv_bn_act_pattern=True)
    nodes = [o for o in pl.ops if o["kind"] == capi.OP_SEP and o["fuse_in"]]
    assert nodes and all(o["fuse_act"] == capi.ACT_NONE and o["act"] == capi.ACT_SWISH and o["bias_off"] < 0 for o in nodes)
    names = {s[0] for s in W.variable_specs(p)}
    assert not any(n.endswith("op_after_combine5/conv/bias") for n in names)
    assert any(n.endswith("op_after_combine5/conv/pointwise_kernel") for n in names)


def test_resample_switches_change_the_weight_set_too():
    p = make_params(apply_bn_for_resampling=False)
    names = {s[0] for s in W.variable_specs(p)}
    assert "resample_p6/conv2d/kernel" in names and "resample_p6/bn" not in names
    pl, _ = _plan(conv_after_downsample=True)
    pool = [i for i, o in enumerate(pl.ops) if o["kind"] == capi.OP_POOL]
    first = pl.ops[pool[0]]
    assert pl.bufs[first["ins"][0]].C == 320 and pl.bufs[first["out"]].C == 320, "P5 is pooled BEFORE the 1x1 conv"
    nxt = pl.ops[pool[0] + 1]
    assert nxt["kind"] == capi.OP_PW and nxt["ins"] == [first["out"]] and pl.bufs[nxt["out"]].C == p["fpn_num_filters"]


def test_attn_fusion_weights_are_a_softmax():
    p = make_params(fpn_weight_method="attn")
    w = make_weights(p)
    pl = plan_mod.Plan(p, w, chunk_images=1, max_images=1)
    node = [o for o in pl.ops if o["kind"] == capi.OP_SEP and o["fuse_in"]][0]
    raw = np.array([w["fpn_cells/cell_0/fnode0/WSM"], w["fpn_cells/cell_0/fnode0/WSM_1"]], np.float64).reshape(-1)
    want = np.exp(raw) / np.exp(raw).sum()
    assert np.allclose(node["fuse_w"][:2], want, rtol=1e-6) and abs(sum(node["fuse_w"][:2]) - 1) < 1e-6


def test_custom_fpn_config_nodes_are_lowered():
    nodes = [dict(feat_level=6, inputs_offsets=[3, 4]), dict(feat_level=5, inputs_offsets=[2, 5]),
             dict(feat_level=4, inputs_offsets=[1, 6]), dict(feat_level=3, inputs_offsets=[0, 7]),
             dict(feat_level=4, inputs_offsets=[1, 7, 8]), dict(feat_level=5, inputs_offsets=[2, 6, 9]),
             dict(feat_level=6, inputs_offsets=[3, 5, 10]), dict(feat_level=7, inputs_offsets=[4, 11])]
    same, _ = _plan(fpn_config=dict(nodes=nodes, weight_method="fastattn"))
    base, _ = _plan()
    assert _signature(same) == _signature(base), "the default BiFPN written out as fpn_config is the default network"
    with pytest.raises(ValueError):
        _plan(fpn_config=dict(nodes=[dict(feat_level=3, inputs_offsets=[0, 1, 2, 3])]))
    with pytest.raises(ValueError):
        _plan(fpn_config=dict(nodes=[dict(feat_level=6, inputs_offsets=[3, 4], weight_method="sum")], weight_method="fastattn"))


def test_unknown_keys_are_reported_not_swallowed():
    pl, _ = _plan(my_own_key=3)
    assert pl.unknown_keys == ["my_own_key"]
