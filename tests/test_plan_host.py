"""Host logic on the CPU: config mirror, topology / structure KATs, the lowering to the op
list (sample-axis propagation, arena liveness), anchors, cost accounting."""
import os

import numpy as np
import pytest


from common import FULL_MC, HEAD_MC, LOSS_ATT, PLAIN, BOX_ONLY_MC, make_params, make_weights
from uda_amd import arch, capi, hparams_config as hp, plan as plan_mod, weights as W


# ------------------------------------------------------------------ config mirror (hparams_config.py)
def test_config_override_forms(tmp_path):
    c = hp.get_efficientdet_config("efficientdet-d0")
    assert (c.fpn_num_filters, c.fpn_cell_repeats, c.box_class_repeats, c.backbone_name) == (64, 3, 3, "efficientnet-b0")
    c.override("mc_dropout=True,mc_dropoutrate=0.05,nms_configs.max_output_size=50,aspect_ratios=1.0*2.0*0.5")
    assert c.mc_dropout is True and c.mc_dropoutrate == 0.05 and c.nms_configs.max_output_size == 50
    assert c.aspect_ratios == [1.0, 2.0, 0.5] and c["nms_configs"]["method"] == "gaussian"
    with pytest.raises(KeyError):
        c.override(dict(no_such_key=1))
    y = tmp_path / "cfg.yaml"
    y.write_text("num_classes: 7\nimage_size: '1024x512'\nmc_dropout: True\nmc_boxheadrate: 0.05\nlabel_map: 'kitti'\n")
    c.override(str(y))
    assert c.num_classes == 7 and hp.parse_image_size(c.image_size) == (512, 1024) and c.label_map == "kitti"
    with pytest.raises(ValueError):
        c.override("nonsense")
    d = hp.get_efficientdet_config("efficientdet-d2")
    assert (d.backbone_name, d.fpn_num_filters, d.fpn_cell_repeats) == ("efficientnet-b2", 112, 5)
    with pytest.raises(ValueError):
        hp.get_detection_config("resnet")


def test_reference_yaml_files_load_if_present():
    ref = "/root/reference/configs/train"
    if not os.path.isdir(ref):
        pytest.skip("reference not mounted (GPU box)")
    for f in sorted(os.listdir(ref)):
        if f.endswith(".yaml"):
            c = hp.get_efficientdet_config("efficientdet-d0")
            c.override(os.path.join(ref, f))
            assert c.num_classes in (7, 10)


def test_image_and_feature_sizes():
    assert hp.parse_image_size("1280x768") == (768, 1280) and hp.parse_image_size(512) == (512, 512)
    assert hp.get_feat_sizes("1280x768", 7)[3:] == [(96, 160), (48, 80), (24, 40), (12, 20), (6, 10)]
    assert hp.get_feat_sizes(1024, 7)[3:] == [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]


# ------------------------------------------------------------------ structure KATs (efficientnet_builder_test.py:46-86)
@pytest.mark.parametrize("bb,full,feat", [("efficientnet-b0", 5288548, 3595388), ("efficientnet-b1", 7794184, None),
                                           ("efficientnet-b2", 9109994, None), ("efficientnet-b3", 12233232, None),
                                           ("efficientnet-b4", 19341616, None), ("efficientnet-b5", 30389784, None),
                                           ("efficientnet-b6", 43040704, None), ("efficientnet-b7", 66347960, None)])
def test_backbone_parameter_counts(bb, full, feat):
    p = dict(make_params(), backbone_name=bb)
    specs = [s for s in W.variable_specs(p) if s[0].startswith(bb)]
    n = W.count_trainable(specs)
    if feat is not None:
        assert n == feat
    last = arch.backbone_blocks(bb)[-1]["cout"]
    head = arch.round_filters(1280, arch.EFFICIENTNET_PARAMS[bb][0])
    assert n + last * head + 2 * head + head * 1000 + 1000 == full


def test_block_table_and_reductions():
    b0 = arch.backbone_blocks("efficientnet-b0")
    assert len(b0) == 16 and [b0[i]["cout"] for i in arch.reduction_block_ids(b0)] == [16, 24, 40, 112, 320]
    assert [b["se"] for b in b0[:4]] == [8, 4, 6, 6]
    b2 = arch.backbone_blocks("efficientnet-b2")
    assert sorted(set(b["cout"] for b in b2)) == [16, 24, 48, 88, 120, 208, 352]
    assert [sum(1 for b in b2 if b["cout"] == c) for c in (16, 24, 48, 88, 120, 208, 352)] == [2, 3, 3, 4, 4, 5, 2]
    nodes = arch.bifpn_nodes(3, 7)
    assert [n["inputs_offsets"] for n in nodes] == [[3, 4], [2, 5], [1, 6], [0, 7], [1, 7, 8], [2, 6, 9], [3, 5, 10], [4, 11]]


def test_weight_names_follow_the_reference():
    p = make_params(**FULL_MC)
    w = make_weights(p)
    for k in ("efficientnet-b0/stem/conv2d/kernel", "efficientnet-b0/blocks_0/conv2d/kernel",
              "efficientnet-b0/blocks_1/conv2d_1/kernel", "efficientnet-b0/blocks_1/tpu_batch_normalization_2/gamma",
              "efficientnet-b0/blocks_3/se/conv2d_1/bias", "resample_p6/conv2d/kernel", "resample_p6/bn/moving_mean",
              "fpn_cells/cell_0/fnode0/WSM_1", "fpn_cells/cell_0/fnode1/resample_0_2_6/conv2d/kernel",
              "fpn_cells/cell_2/fnode7/op_after_combine12/conv/pointwise_kernel",
              "class_net/class-2-bn-7/beta", "class_net/class-predict/bias", "box_net/box-predict/pointwise_kernel"):
        assert k in w, k
    assert "resample_p7/conv2d/kernel" not in w and "fpn_cells/cell_1/fnode1/resample_0_2_6/conv2d/kernel" not in w
    assert w["box_net/box-predict/pointwise_kernel"].shape == (1, 1, 64, 72)
    np.testing.assert_allclose(w["class_net/class-predict/bias"], -np.log(99.0), rtol=1e-6)


# ------------------------------------------------------------------ lowering
def _plan(over, **kw):
    p = make_params(**over)
    return plan_mod.Plan(p, make_weights(p), **kw), p


def test_op_list_shape_d0():
    pl, _ = _plan(FULL_MC, chunk_images=2, max_images=4)
    kinds = [o["kind"] for o in pl.ops]
    fused = kinds.count(capi.OP_MBX)           # expand+depthwise as one op: blocks 1-5 (Cin <= 48) and the deep
    shallow = 5 if plan_mod.mbx_supported(16, 96, 3, 2) else 0      # stride-1 blocks 6-10, 12-15; block 11 (5x5 stride 2): round 3
    deep = 9 if plan_mod.mbx_supported(112, 672, 5, 1) else 0
    deep += 1 if plan_mod.mbx_supported(112, 672, 5, 2) else 0
    assert fused == shallow + deep and fused == 15
    sep = kinds.count(capi.OP_SEP)             # 24 BiFPN nodes + 2 heads x 5 levels x (3 + 1) separable convs
    assert sep == (64 if pl.fuse_sep else 0)
    proj = sum(1 for o in pl.ops if o["kind"] == capi.OP_MBX and o["se_scale"] >= 0)   # block 0's projection inside block 1's op
    assert proj == (1 if pl.fuse_proj and shallow else 0)
    fin = sum(1 for o in pl.ops if o.get("fuse_in"))      # BiFPN fusions computed inside the node's separable conv (round 4)
    assert fin == (24 if pl.fuse_sep and plan_mod.sepf_supported(64, 64) else 0)
    assert all(o["kind"] == capi.OP_SEP and len(o["ins"]) in (2, 3) for o in pl.ops if o.get("fuse_in"))
    assert len(pl.ops) == 224 - fused - sep - proj - fin
    assert kinds.count(capi.OP_STEM) == 1 and kinds.count(capi.OP_SE) == 16 and kinds.count(capi.OP_FUSE) == 24 - fin
    assert kinds.count(capi.OP_POOL) == 2 and kinds.count(capi.OP_DW) == 16 + 24 + 40 - fused - sep
    assert kinds.count(capi.OP_PW) == 31 + 1 + 5 + 24 + 40 - fused - sep - proj
    assert len(pl.sites) == 61 and pl.T == 3


def test_sample_axis_propagation():
    full, _ = _plan(FULL_MC)
    bn = full.buffer_names
    assert not full.bufs[bn["image"]].per_sample and not full.bufs[bn["stem"]].per_sample
    # block 0: the dropout after the shared depthwise is deferred into the SE gate (it commutes with the
    # squeeze and with the 1x1 projection), so the sample axis starts at the gate / the projection output
    assert not full.bufs[bn["blocks_0/dw"]].per_sample and full.bufs[bn["blocks_0/se"]].per_sample
    assert ("blocks_0/out" not in bn) == bool(full.fuse_proj)      # absorbed by block 1's fused op: never materialised
    assert full.bufs[bn["blocks_1/dw"]].per_sample and full.bufs[bn["blocks_15/out"]].per_sample
    se0 = [o for o in full.ops if full.bufs[o["out"]].name == "blocks_0/se"][0]
    assert se0["drop_site"] == full.site_index["blocks_0/dw"]
    assert all(o["drop_site"] == -1 for o in full.ops if full.bufs[o["out"]].name == "blocks_0/dw")
    assert full.cls_stacked_dev and full.box_stacked_dev
    head, _ = _plan(HEAD_MC)
    hb = head.buffer_names
    assert not head.bufs[hb["blocks_15/out"]].per_sample and not head.bufs[hb["cell2/fnode7/out"]].per_sample
    assert "class-0-3/dw" not in hb                                # the head's separable convs are single fused ops
    assert any(o["kind"] == capi.OP_SEP and head.bufs[o["out"]].name == "class-0-3" for o in head.ops)
    # the first head layer reads a per-image tensor: its dropout site is deferred into the layer behind it (round 5) - its own
    # output stays per image, the second layer applies the site to its input channels and opens the sample axis
    deferred = head.fuse_sep and plan_mod.sep_tin_supported(64, 64)
    assert head.bufs[hb["class-0-3"]].per_sample == (not deferred) and head.bufs[hb["box-predict-7"]].per_sample
    l0 = [o for o in head.ops if head.bufs[o["out"]].name == "class-0-3"][0]
    l1 = [o for o in head.ops if head.bufs[o["out"]].name == "class-1-3"][0]
    assert head.bufs[hb["class-1-3"]].per_sample
    if deferred:
        assert l0["drop_site"] == -1 and l1["drop_site2"] == head.site_index["class-0-3"] and l1["drop_site"] == head.site_index["class-1-3"]
    else:
        assert l0["drop_site"] == head.site_index["class-0-3"] and l1["drop_site2"] == -1
    assert all(o["drop_site2"] == -1 for o in full.ops if o["kind"] == capi.OP_SEP)       # full MC: the inputs are per sample already
    assert all(o["drop_site"] == -1 for o in head.ops if head.bufs[o["out"]].name.startswith("blocks_"))
    box_only, _ = _plan(BOX_ONLY_MC)
    assert not box_only.cls_stacked_dev and box_only.box_stacked_dev and not box_only.cls_stacked
    plain, _ = _plan(PLAIN)
    assert plain.T == 1 and not any(b.per_sample for b in plain.bufs) and not plain.sites
    la, _ = _plan(LOSS_ATT)
    assert la.bufs[la.head_out["box"][0]].C == 72 and plain.bufs[plain.head_out["box"][0]].C == 36


@pytest.mark.parametrize("over,chunk", [(FULL_MC, 1), (FULL_MC, 3), (HEAD_MC, 2), (PLAIN, 4)])
def test_arena_liveness_has_no_overlap(over, chunk):
    pl, _ = _plan(over, chunk_images=chunk, max_images=4)
    live = []
    for bi, b in enumerate(pl.bufs):
        if b.kind != 0 or b.first is None:
            continue
        size = pl._rows(b) * b.H * b.W * b.C
        assert b.offset % plan_mod.ALIGN == 0 and b.offset + size <= pl.arena_floats
        live.append((b.first, b.last, b.offset, b.offset + size, bi))
    for i, (f1, l1, o1, e1, b1) in enumerate(live):
        for f2, l2, o2, e2, b2 in live[i + 1:]:
            if f1 <= l2 and f2 <= l1:                      # lifetimes intersect -> memory must not
                assert e1 <= o2 or e2 <= o1, (pl.bufs[b1].name, pl.bufs[b2].name)
    # every op's inputs are written before they are read
    written = {i for i, b in enumerate(pl.bufs) if b.kind == 1}
    for o in pl.ops:
        for i in o["ins"] + [o[k] for k in ("se_scale", "residual") if o[k] >= 0]:
            assert i in written, pl.bufs[i].name
        written.add(o["out"])
        if o["se_partial"] >= 0:
            written.add(o["se_partial"])


def test_to_c_structures_roundtrip():
    pl, p = _plan(FULL_MC, chunk_images=2, max_images=5)
    m, bufs, ops, sites, blob, anchors = pl.to_c()
    assert m.mc_samples == 3 and m.cls_stacked == 1 and m.has_uncert == 1 and m.decode_method == capi.DECODE_LNORM
    assert abs(m.nms_soft_sigma - 0.25) < 1e-7 and abs(m.nms_score_thresh - 0.001) < 1e-9 and m.max_output_size == 100
    assert m.num_levels == 5 and [m.level_h[i] for i in range(5)] == [16, 8, 4, 2, 1]
    assert blob.dtype == np.float32 and blob.size == pl.blob_len and len(ops) == len(pl.ops)
    assert anchors.shape == (sum(h * w for h, w in pl.level_hw) * 9, 4)
    for o, c in zip(pl.ops, ops):
        assert c.kind == o["kind"] and c.out == o["out"] and c.n_in == len(o["ins"]) and c.w_off == o["w_off"]
    ms = plan_mod.Plan(dict(p, uncert_adjust_method="sample", decode_nsamples=30), pl.w).to_c()[0]
    assert ms.decode_method == capi.DECODE_SAMPLE and ms.decode_nsamples == 30
    with pytest.raises(ValueError):
        plan_mod.Plan(dict(p, uncert_adjust_method="bootstrap"), pl.w).to_c()
    hard = dict(p, nms_configs=dict(p["nms_configs"], method="hard"))
    assert plan_mod.nms_params(hard) == (0.0, 0.5, float("-inf"))
    with pytest.raises(ValueError):
        plan_mod.nms_params(dict(p, nms_configs=dict(p["nms_configs"], method="linear")))


def test_anchor_table_matches_oracle():
    from oracle import post_ref
    for size in ("192x128", 64, "1280x768"):
        p = make_params(image_size=size)
        pl = plan_mod.Plan(p, make_weights(p)) if size != "1280x768" else None
        got = pl.anchors() if pl else plan_mod.Plan.anchors(type("X", (), {"cfg": p})())
        np.testing.assert_array_equal(got, post_ref.anchor_boxes(p))


def test_sites_match_oracle_convention():
    from oracle import effdet_ref
    for over in (FULL_MC, HEAD_MC, BOX_ONLY_MC):
        pl, p = _plan(over)
        want = effdet_ref.dropout_sites(p)
        assert [(n, c) for n, c, _ in pl.sites] == [(n, c) for n, c, _ in want]
        np.testing.assert_allclose([r for _, _, r in pl.sites], [r for _, _, r in want])


def test_cost_accounting_close_to_survey_figures():
    """D0 768x1280 C=7 loss-att = 8.487 GMAC/W; the plan shares the stem across T."""
    p = make_params(image_size="1280x768", mc_dropout=True, mc_dropoutrate=0.05, mc_dropoutsamp=10, loss_attenuation=True)
    pl = plan_mod.Plan(p, make_weights(p), chunk_images=2, max_images=32)
    costs = plan_mod.op_costs(pl, 32)
    gmac_per_w = sum(v["flops"] for v in costs.values()) / 2 / 320 / 1e9
    assert 7.9 < gmac_per_w < 8.5        # block 0's depthwise also runs once per image (deferred dropout)
    fused = costs.get(capi.OP_MBX, dict(launches=0))["launches"] // 16
    sep = costs.get(capi.OP_SEP, dict(launches=0))["launches"] // 16
    proj = 1 if pl.fuse_proj else 0
    assert costs[capi.OP_PW]["launches"] == (101 - fused - sep - proj) * 16
    assert costs[capi.OP_DW]["launches"] == (80 - fused - sep) * 16


def test_prediction_data_records_round_trip(tmp_path):
    """the prediction_data.txt line format parses back the way the reference's readers parse it."""
    import ast
    from uda_amd import writers
    rng = np.random.default_rng(0)
    M, C = 6, 3
    un = dict(boxes=rng.uniform(0, 100, (2, M, 4)).astype(np.float32), scores=np.linspace(0.9, 0.1, 2 * M).reshape(2, M).astype(np.float32),
              classes=rng.integers(1, C + 1, (2, M)).astype(np.float32), valid_len=np.array([M, M], np.int32),
              logits=rng.normal(0, 1, (2, M, C)).astype(np.float32), probab=rng.uniform(0, 1, (2, M, C)).astype(np.float32),
              entropy=rng.uniform(0, 1, (2, M)).astype(np.float32), albox=rng.uniform(0, 5, (2, M, 4)).astype(np.float32),
              mcbox=None, mcclass=rng.uniform(0, 1, (2, M, C)).astype(np.float32))
    un["albox"][0, 0, 1] = np.nan
    cal = {"iso_all_albox": un["albox"] * 2}
    recs = writers.prediction_records(un, ["000001", "000002"], 0.45, calibrated=cal)
    assert len(recs) == int((un["scores"] > 0.45).sum())
    path = tmp_path / "prediction_data.txt"
    writers.write_prediction_data(str(path), recs)
    back = [ast.literal_eval(l.replace("inf", "2e308")) for l in open(path)]
    assert back == recs
    r0 = back[0]
    assert list(r0)[:6] == ["image_name", "score_thresh", "top_5scores", "det_score", "bbox", "class"]
    assert r0["image_name"] == "000001.jpg" and r0["uncalib_albox"][1] == 0.0 and "uncalib_mcbox" not in r0
    assert r0["logits"] == [float(str(v)) for v in np.around(un["logits"][0, 0], 4)] and len(r0["iso_all_albox"]) == 4
    assert isinstance(r0["entropy"], float) and len(r0["probab"]) == C and len(r0["uncalib_mcclass"]) == C


def test_reference_block_kats_structure():
    """The reference's own block-level known answers (src/backbone/efficientnet_model_test.py:25-248) on the host side of the
    plan: its BlockArgs (kernel 3, 3 -> 6 filters, expand 6, stride 2, optional SE 0.8, 1 or 3 repeats) expand into the
    table the builder uses, a single block has exactly ONE reduction endpoint (:188-248), and the first conv of an expanding
    block is `blocks_0/conv2d/kernel` (:160-188).  (The (10, 10) logits shapes of :25-158 belong to the classifier head,
    which `features_only` backbones do not build.)"""
    from uda_amd import arch, weights as W
    ref_args = dict(kernel_size=3, num_repeat=3, input_filters=3, output_filters=6, expand_ratio=6, id_skip=True, strides=[2, 2],
                    conv_type=0, fused_conv=0, super_pixel=0)
    blocks = arch.backbone_blocks("efficientnet-b0", {"blocks": [ref_args]})
    assert [(b["cin"], b["cout"], b["stride"], b["expand"], b["kernel"]) for b in blocks] == [(8, 8, 2, 6, 3), (8, 8, 1, 6, 3), (8, 8, 1, 6, 3)]
    assert [b["skip"] for b in blocks] == [False, True, True] and all(b["se"] == 0 for b in blocks)
    assert arch.reduction_block_ids(blocks) == [2]
    se_args = dict(ref_args, id_skip=False, se_ratio=0.8)
    blocks = arch.backbone_blocks("efficientnet-b0", {"blocks": [se_args]})
    assert [b["se"] for b in blocks] == [6, 6, 6] and not any(b["skip"] for b in blocks)       # max(1, int(8 * 0.8))
    single = arch.backbone_blocks("efficientnet-b0", {"blocks": [dict(se_args, num_repeat=1)]})
    assert len(single) == 1 and arch.reduction_block_ids(single) == [0]     # 'reduction_1' and no 'reduction_2'
    # the same table as a block string, and the variable the reference's test looks up
    assert (arch.backbone_blocks("efficientnet-b0", {"blocks": ["r3_k3_s22_e6_i3_o6_se0.8"]})
            == arch.backbone_blocks("efficientnet-b0", {"blocks": [dict(se_args, id_skip=True)]}))
    cfg = make_params(image_size="128x128")
    cfg["backbone_config"] = {"blocks": ["r1_k3_s11_e6_i32_o16_se0.25", "r1_k3_s22_e6_i16_o24_se0.25", "r1_k5_s22_e6_i24_o40_se0.25",
                                         "r1_k3_s22_e6_i40_o80_se0.25", "r1_k5_s22_e6_i80_o112_se0.25"]}
    names = {n: s for n, s, _ in W.variable_specs(cfg)}
    assert names["efficientnet-b0/blocks_0/conv2d/kernel"] == (1, 1, 32, 192) and "efficientnet-b0/blocks_5/conv2d/kernel" not in names
    with pytest.raises(ValueError):
        arch.backbone_blocks("efficientnet-b0", {"blocks": [dict(ref_args, fused_conv=1)]})
    # a plan over a custom table lowers (a stride-1 stage and four stride-2 stages: the three FPN inputs are blocks 2..4)
    pl = plan_mod.Plan(cfg, W.init_weights(cfg, 0))
    assert pl.level_hw[0] == (16, 16) and len(pl.level_hw) == 5


def test_pw_scheme_switches(monkeypatch):
    """UDA_PW_SCHEME names the split scheme of the 1x1 contractions (default: two fp16 pieces); the older UDA_PW_TERMS is
    honoured when the newer switch is silent; anything else is refused (mirror of parse_pw_scheme in csrc/uda_api.hip)."""
    from uda_amd import plan
    monkeypatch.delenv("UDA_PW_SCHEME", raising=False)
    monkeypatch.delenv("UDA_PW_TERMS", raising=False)
    assert plan.pw_scheme() == "f16x2" == plan.PW_SCHEME_DEFAULT
    for terms, want in (("6", "bf16x3"), ("3", "bf16x2"), ("0", "f32")):
        monkeypatch.setenv("UDA_PW_TERMS", terms)
        assert plan.pw_scheme() == want
    monkeypatch.setenv("UDA_PW_SCHEME", "bf16x3")          # the newer switch wins
    assert plan.pw_scheme() == "bf16x3"
    monkeypatch.setenv("UDA_PW_SCHEME", "fp8")
    with pytest.raises(ValueError):
        plan.pw_scheme()
    monkeypatch.delenv("UDA_PW_SCHEME")
    monkeypatch.setenv("UDA_PW_TERMS", "4")
    with pytest.raises(ValueError):
        plan.pw_scheme()
    # the exact-f32 scheme switches the fusions off in the planner, the split schemes keep them
    monkeypatch.setenv("UDA_PW_TERMS", "0")
    assert not plan.mbx_supported(192, 1152, 5, 1)
    monkeypatch.delenv("UDA_PW_TERMS")
    assert plan.mbx_supported(192, 1152, 5, 1)


def test_bifpn_fusion_folds_into_the_separable_conv(monkeypatch):
    """UDA_FUSE_IN (default 1): a BiFPN node is ONE op - a separable conv whose inputs are the node's fusion inputs with their
    resample modes and normalised weights (efficientdet_keras.py:90-136 + 207-227); no `fused` buffer is planned, and the
    arena shrinks by what it held.  UDA_FUSE_IN=0 and the f32 scheme (no fused separable convs at all) keep the FUSE ops."""
    pl, _ = _plan(FULL_MC, chunk_images=2, max_images=4)
    monkeypatch.setenv("UDA_FUSE_IN", "0")
    old, _ = _plan(FULL_MC, chunk_images=2, max_images=4)
    assert not any("/fused" in n for n in pl.buffer_names) and sum("/fused" in n for n in old.buffer_names) == 24
    fin = [o for o in pl.ops if o.get("fuse_in")]
    fuse = [o for o in old.ops if o["kind"] == capi.OP_FUSE]
    assert len(fin) == len(fuse) == 24
    for a, b in zip(fin, fuse):             # same inputs, modes and weights, in node order
        assert [old.bufs[i].name for i in b["ins"]] == [pl.bufs[i].name for i in a["ins"]]
        assert a["resample"] == b["resample"] and a["fuse_w"] == b["fuse_w"] and a["w2_off"] >= 0 and a["w_off"] >= 0
    modes = {tuple(o["resample"][:len(o["ins"])]) for o in fin}
    assert modes == {(capi.RS_NONE, capi.RS_NEAREST_UP), (capi.RS_NONE, capi.RS_NONE, capi.RS_MAXPOOL), (capi.RS_NONE, capi.RS_MAXPOOL)}
    assert pl.arena_floats <= old.arena_floats
    assert plan_mod.op_costs(pl, 4)[capi.OP_SEP]["bytes"] < plan_mod.op_costs(old, 4)[capi.OP_SEP]["bytes"] + plan_mod.op_costs(old, 4)[capi.OP_FUSE]["bytes"]
    monkeypatch.delenv("UDA_FUSE_IN")
    monkeypatch.setenv("UDA_PW_SCHEME", "f32")
    f32, _ = _plan(FULL_MC, chunk_images=2, max_images=4)
    assert not any(o.get("fuse_in") for o in f32.ops) and sum(o["kind"] == capi.OP_FUSE for o in f32.ops) == 24
