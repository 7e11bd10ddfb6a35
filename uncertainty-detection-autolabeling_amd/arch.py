"""Static description of the detector network on the hot path.

What the reference builds as Keras layers is described here as plain data that
`plan.py` lowers to the C-ABI op list and `weights.py` uses to enumerate the
reference-named variables:

  * EfficientNet block table, width/depth scaling and rounding
      (backbone/efficientnet_builder.py:34-49,166-171;
       backbone/efficientnet_model.py:162-184,731-834)
  * which block outputs are the stride-8/16/32 features ("reduction" endpoints)
      (backbone/efficientnet_model.py:863-909; efficientdet_keras.py:1000-1001)
  * BiFPN node list                                  (fpn_configs.py:27-78)
"""
import math
import re

# (width_coefficient, depth_coefficient)  efficientnet_builder.py:36-48
EFFICIENTNET_PARAMS = {
    "efficientnet-b0": (1.0, 1.0), "efficientnet-b1": (1.0, 1.1),
    "efficientnet-b2": (1.1, 1.2), "efficientnet-b3": (1.2, 1.4),
    "efficientnet-b4": (1.4, 1.8), "efficientnet-b5": (1.6, 2.2),
    "efficientnet-b6": (1.8, 2.6), "efficientnet-b7": (2.0, 3.1),
}

# r=repeats k=kernel s=stride e=expand i=in o=out se=squeeze ratio (builder.py:166-171)
DEFAULT_BLOCKS = [
    "r1_k3_s11_e1_i32_o16_se0.25", "r2_k3_s22_e6_i16_o24_se0.25",
    "r2_k5_s22_e6_i24_o40_se0.25", "r3_k3_s22_e6_i40_o80_se0.25",
    "r3_k5_s11_e6_i80_o112_se0.25", "r4_k5_s22_e6_i112_o192_se0.25",
    "r1_k3_s11_e6_i192_o320_se0.25",
]
DEPTH_DIVISOR = 8
BN_EPS = 1e-3


def round_filters(filters, width):
    """Scale by `width`, round half-up to a multiple of 8, never lose >10 %."""
    if not width:
        return filters
    f = filters * width
    new = max(DEPTH_DIVISOR, int(f + DEPTH_DIVISOR / 2) // DEPTH_DIVISOR * DEPTH_DIVISOR)
    if new < 0.9 * f:
        new += DEPTH_DIVISOR
    return int(new)


def round_repeats(repeats, depth):
    return int(math.ceil(depth * repeats)) if depth else repeats


def _decode(block_string):
    opts = {}
    for op in block_string.split("_"):
        m = re.match(r"([a-z]+)(\d.*)", op)
        if m:
            opts[m.group(1)] = m.group(2)
    return dict(kernel=int(opts["k"]), repeat=int(opts["r"]), cin=int(opts["i"]),
                cout=int(opts["o"]), expand=int(opts["e"]), stride=int(opts["s"][0]),
                se_ratio=float(opts["se"]) if "se" in opts else None)


def _stage_args(entry):
    """One stage of a custom block table (`config.backbone_config.blocks`, efficientdet_keras.py:873-878): a block string
    ("r1_k3_s11_e1_i32_o16_se0.25") or the fields of the reference's `BlockArgs` (efficientnet_model.py:56-70)."""
    if isinstance(entry, str):
        return _decode(entry)
    e = entry if isinstance(entry, dict) else entry._asdict()
    for key in ("conv_type", "fused_conv", "super_pixel"):
        if e.get(key):
            raise ValueError("block option %s=%r is not on the hot path (EfficientDet backbones use plain MBConv)" % (key, e[key]))
    strides = e.get("strides", [1, 1])
    return dict(kernel=int(e["kernel_size"]), repeat=int(e["num_repeat"]), cin=int(e["input_filters"]),
                cout=int(e["output_filters"]), expand=int(e["expand_ratio"]),
                stride=int(strides[0] if isinstance(strides, (list, tuple)) else strides),
                se_ratio=float(e["se_ratio"]) if e.get("se_ratio") else None, id_skip=e.get("id_skip", True))


def backbone_blocks(backbone_name, backbone_config=None):
    """Expanded list of MBConv blocks: dicts with kernel, stride, expand, cin, cout, se.

    `se` is the squeeze width max(1, int(block input filters * ratio)); the
    first block of a stage carries the stage stride and input width, the
    repeats have stride 1 and cin == cout (efficientnet_model.py:741-834,393-397).
    `backbone_config` = {"blocks": [...]} replaces the default stage table (efficientdet_keras.py:873-878).
    """
    width, depth = EFFICIENTNET_PARAMS[backbone_name]
    blocks = []
    table = DEFAULT_BLOCKS
    if backbone_config:
        cfg = backbone_config if isinstance(backbone_config, dict) else backbone_config.as_dict()
        table = cfg.get("blocks") or DEFAULT_BLOCKS
    for s in table:
        a = _stage_args(s)
        cin, cout = round_filters(a["cin"], width), round_filters(a["cout"], width)
        for r in range(round_repeats(a["repeat"], depth)):
            b_in = cin if r == 0 else cout
            stride = a["stride"] if r == 0 else 1
            se = max(1, int(b_in * a["se_ratio"])) if a["se_ratio"] else 0
            blocks.append(dict(kernel=a["kernel"], stride=stride, expand=a["expand"],
                               cin=b_in, cout=cout, se=se,
                               skip=(a.get("id_skip", True) and stride == 1 and b_in == cout)))
    return blocks


def stem_filters(backbone_name):
    return round_filters(32, EFFICIENTNET_PARAMS[backbone_name][0])


def reduction_block_ids(blocks):
    """Indices of blocks whose output is reduction_1..5 (model.py:863-885)."""
    ids = []
    for i in range(len(blocks)):
        if i == len(blocks) - 1 or blocks[i + 1]["stride"] > 1:
            ids.append(i)
    return ids


def bifpn_nodes(min_level, max_level):
    """[{feat_level, inputs_offsets}] – top-down then bottom-up (fpn_configs.py:27-78)."""
    num_levels = max_level - min_level + 1
    node_ids = {min_level + i: [i] for i in range(num_levels)}
    nxt = num_levels
    nodes = []
    for lvl in range(max_level - 1, min_level - 1, -1):
        nodes.append(dict(feat_level=lvl,
                          inputs_offsets=[node_ids[lvl][-1], node_ids[lvl + 1][-1]]))
        node_ids[lvl].append(nxt)
        nxt += 1
    for lvl in range(min_level + 1, max_level + 1):
        nodes.append(dict(feat_level=lvl,
                          inputs_offsets=node_ids[lvl] + [node_ids[lvl - 1][-1]]))
        node_ids[lvl].append(nxt)
        nxt += 1
    return nodes


def dropout_rates(config):
    """(backbone rate, class-head rate, box-head rate)  efficientdet_keras.py:905-916,
    efficientnet_model.py:300-305."""
    if not config["mc_dropout"]:
        return 0.0, 0.0, 0.0
    base = float(config["mc_dropoutrate"] or 0.0)
    return (base, float(config["mc_classheadrate"] or base),
            float(config["mc_boxheadrate"] or base))


def mc_flags(config):
    """(class outputs are stacked over T, box outputs are stacked over T, T).

    The reference enters its MC loop whenever `mc_dropout` is set and stacks a
    head's outputs iff that head's rate or the global rate is non-zero
    (efficientdet_keras.py:981-1050)."""
    if not config["mc_dropout"]:
        return False, False, 1
    g = bool(config["mc_dropoutrate"])
    return (g or bool(config["mc_classheadrate"]), g or bool(config["mc_boxheadrate"]),
            int(config["mc_dropoutsamp"]))
