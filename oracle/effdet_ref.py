"""ORACLE (test infrastructure): EfficientDet forward on the CPU with torch ops.

Restates, op for op in float32, what the reference's Keras model computes at
inference time.  TF op semantics follow SURVEY §9 (parity unpinned: TF is not
installable here).

  stem / MBConv / SE            backbone/efficientnet_model.py:187-232,420-490,588-612
  block table, rounding         backbone/efficientnet_builder.py:34-49,166-171; model.py:162-184
  reduction endpoints           backbone/efficientnet_model.py:863-909
  P6/P7 resample                efficientdet_keras.py:321-350,886-899,1004-1005
  BiFPN node / fusion / cells   efficientdet_keras.py:86-127,174-182,229-236,788-801; fpn_configs.py:27-78
  class / box heads             efficientdet_keras.py:449-483,629-664
  MC loop and stacking          efficientdet_keras.py:979-1050; utils_extra.py:201-217
  act_type family               utils.py:42-59 (the same function in backbone, BiFPN and heads: efficientdet_keras.py:864-868)
  architecture switches         apply_bn_for_resampling / conv_after_downsample :313-338, conv_bn_act_pattern :218,229-236,
                                fpn weight methods attn | fastattn | sum :96-124, fpn_config.nodes :773-781

Dropout is injected: `masks[site]` is a float32 array [N, T, C] holding the
keep-scale (0 or 1/(1-p)) of SpatialDropout2D (noise shape [N,1,1,C]) for
every image n and MC sample t.  Site names: "blocks_{i}/expand", "blocks_{i}/dw",
"class-{rep}-{level}", "box-{rep}-{level}".
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-3
_WIDTH_DEPTH = {"efficientnet-b0": (1.0, 1.0), "efficientnet-b1": (1.0, 1.1),
                "efficientnet-b2": (1.1, 1.2), "efficientnet-b3": (1.2, 1.4),
                "efficientnet-b4": (1.4, 1.8), "efficientnet-b5": (1.6, 2.2),
                "efficientnet-b6": (1.8, 2.6), "efficientnet-b7": (2.0, 3.1)}
# (repeats, kernel, stride, expand, in, out) with se_ratio 0.25 everywhere
_STAGES = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
           (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]


def _round_ch(c, width):
    c = c * width
    r = max(8, int(c + 4) // 8 * 8)
    return int(r + 8 if r < 0.9 * c else r)


def block_table(backbone):
    width, depth = _WIDTH_DEPTH[backbone]
    out = []
    for rep, k, s, e, ci, co in _STAGES:
        ci, co = _round_ch(ci, width), _round_ch(co, width)
        for r in range(int(math.ceil(depth * rep))):
            cin = ci if r == 0 else co
            out.append((k, s if r == 0 else 1, e, cin, co, max(1, int(cin * 0.25))))
    return out


# ---------------------------------------------------------------- TF op semantics
def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _same_pad(x, k, s, value=0.0):
    """TF 'SAME': out=ceil(in/s), total=max((out-1)s+k-in,0), before=total//2."""
    H, W = x.shape[-2:]
    ph = max((-(-H // s) - 1) * s + k - H, 0)
    pw = max((-(-W // s) - 1) * s + k - W, 0)
    if ph or pw:
        x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=value)
    return x


def conv2d(x, kernel, stride=1, bias=None):
    """x NCHW; kernel TF layout [kh, kw, cin, cout]; padding SAME."""
    k = kernel.shape[0]
    wt = _t(kernel).permute(3, 2, 0, 1).contiguous()
    return F.conv2d(_same_pad(x, k, stride), wt, None if bias is None else _t(bias), stride)


def depthwise(x, kernel, stride=1):
    """kernel TF layout [kh, kw, c, 1]."""
    k, c = kernel.shape[0], kernel.shape[2]
    wt = _t(kernel).permute(2, 3, 0, 1).contiguous()
    return F.conv2d(_same_pad(x, k, stride), wt, None, stride, groups=c)


def batch_norm(x, w, prefix):
    g, b = _t(w[prefix + "/gamma"]), _t(w[prefix + "/beta"])
    m, v = _t(w[prefix + "/moving_mean"]), _t(w[prefix + "/moving_variance"])
    scale = g * torch.rsqrt(v + EPS)
    return (x - m[None, :, None, None]) * scale[None, :, None, None] + b[None, :, None, None]


def swish(x):
    return x * torch.sigmoid(x)


def activation(params):
    """utils.activation_fn(features, act_type) as a function of the tensor (utils.py:42-59)."""
    name = params.get("act_type", "swish")
    if name in ("silu", "swish", "swish_native"):
        return swish
    if name == "hswish":
        return lambda x: x * F.relu6(x + 3) / 6
    if name == "relu":
        return F.relu
    if name == "relu6":
        return F.relu6
    if name == "mish":
        return lambda x: x * torch.tanh(F.softplus(x))
    raise ValueError("Unsupported act_type {}".format(name))


def max_pool_same(x, k, s):
    return F.max_pool2d(_same_pad(x, k, s, value=float("-inf")), k, s)


def nearest_upsample(x, th, tw):
    """resize_nearest_neighbor, align_corners=False, half_pixel_centers=False:
    src = min(floor(dst * in/out), in-1)."""
    H, W = x.shape[-2:]
    ys = torch.clamp(torch.floor(torch.arange(th, dtype=torch.float32) * (H / th)).long(), max=H - 1)
    xs = torch.clamp(torch.floor(torch.arange(tw, dtype=torch.float32) * (W / tw)).long(), max=W - 1)
    return x[:, :, ys][:, :, :, xs]


def _drop(x, masks, site, t):
    if masks is None or site not in masks:
        return x
    return x * _t(masks[site][:, t])[:, :, None, None]


# ---------------------------------------------------------------- network pieces
def _tap(taps, name, x):
    if taps is not None:
        taps[name] = x.permute(0, 2, 3, 1).contiguous().numpy()


def backbone(w, params, x, masks, t, taps=None):
    """Returns block outputs at reduction_1..5 (list of NCHW tensors)."""
    bb = params["backbone_name"]
    table = block_table(bb)
    act = activation(params)
    x = act(batch_norm(conv2d(x, w[bb + "/stem/conv2d/kernel"], 2),
                       w, bb + "/stem/tpu_batch_normalization"))
    _tap(taps, "stem", x)
    feats = []
    for i, (k, s, e, cin, cout, _se) in enumerate(table):
        p = "%s/blocks_%d/" % (bb, i)
        inp = x
        nb = 0
        bn_name = lambda j: p + "tpu_batch_normalization" + ("" if j == 0 else "_%d" % j)
        if e != 1:
            x = act(batch_norm(conv2d(x, w[p + "conv2d/kernel"]), w, bn_name(nb)))
            nb += 1
            x = _drop(x, masks, "blocks_%d/expand" % i, t)
            _tap(taps, "blocks_%d/expand" % i, x)
            proj = p + "conv2d_1/kernel"
        else:
            proj = p + "conv2d/kernel"
        x = act(batch_norm(depthwise(x, w[p + "depthwise_conv2d/depthwise_kernel"], s),
                           w, bn_name(nb)))
        nb += 1
        x = _drop(x, masks, "blocks_%d/dw" % i, t)
        _tap(taps, "blocks_%d/dw" % i, x)
        # squeeze-excite: global mean -> 1x1+bias -> swish -> 1x1+bias -> sigmoid -> scale
        sq = x.mean(dim=(2, 3), keepdim=True)
        sq = act(conv2d(sq, w[p + "se/conv2d/kernel"], 1, w[p + "se/conv2d/bias"]))
        sq = conv2d(sq, w[p + "se/conv2d_1/kernel"], 1, w[p + "se/conv2d_1/bias"])
        _tap(taps, "blocks_%d/se" % i, torch.sigmoid(sq))
        x = torch.sigmoid(sq) * x
        x = batch_norm(conv2d(x, w[proj]), w, bn_name(nb))
        if s == 1 and cin == cout:
            x = x + inp
        _tap(taps, "blocks_%d/out" % i, x)
        if i == len(table) - 1 or table[i + 1][1] > 1:
            feats.append(x)
    return feats


def _resample(w, prefix, feat, th, tw, F_ch, params=None):
    """ResampleFeatureMap.call (efficientdet_keras.py:313-350)."""
    H, W = feat.shape[-2:]
    params = params or {}
    apply_bn = params.get("apply_bn_for_resampling", True)
    after = params.get("conv_after_downsample", False)

    def maybe_1x1(f):
        if f.shape[1] != F_ch:
            f = conv2d(f, w[prefix + "/conv2d/kernel"], 1, w[prefix + "/conv2d/bias"])
            if apply_bn:
                f = batch_norm(f, w, prefix + "/bn")
        return f

    if H > th and W > tw:
        if not after:
            feat = maybe_1x1(feat)
        sh, sw = (H - 1) // th + 1, (W - 1) // tw + 1
        assert sh == sw, "square pooling windows only"
        feat = max_pool_same(feat, sh + 1, sh)
        if after:
            feat = maybe_1x1(feat)
    elif H <= th and W <= tw:
        feat = maybe_1x1(feat)
        if H < th or W < tw:
            feat = nearest_upsample(feat, th, tw)
    else:
        raise ValueError("Incompatible Resampling")
    return feat


def _sepconv(x, w, prefix, dwk="depthwise_kernel", pwk="pointwise_kernel", use_bias=True):
    x = depthwise(x, w[prefix + "/" + dwk], 1)
    return conv2d(x, w[prefix + "/" + pwk], 1, w[prefix + "/bias"] if use_bias else None)


def bifpn_nodes(min_level, max_level):
    n = max_level - min_level + 1
    ids = {min_level + i: [i] for i in range(n)}
    nodes, nxt = [], n
    for lvl in range(max_level - 1, min_level - 1, -1):
        nodes.append((lvl, [ids[lvl][-1], ids[lvl + 1][-1]]))
        ids[lvl].append(nxt)
        nxt += 1
    for lvl in range(min_level + 1, max_level + 1):
        nodes.append((lvl, ids[lvl] + [ids[lvl - 1][-1]]))
        ids[lvl].append(nxt)
        nxt += 1
    return nodes


def fpn(w, params, feats, taps=None):
    F_ch, lo, hi = params["fpn_num_filters"], params["min_level"], params["max_level"]
    nodes = bifpn_nodes(lo, hi)
    method = params.get("fpn_weight_method") or "fastattn"
    if params.get("fpn_config"):
        nodes = [(int(n["feat_level"]), [int(o) for o in n["inputs_offsets"]]) for n in params["fpn_config"]["nodes"]]
        method = params["fpn_config"].get("weight_method") or "fastattn"
    act = activation(params)
    cba = bool(params.get("conv_bn_act_pattern", False))
    for rep in range(params["fpn_cell_repeats"]):
        cell = list(feats)
        for n, (lvl, offsets) in enumerate(nodes):
            p = "fpn_cells/cell_%d/fnode%d/" % (rep, n)
            nf = len(cell)
            th, tw = cell[lvl - lo].shape[-2:]
            ins = [_resample(w, p + "resample_%d_%d_%d" % (i, off, nf), cell[off], th, tw, F_ch, params)
                   for i, off in enumerate(offsets)]
            if method == "attn":
                ew = torch.softmax(torch.stack([_t(w[p + "WSM" + ("" if i == 0 else "_%d" % i)]).reshape(())
                                                for i in range(len(ins))]), 0)
                new = (torch.stack(ins, -1) * ew).sum(-1)
            elif method == "fastattn":
                ew = [torch.relu(_t(w[p + "WSM" + ("" if i == 0 else "_%d" % i)]))
                      for i in range(len(ins))]
                tot = ew[0]
                for e in ew[1:]:
                    tot = tot + e
                new = None
                for xi, e in zip(ins, ew):
                    term = xi * e / (tot + 0.0001)
                    new = term if new is None else new + term
            elif method == "sum":
                new = ins[0]
                for xi in ins[1:]:
                    new = new + xi
            else:
                raise ValueError("unknown weight_method %s" % method)
            op = p + "op_after_combine%d" % nf
            if not cba:
                new = act(new)
            _tap(taps, "cell%d/fnode%d/fused" % (rep, n), new)
            new = batch_norm(_sepconv(new, w, op + "/conv", use_bias=not cba), w, op + "/bn")
            if cba:
                new = act(new)
            _tap(taps, "cell%d/fnode%d/out" % (rep, n), new)
            cell.append(new)
        feats = []
        for lvl in range(lo, hi + 1):
            for i, (nl, _) in enumerate(reversed(nodes)):
                if nl == lvl:
                    feats.append(cell[-1 - i])
                    break
    return feats


def head(w, params, feats, net, tag, masks, t):
    outs = []
    lo = params["min_level"]
    act = activation(params)
    for li, x in enumerate(feats):
        for i in range(params["box_class_repeats"]):
            x = _sepconv(x, w, "%s/%s-%d" % (net, tag, i))
            x = act(batch_norm(x, w, "%s/%s-%d-bn-%d" % (net, tag, i, lo + li)))
            x = _drop(x, masks, "%s-%d-%d" % (tag, i, lo + li), t)
        outs.append(_sepconv(x, w, "%s/%s-predict" % (net, tag)))
    return outs


def forward_once(w, params, images, masks=None, t=0, taps=None):
    """One forward pass. images float32 [N,H,W,3] -> (cls[5], box[5]) as NHWC numpy."""
    x = _t(images).permute(0, 3, 1, 2)
    with torch.no_grad():
        feats = backbone(w, params, x, masks, t, taps)[params["min_level"] - 1:]
        F_ch = params["fpn_num_filters"]
        for lvl in range(len(feats) + params["min_level"], params["max_level"] + 1):
            h, wd = feats[-1].shape[-2:]
            feats.append(_resample(w, "resample_p%d" % lvl, feats[-1],
                                   (h + 1) // 2, (wd + 1) // 2, F_ch, params))
            _tap(taps, "p%d_in" % lvl, feats[-1])
        pyr = fpn(w, params, feats, taps)
        cls = head(w, params, pyr, "class_net", "class", masks, t)
        box = head(w, params, pyr, "box_net", "box", masks, t)
    nhwc = lambda v: v.permute(0, 2, 3, 1).contiguous().numpy()
    return [nhwc(c) for c in cls], [nhwc(b) for b in box]


def forward(w, params, images, masks=None):
    """EfficientDetNet.call: MC branch repeats the WHOLE network T times and
    stacks per level on a new leading axis for the heads whose rate (or the
    global rate) is non-zero; the other head keeps the LAST iteration's output
    (efficientdet_keras.py:981-1050)."""
    if not params["mc_dropout"]:
        return forward_once(w, params, images, None)
    T = int(params["mc_dropoutsamp"])
    stack_cls = bool(params["mc_classheadrate"] or params["mc_dropoutrate"])
    stack_box = bool(params["mc_boxheadrate"] or params["mc_dropoutrate"])
    all_cls, all_box = [], []
    for t in range(T):
        c, b = forward_once(w, params, images, masks, t)
        all_cls.append(c)
        all_box.append(b)
    cls = ([np.stack([all_cls[t][l] for t in range(T)], 0) for l in range(len(all_cls[0]))]
           if stack_cls else all_cls[-1])
    box = ([np.stack([all_box[t][l] for t in range(T)], 0) for l in range(len(all_box[0]))]
           if stack_box else all_box[-1])
    return cls, box


def dropout_sites(params):
    """Ordered [(site name, channels, rate)] — the site index is the position in
    this list (shared convention with the HIP path)."""
    if not params["mc_dropout"]:
        return []
    base = float(params["mc_dropoutrate"] or 0.0)
    rc = float(params["mc_classheadrate"] or base)
    rb = float(params["mc_boxheadrate"] or base)
    sites = []
    for i, (k, s, e, cin, cout, _se) in enumerate(block_table(params["backbone_name"])):
        if e != 1:
            sites.append(("blocks_%d/expand" % i, cin * e, base))
        sites.append(("blocks_%d/dw" % i, cin * e, base))
    F_ch = params["fpn_num_filters"]
    for tag, r in (("class", rc), ("box", rb)):
        for i in range(params["box_class_repeats"]):
            for lvl in range(params["min_level"], params["max_level"] + 1):
                sites.append(("%s-%d-%d" % (tag, i, lvl), F_ch, r))
    return sites
