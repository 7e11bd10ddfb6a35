"""ORACLE (test infrastructure): post-processing of the hot path on the CPU.

  anchors                      src/anchors.py:100-218
  decode_box_outputs           src/anchors.py:41-75
  decode_uncert (l-norm, falsedec; n-flow == l-norm analytically)   src/utils_box.py:105-276
  get_mcuncert (mean, population std over T)                         src/utils_extra.py:220-244
  merge levels / argmax-or-topk / pre_nms                            src/postprocess.py:75-339
  nms (NonMaxSuppressionV5 + gathers)                                src/postprocess.py:342-420
  postprocess_global                                                 src/postprocess.py:472-621
  per_class_nms / postprocess_per_class                              src/postprocess.py:624-740

Numerics the build pins down where TF leaves them open (so the HIP path can be
compared bit for bit on identical head outputs):
  * reductions over the MC axis are sequential float32 sums t = 0..T-1, mean = sum / T,
    std = sqrt(mean((x - mean)^2))  (population)
  * sigmoid(x) := float32(1 / (1 + exp(-float64(x)))); float32 exp(x) := float32(exp(float64(x)))
  * top_k orders by value descending, ties -> lower index
Values are **parity unpinned** against TF (not installable); see oracle/__init__.py.

Layout of head outputs consumed here (as `effdet_ref.forward` returns them):
  cls: list over levels of [N,h,w,A*C] or [T,N,h,w,A*C];  box: [.., 4A] or [.., 8A] (loss attenuation)
"""
import ctypes
import os

import numpy as np

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libpost_ref.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.dirname(os.path.abspath(__file__))])
        lib = ctypes.CDLL(path)
        lib.oracle_nms_v5.restype = ctypes.c_int
        lib.oracle_nms_v5.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p]
        lib.oracle_sigmoid.restype = None
        lib.oracle_sigmoid.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        lib.oracle_iou.restype = ctypes.c_float
        lib.oracle_iou.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _LIB = lib
    return _LIB


# ------------------------------------------------------------------ helpers
def parse_image_size(image_size):
    if isinstance(image_size, int):
        return image_size, image_size
    if isinstance(image_size, str):
        w, h = image_size.lower().split("x")
        return int(h), int(w)
    return int(image_size[0]), int(image_size[1])


def feat_sizes(image_size, max_level):
    h, w = parse_image_size(image_size)
    out = [(h, w)]
    for _ in range(max_level):
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        out.append((h, w))
    return out


def sigmoid32(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    _lib().oracle_sigmoid(x.ctypes.data, y.ctypes.data, x.size)
    return y


def exp32(x):
    return np.exp(np.asarray(x, dtype=np.float32).astype(np.float64)).astype(np.float32)


def seq_mean(x):
    """Sequential float32 mean over axis 0."""
    acc = x[0].astype(np.float32).copy()
    for t in range(1, x.shape[0]):
        acc = acc + x[t]
    return (acc / np.float32(x.shape[0])).astype(np.float32)


def seq_mean_std(x):
    m = seq_mean(x)
    acc = np.zeros_like(m)
    for t in range(x.shape[0]):
        d = x[t] - m
        acc = acc + d * d
    return m, np.sqrt(acc / np.float32(x.shape[0])).astype(np.float32)


# ------------------------------------------------------------------ anchors (a11)
def anchor_boxes(params):
    """[A_tot, 4] float32 (ymin, xmin, ymax, xmax); order level, y, x, (octave, aspect)."""
    lo, hi = params["min_level"], params["max_level"]
    H, W = parse_image_size(params["image_size"])
    fs = feat_sizes(params["image_size"], hi)
    scales = params["anchor_scale"]
    if not isinstance(scales, (list, tuple)):
        scales = [scales] * (hi - lo + 1)
    out = []
    for lvl in range(lo, hi + 1):
        sy, sx = fs[0][0] / float(fs[lvl][0]), fs[0][1] / float(fs[lvl][1])
        per_cfg = []
        for octave in range(params["num_scales"]):
            for aspect in params["aspect_ratios"]:
                o = octave / float(params["num_scales"])
                bx = scales[lvl - lo] * sx * 2 ** o
                by = scales[lvl - lo] * sy * 2 ** o
                if isinstance(aspect, (list, tuple)):
                    ax, ay = aspect
                else:
                    ax = np.sqrt(aspect)
                    ay = 1.0 / ax
                hx, hy = bx * ax / 2.0, by * ay / 2.0
                xs = np.arange(sx / 2, W, sx)
                ys = np.arange(sy / 2, H, sy)
                xv, yv = np.meshgrid(xs, ys)
                xv, yv = xv.reshape(-1), yv.reshape(-1)
                b = np.stack([yv - hy, xv - hx, yv + hy, xv + hx], axis=1)
                per_cfg.append(b[:, None, :])
        out.append(np.concatenate(per_cfg, axis=1).reshape(-1, 4))
    return np.vstack(out).astype(np.float32)


# ------------------------------------------------------------------ decode (a12, a13)
def decode_box_outputs(pred, anchors):
    """float32 throughout; exp via exp32."""
    pred = pred.astype(np.float32)
    a = anchors.astype(np.float32)
    two = np.float32(2.0)
    ya, xa = (a[..., 0] + a[..., 2]) / two, (a[..., 1] + a[..., 3]) / two
    ha, wa = a[..., 2] - a[..., 0], a[..., 3] - a[..., 1]
    ty, tx, th, tw = [pred[..., i] for i in range(4)]
    w = exp32(tw) * wa
    h = exp32(th) * ha
    yc = ty * ha + ya
    xc = tx * wa + xa
    return np.stack([yc - h / two, xc - w / two, yc + h / two, xc + w / two], axis=-1).astype(np.float32)


DECODE_SEED_XOR = 0x5DEC0DE5A3B1E5          # csrc/uda_api.hip run_candidates: decode_seed = dropout seed ^ this


def decode_uncert(pred, sigma, anchors, method="l-norm", sample=None):
    """float64 inside, cast back to float32 (utils_box.py:122-137,268-271).  method "sample" (:162-184): moments over
    `nsamples` decoded Normal draws; the draws come from the build's Philox stream instead of TFP's
    (sample = dict(nsamples, seed, id0 [..] anchor index, id1 [..] global sample row), broadcastable to pred[..., 0])."""
    a = anchors.astype(np.float64)
    ya, xa = (a[..., 0] + a[..., 2]) / 2, (a[..., 1] + a[..., 3]) / 2
    ha, wa = a[..., 2] - a[..., 0], a[..., 3] - a[..., 1]
    p = pred.astype(np.float64)
    ty, tx, th, tw = [p[..., i] for i in range(4)]
    var = np.square(sigma.astype(np.float64))
    dty, dtx, dth, dtw = [var[..., i] for i in range(4)]
    if method in ("l-norm", "n-flow"):
        w = np.exp(tw + dtw / 2) * wa
        h = np.exp(th + dth / 2) * ha
        yc = ty * ha + ya
        xc = tx * wa + xa
        dw = (np.exp(dtw) - 1) * np.exp(2 * tw + dtw) * (wa * wa)
        dh = (np.exp(dth) - 1) * np.exp(2 * th + dth) * (ha * ha)
        dyc = dty * (ha * ha)
        dxc = dtx * (wa * wa)
        dymin = dyc + dh / 4.0
        dxmin = dxc + dw / 4.0
        dymax, dxmax = dymin, dxmin
    elif method == "falsedec":
        w = np.exp(tw) * wa
        h = np.exp(th) * ha
        yc = ty * ha + ya
        xc = tx * wa + xa
        dw = np.exp(dtw) * wa
        dh = np.exp(dth) * ha
        dyc = dty * ha + ya
        dxc = dtx * wa + xa
        dymin = np.abs(dyc - dh / 2.0)
        dxmin = np.abs(dxc - dw / 2.0)
        dymax = dyc + dh / 2.0
        dxmax = dxc + dw / 2.0
    elif method == "sample":
        from . import philox_ref
        S = int(sample["nsamples"])
        shape = np.broadcast(ty, ya).shape
        id0 = np.broadcast_to(np.asarray(sample["id0"], np.uint32), shape)
        id1 = np.broadcast_to(np.asarray(sample["id1"], np.uint32), shape)
        sc = [np.sqrt(v) for v in (dty, dtx, dth, dtw)]
        corners = np.empty((S,) + shape + (4,), np.float64)
        for s_ in range(S):
            zy, zx = philox_ref.normal2(sample["seed"], id0, id1, np.uint32(2 * s_), 0xD5)
            zh, zw = philox_ref.normal2(sample["seed"], id0, id1, np.uint32(2 * s_ + 1), 0xD5)
            sy, sx, sh, sw = ty + sc[0] * zy, tx + sc[1] * zx, th + sc[2] * zh, tw + sc[3] * zw
            ws, hs = np.exp(sw) * wa, np.exp(sh) * ha
            ycs, xcs = sy * ha + ya, sx * wa + xa
            corners[s_] = np.stack([ycs - hs / 2.0, xcs - ws / 2.0, ycs + hs / 2.0, xcs + ws / 2.0], -1)
        mean = corners.mean(0)
        var = ((corners - mean) ** 2).mean(0)          # tf.nn.moments: population variance
        return mean.astype(np.float32), np.sqrt(var).astype(np.float32)
    else:
        raise ValueError("unknown decode method %r" % method)
    coords = np.stack([yc - h / 2.0, xc - w / 2.0, yc + h / 2.0, xc + w / 2.0], -1).astype(np.float32)
    unc = np.sqrt(np.stack([dymin, dxmin, dymax, dxmax], -1)).astype(np.float32)
    return coords, unc


# ------------------------------------------------------------------ level merge (a9)
def _merge(levels, last):
    """list of [..., h, w, A*last] -> [..., A_tot, last] (anchor = ((y*w)+x)*A + a)."""
    lead = levels[0].shape[:-3]
    return np.concatenate([l.reshape(lead + (-1, last)) for l in levels], axis=len(lead))


def mc_layout(params):
    stacked_c = bool(params["mc_dropout"] and (params["mc_classheadrate"] or params["mc_dropoutrate"]))
    stacked_b = bool(params["mc_dropout"] and (params["mc_boxheadrate"] or params["mc_dropoutrate"]))
    return stacked_c, stacked_b


# ------------------------------------------------------------------ NMS (a15)
def nms_v5(boxes, scores, max_out, iou_thr, score_thr, soft_sigma, pad):
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    idx = np.zeros(max_out, dtype=np.int32)
    sc = np.zeros(max_out, dtype=np.float32)
    valid = _lib().oracle_nms_v5(boxes.ctypes.data, scores.ctypes.data, len(scores), max_out,
                                 iou_thr, score_thr, soft_sigma, 1 if pad else 0,
                                 idx.ctypes.data, sc.ctypes.data)
    if not pad:
        idx, sc = idx[:valid], sc[:valid]
    return idx, sc, np.int32(valid)


def nms_v5_py(boxes, scores, max_out, iou_thr, score_thr, soft_sigma, pad):
    """Pure-Python twin of csrc/post_ref.c (small cases; cross-checks the C build)."""
    import heapq
    f32 = np.float32
    boxes = np.asarray(boxes, dtype=f32)
    heap = [(-float(s), i, 0) for i, s in enumerate(np.asarray(scores, dtype=f32)) if s > f32(score_thr)]
    heapq.heapify(heap)
    soft = soft_sigma > 0
    scale = f32(-0.5) / f32(soft_sigma) if soft else f32(0)
    sel, sel_sc = [], []
    lib = _lib()
    while len(sel) < max_out and heap:
        negs, i, begin = heapq.heappop(heap)
        score = f32(-negs)
        orig = score
        hard = False
        for j in range(len(sel) - 1, begin - 1, -1):
            sim = f32(lib.oracle_iou(boxes[i].ctypes.data, boxes[sel[j]].ctypes.data))
            if soft or sim <= f32(iou_thr):
                wgt = f32(np.exp(np.float64(f32(f32(scale * sim) * sim))))
            else:
                wgt = f32(0)
            score = f32(score * wgt)
            if not soft and sim > f32(iou_thr):
                hard = True
                break
            if score <= f32(score_thr):
                break
        if not hard:
            if score == orig:
                sel.append(i)
                sel_sc.append(score)
                continue
            if score > f32(score_thr):
                heapq.heappush(heap, (-float(score), i, len(sel)))
    valid = len(sel)
    if pad:
        sel += [0] * (max_out - valid)
        sel_sc += [f32(0)] * (max_out - valid)
    return np.asarray(sel, dtype=np.int32), np.asarray(sel_sc, dtype=f32), np.int32(valid)


def nms_params(params):
    """(sigma/2, iou_thresh, score_thresh) as postprocess.nms derives them (:373-398)."""
    cfg = params["nms_configs"]
    method = cfg["method"]
    if method == "hard" or not method:
        return 0.0, cfg["iou_thresh"] or 0.5, cfg["score_thresh"] or float("-inf")
    if method == "gaussian":
        return (cfg["sigma"] or 0.5) / 2, 0.5, cfg["score_thresh"] or 0.001
    raise ValueError("Inference has invalid nms method {}".format(method))


# ------------------------------------------------------------------ pre-NMS (a8-a14)
def pre_nms(params, cls_outputs, box_outputs, decode_seed=0, first_image=0):
    """-> dict(boxes [N,K,4], scores [N,K], classes [N,K] int32, logits [N,K,C],
               u_cls [N,K,C]|None, u_al [N,K,4]|None, u_ep [N,K,4]|None, indices [N,K])"""
    C = params["num_classes"]
    stacked_c, stacked_b = mc_layout(params)
    loss_att = bool(params["loss_attenuation"])
    anchors = anchor_boxes(params)

    u_cls = None
    if stacked_c:
        ms = [seq_mean_std(l) for l in cls_outputs]
        cls_levels = [m for m, _ in ms]
        u_cls = _merge([s for _, s in ms], C)
    else:
        cls_levels = cls_outputs
    cls_all = _merge(cls_levels, C)                                # [N, A, C]

    if loss_att:
        half = box_outputs[0].shape[-1] // 2
        box_all = _merge([b[..., :half] for b in box_outputs], 4)  # [(T,) N, A, 4]
        sig_all = _merge([b[..., half:] for b in box_outputs], 4)
    else:
        box_all = _merge(box_outputs, 4)
        sig_all = None

    N, A = cls_all.shape[:2]
    k = int(params["nms_configs"].get("max_nms_inputs", 0) or 0)
    if k > 0:
        flat = cls_all.reshape(N, -1)
        order = np.argsort(-flat, axis=1, kind="stable")[:, :k]    # desc, ties -> lower index
        indices = (order // C).astype(np.int32)
        classes = (order % C).astype(np.int32)
        top_logit = np.take_along_axis(flat, order, 1)
        logits = np.take_along_axis(cls_all, indices[..., None], 1)
        if u_cls is not None:
            # the reference gathers one value per (anchor, class) pair here (:117-121)
            u_cls = np.take_along_axis(u_cls.reshape(N, -1), order, 1)[..., None]
        gather = lambda v: (np.take_along_axis(v, indices[None, ..., None], v.ndim - 2)
                            if v.ndim == 4 else np.take_along_axis(v, indices[..., None], 1))
        box_all = gather(box_all)
        if sig_all is not None:
            sig_all = gather(sig_all)
        anc = anchors[indices]                                     # [N, K, 4]
    else:
        classes = np.argmax(cls_all, axis=-1).astype(np.int32)     # first max on ties
        top_logit = np.max(cls_all, axis=-1)
        indices = np.tile(np.arange(A, dtype=np.int32)[None], (N, 1))
        logits = cls_all
        anc = anchors[None]

    scores = sigmoid32(top_logit)
    method = params["uncert_adjust_method"]
    u_al = u_ep = None

    def samp(t, T):      # ids of the Philox normal stream of the "sample" decode: (anchor, global sample row)
        if method != "sample":
            return None
        rows = ((np.arange(N, dtype=np.int64) + first_image) * T + t).astype(np.uint32)[:, None]
        return dict(nsamples=params.get("decode_nsamples", 100), seed=int(decode_seed) ^ DECODE_SEED_XOR, id0=indices.astype(np.uint32), id1=rows)
    if stacked_b:
        T = box_all.shape[0]
        if loss_att:
            dec = [decode_uncert(box_all[t], sig_all[t], anc, method, samp(t, T)) for t in range(T)]
            boxes_t = np.stack([d[0] for d in dec])
            u_al = seq_mean(np.stack([d[1] for d in dec]))
        else:
            boxes_t = np.stack([decode_box_outputs(box_all[t], anc) for t in range(T)])
        boxes, u_ep = seq_mean_std(boxes_t)
    elif loss_att:
        boxes, u_al = decode_uncert(box_all, sig_all, anc, method, samp(0, 1))
    else:
        boxes = decode_box_outputs(box_all, anc)
    return dict(boxes=boxes, scores=scores, classes=classes, logits=logits.astype(np.float32),
                u_cls=u_cls, u_al=u_al, u_ep=u_ep, indices=indices)


# ------------------------------------------------------------------ global mode (a16)
def postprocess_global(params, cls_outputs, box_outputs, image_scales=None, decode_seed=0, first_image=0):
    """Returns the reference's output tuple:
    (boxes [N,M,4(+4 al)(+4 ep)], scores [N,M], classes [N,M] or [N,M,1+C], valid_len [N]
     [, logits [N,M,C] if enable_softmax])"""
    p = pre_nms(params, cls_outputs, box_outputs, decode_seed, first_image)
    sigma2, iou_thr, score_thr = nms_params(params)
    M = params["nms_configs"]["max_output_size"]
    has_unc = bool(params["loss_attenuation"] or params["mc_dropout"])
    N = p["boxes"].shape[0]
    H, W = parse_image_size(params["image_size"])
    out_b, out_s, out_c, out_v, out_l = [], [], [], [], []
    out_u = {"u_cls": [], "u_al": [], "u_ep": []}
    for n in range(N):
        idx, sc, valid = nms_v5(p["boxes"][n], p["scores"][n], M, iou_thr, score_thr, sigma2, True)
        out_b.append(p["boxes"][n][idx])
        out_s.append(sc)
        out_c.append((p["classes"][n][idx] + 1).astype(np.float32))
        out_v.append(valid)
        out_l.append(p["logits"][n][idx])
        for key in out_u:
            if p[key] is not None:
                out_u[key].append(p[key][n][idx])
    boxes = np.stack(out_b)
    boxes = np.clip(boxes, np.float32(0), np.array([H, W, H, W], dtype=np.float32))
    unc = {k: (np.stack(v) if v else None) for k, v in out_u.items()}
    if image_scales is not None:
        s = np.asarray(image_scales, dtype=np.float32)[:, None, None]
        boxes = boxes * s
        for key in ("u_al", "u_ep"):
            if unc[key] is not None:
                unc[key] = unc[key] * s
    classes = np.stack(out_c)
    if has_unc:
        if unc["u_cls"] is not None:
            classes = np.concatenate([classes[..., None], unc["u_cls"]], -1)
        if unc["u_al"] is not None:
            boxes = np.concatenate([boxes, unc["u_al"]], -1)
        if unc["u_ep"] is not None:
            boxes = np.concatenate([boxes, unc["u_ep"]], -1)
    out = [boxes.astype(np.float32), np.stack(out_s), classes.astype(np.float32),
           np.asarray(out_v, dtype=np.int32)]
    if params["enable_softmax"]:
        out.append(np.stack(out_l))
    return tuple(out)


# ------------------------------------------------------------------ per-class mode (a17)
def postprocess_per_class(params, cls_outputs, box_outputs, image_scales=None):
    """(boxes [N,M,4], scores [N,M], classes [N,M], valid_len [N]).  Uncertainties are
    dropped (:737); no clipping.  The reference's logits output in this mode is corrupted
    by a variable overwrite when >1 class is present (:659-666) and is not restated."""
    p = pre_nms(params, cls_outputs, box_outputs)
    sigma2, iou_thr, score_thr = nms_params(params)
    M = params["nms_configs"].get("max_output_size", 100)
    N = p["boxes"].shape[0]
    res_b, res_s, res_c, res_v = [], [], [], []
    for n in range(N):
        bs, ss, cs, tot = [], [], [], 0
        for c in range(params["num_classes"]):
            sel = np.nonzero(p["classes"][n] == c)[0]
            if sel.size == 0:
                continue
            idx, sc, valid = nms_v5(p["boxes"][n][sel], p["scores"][n][sel], M, iou_thr,
                                    score_thr, sigma2, False)
            bs.append(p["boxes"][n][sel][idx])
            ss.append(sc)
            cs.append(np.full(len(idx), c + 1, dtype=np.float32))
            tot += int(valid)
        b = np.concatenate(bs + [np.zeros((M, 4), np.float32)])
        s = np.concatenate(ss + [np.zeros(M, np.float32)])
        c = np.concatenate(cs + [np.zeros(M, np.float32)])
        top = np.argsort(-s, kind="stable")[:M]
        res_b.append(b[top])
        res_s.append(s[top])
        res_c.append(c[top])
        res_v.append(min(M, tot))
    boxes = np.stack(res_b)
    if image_scales is not None:
        boxes = boxes * np.asarray(image_scales, dtype=np.float32)[:, None, None]
    return boxes.astype(np.float32), np.stack(res_s), np.stack(res_c), np.asarray(res_v, np.int32)
