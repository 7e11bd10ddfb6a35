"""ORACLE (test infrastructure): CalibrateBoxUncert.calibrate_boxuncert restated in numpy
(reference src/utils_box.py:404-524), one image at a time like the reference ([M,4] uncert, [M] classes, [M,4] boxes).

An isotonic model is its (X_thresholds_, y_thresholds_) table; `predict` follows sklearn's
IsotonicRegression(out_of_bounds="clip"): clip to the fitted range, interpolate linearly (float64), cast to the
input dtype.  tests/test_oracle_kats.py pins this against the real sklearn class on fitted models.
"""
import numpy as np


def iso_predict(table, x):
    xs, ys = table
    x = np.asarray(x)
    t = np.clip(x.astype(np.float64), xs[0], xs[-1])
    if len(xs) == 1:
        return np.full(x.shape, ys[0]).astype(x.dtype)
    return np.interp(t, xs, ys).astype(x.dtype)


def calibrate_boxuncert(method, models, num_classes, uncert, classes, boxes):
    uncert = np.nan_to_num(np.asarray(uncert, np.float32))
    m = models[method]
    if method == "ts_all":
        return uncert / np.float32(m)
    if method == "ts_percoo":
        return np.swapaxes([uncert[:, j] / np.float32(m[j]) for j in range(4)], 0, 1)
    if method == "iso_all":
        return iso_predict(m, uncert.flatten()).reshape([-1, 4])
    if method == "iso_percoo":
        return np.swapaxes([iso_predict(m[j], uncert[:, j]) for j in range(4)], 0, 1)
    cal = [[m[ci * 4 + j] for j in range(4)] for ci in range(num_classes)]
    if method == "iso_perclscoo":
        out = np.zeros_like(uncert)
        for ci in range(1, num_classes + 1):
            sel = classes.astype(int) == ci
            if np.any(sel):
                for j in range(4):
                    out[:, j][sel] = iso_predict(cal[ci - 1][j], uncert[:, j][sel])
        return out
    if method == "rel_iso_perclscoo":
        width = np.asarray(boxes[:, 3] - boxes[:, 1])
        height = np.asarray(boxes[:, 2] - boxes[:, 0])
        norm = np.swapaxes([height, width, height, width], 0, 1)
        rel = np.divide(uncert, norm, out=np.zeros_like(uncert), where=norm != 0, dtype=np.float16)
        out = np.zeros_like(uncert)
        for ci in range(1, num_classes + 1):
            sel = classes.astype(int) == ci
            if np.any(sel):
                for j in range(4):
                    out[:, j][sel] = iso_predict(cal[ci - 1][j], rel[:, j][sel])
        return out * norm
    raise ValueError("Unknown calibration method")


# ------------------------------------------------------------------ class calibration (utils_class.py:109-187)
def _stable_softmax(logits):
    out = []
    for x in np.asarray(logits, np.float32):
        e = np.exp(x - max(x))
        out.append(e / np.sum(e))
    return np.asarray(out, np.float32)


def philox_normal(seed, i0, i1, i2, tag):
    """The build's standard-normal stream (csrc/uda_internal.h philox_normal), vectorised: Box-Muller on two 24-bit
    uniforms of Philox4x32-10(counter = (i0, i1, i2, tag), key = seed)."""
    from . import philox_ref
    w = philox_ref.philox4x32_10(i0, i1, i2, np.uint32(tag), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u1 = ((w[0] >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24
    u2 = (w[1] >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def perform_class_calib(method, models, logits, uncert=None, draws=10, seed=0):
    """logits [R, C] (all selected rows of the batch, row-major), uncert [R, C] MC std of the logits or None ->
    (entropy [R], probab [R, C][, uncert [R, C]]).  Sampling uses the build's Philox stream where the reference calls
    tfp Normal(...).sample(10)."""
    logits = np.asarray(logits, np.float32)
    R, C = logits.shape
    if uncert is not None:
        r = np.arange(R, dtype=np.uint32)[None, :, None]
        c = np.arange(C, dtype=np.uint32)[None, None, :]
        d = np.arange(draws, dtype=np.uint32)[:, None, None]
        z = philox_normal(seed, c + 0 * r + 0 * d, r + 0 * c + 0 * d, d + 0 * r + 0 * c, 0x5A).astype(np.float32)
        samp = (logits[None] + np.asarray(uncert, np.float32)[None] * z).reshape(-1, C).astype(np.float32)
    else:
        samp = logits
    m = models[method]
    if method.startswith("ts"):
        t = np.full((C,), m, np.float32) if method == "ts_all" else np.asarray(m, np.float32)
        prob = _stable_softmax(samp / t)
    else:
        p = _stable_softmax(samp)
        if method == "iso_all":
            post = iso_predict(m, p.flatten()).reshape(p.shape)
        else:
            post = np.stack([iso_predict(m[i], p[:, i]) for i in range(C)], axis=1)
        prob = (post / np.stack([np.sum(post, axis=-1)] * C, axis=-1)).astype(np.float32)
    if uncert is not None:
        prob = prob.reshape([draws, -1, C])
        new_unc = np.std(prob, axis=0)
        prob = np.mean(prob, axis=0)
    ent = -np.sum(prob * np.nan_to_num(np.log2(np.maximum(prob, 10 ** -7))), axis=1)
    return (ent, prob, new_unc) if uncert is not None else (ent, prob)
