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
