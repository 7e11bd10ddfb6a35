"""Calibrated box uncertainty and calibrated class probabilities on top of the driver.

Mirrors `CalibrateBoxUncert` of the reference (src/utils_box.py:279-524): the calibration models it loads are a
temperature (one, or one per box coordinate) or fitted `sklearn.isotonic.IsotonicRegression(out_of_bounds="clip")`
objects (one for all values, one per coordinate, or one per (class, coordinate), optionally on the uncertainty
relative to the box size).  Here a fitted isotonic model is reduced to its thresholds table, and the lookups of
the <= 100 selected rows per image run on the device (`uda_calibrate_box`) next to the outputs they refine.
"""
import ctypes as C

import numpy as np

from . import capi

METHODS = ("ts_all", "ts_percoo", "iso_all", "iso_percoo", "iso_perclscoo", "rel_iso_perclscoo")


class IsoTable:
    """The thresholds of a fitted isotonic regression: predict(x) = linear interpolation, clipped to the range."""

    def __init__(self, x_thresholds, y_thresholds):
        self.x = np.ascontiguousarray(x_thresholds, np.float64).reshape(-1)
        self.y = np.ascontiguousarray(y_thresholds, np.float64).reshape(-1)
        if self.x.size != self.y.size or self.x.size < 1 or np.any(np.diff(self.x) <= 0):
            raise ValueError("isotonic table needs >= 1 strictly increasing thresholds")

    @classmethod
    def from_sklearn(cls, model):
        return cls(model.X_thresholds_, model.y_thresholds_)


class BoxCalibrator:
    """calibrate_boxuncert(method) for the uncertainty columns of the driver's last global post-process.

    models: dict with any of  ts_all: float;  ts_percoo: 4 floats;  iso_all: IsoTable;  iso_percoo: 4 IsoTables
    (ymin, xmin, ymax, xmax);  iso_perclscoo / rel_iso_perclscoo: num_classes x 4 IsoTables (class-major)."""

    def __init__(self, driver, models):
        self.driver, self.models = driver, dict(models)

    def calibrate_boxuncert(self, n, which="albox", method=None):
        d = self.driver
        method = method or d.params.get("calib_method_box")
        if method not in METHODS:
            raise ValueError("Unknown calibration method {}".format(method))
        if method not in self.models:
            raise ValueError("no calibration model for {}".format(method))
        la, mc = bool(d.params["loss_attenuation"]), bool(d.plan.box_stacked_dev)
        if which == "albox" and la:
            col0 = 4
        elif which == "mcbox" and mc:
            col0 = 8 if la else 4
        else:
            raise ValueError("the model produces no {} uncertainty".format(which))
        out = np.empty((n, d.M, 4), np.float32)
        m = self.models[method]
        if method.startswith("ts"):
            temps = np.ascontiguousarray(np.atleast_1d(m), np.float32)
            mode = capi.CALIB_TS_ALL if method == "ts_all" else capi.CALIB_TS_PERCOO
            if temps.size != (1 if method == "ts_all" else 4):
                raise ValueError("{} needs {} temperatures".format(method, 1 if method == "ts_all" else 4))
            d._ck(d._lib.uda_calibrate_box(d._h, col0, mode, 0, 0, None, None, None, temps.ctypes.data, out.ctypes.data),
                  "uda_calibrate_box")
            return out
        tables = [m] if isinstance(m, IsoTable) else list(np.asarray(m, dtype=object).reshape(-1))
        mode = {"iso_all": capi.CALIB_ISO_ALL, "iso_percoo": capi.CALIB_ISO_PERCOO}.get(method, capi.CALIB_ISO_PERCLSCOO)
        off = np.zeros(len(tables) + 1, np.int32)
        off[1:] = np.cumsum([t.x.size for t in tables])
        xs = np.concatenate([t.x for t in tables])
        ys = np.concatenate([t.y for t in tables])
        d._ck(d._lib.uda_calibrate_box(d._h, col0, mode, int(method.startswith("rel_")), len(tables), off.ctypes.data,
                                       xs.ctypes.data, ys.ctypes.data, None, out.ctypes.data), "uda_calibrate_box")
        return out


CLASS_METHODS = ("ts_all", "ts_percls", "iso_all", "iso_percls")


class ClassCalibrator:
    """`CalibrateClass` of the reference (src/utils_class.py:44-272) for the logits of the driver's last global
    post-process: temperature scaling (one temperature / one per class) or isotonic regression of the softmax
    probabilities (one table / one per class), entropy recomputed; with MC class uncertainty the logits are sampled
    (`draws`, the reference uses 10) and the std of the calibrated probabilities is returned as well.  The arithmetic
    runs on the device (`uda_calibrate_class`).

    models: dict with any of  ts_all: float;  ts_percls: num_classes floats;  iso_all: IsoTable;  iso_percls:
    num_classes IsoTables - the contents of the reference's pickled calibrators
    (results/calibration/<model>/classification/[unc_]classification_<method>)."""

    def __init__(self, driver, models, calib_method="ts_all", draws=10, seed=0):
        self.driver, self.models, self.calib_method = driver, dict(models), calib_method
        self.draws, self.seed = int(draws), int(seed)
        self.available_calib = list(CLASS_METHODS)

    def _has_uncert(self):
        d = self.driver
        return bool(d.plan.cls_stacked_dev) and not d.params["nms_configs"].get("max_nms_inputs", 0)

    def perform_class_calib(self, n, calib_method):
        """(entropy [n,M], probab [n,M,C][, uncert [n,M,C]]) - `_perform_class_calib` (:109-187)."""
        d = self.driver
        if calib_method not in CLASS_METHODS:
            raise ValueError("Unknown calibration method {}".format(calib_method))
        if calib_method not in self.models:
            raise ValueError("no calibration model for {}".format(calib_method))
        C_ = d.num_classes
        probs = np.empty((n, d.M, C_), np.float32)
        ent = np.empty((n, d.M), np.float32)
        with_unc = self._has_uncert()
        unc = np.empty((n, d.M, C_), np.float32) if with_unc else None
        draws = self.draws if with_unc else 0
        m = self.models[calib_method]
        up = lambda a: None if a is None else a.ctypes.data
        if calib_method.startswith("ts"):
            t = np.atleast_1d(np.asarray(m, np.float32))
            if calib_method == "ts_all":
                t = np.full((C_,), t.reshape(-1)[0], np.float32)
            if t.size != C_:
                raise ValueError("ts_percls needs {} temperatures".format(C_))
            t = np.ascontiguousarray(t)
            d._ck(d._lib.uda_calibrate_class(d._h, capi.CLS_TS, 0, None, None, None, t.ctypes.data, draws, C.c_uint64(self.seed),
                                             probs.ctypes.data, ent.ctypes.data, up(unc)), "uda_calibrate_class")
        else:
            tables = [m] if isinstance(m, IsoTable) else list(m)
            mode = capi.CLS_ISO_ALL if calib_method == "iso_all" else capi.CLS_ISO_PERCLS
            off = np.zeros(len(tables) + 1, np.int32)
            off[1:] = np.cumsum([t.x.size for t in tables])
            xs = np.concatenate([t.x for t in tables])
            ys = np.concatenate([t.y for t in tables])
            d._ck(d._lib.uda_calibrate_class(d._h, mode, len(tables), off.ctypes.data, xs.ctypes.data, ys.ctypes.data, None, draws,
                                             C.c_uint64(self.seed), probs.ctypes.data, ent.ctypes.data, up(unc)), "uda_calibrate_class")
        return (ent, probs, unc) if with_unc else (ent, probs)

    def calibrate_class(self, n):
        """The reference's return tuple (:188-272): the selected method's (uncertainty,) entropy first, then
        (probab, [uncert,] entropy) of ts_all, ts_percls, iso_all, iso_percls; empty arrays for methods without a model."""
        with_unc = self._has_uncert()
        empty = [np.array([]), np.array([]), np.array([])]
        res = {m: (list(self.perform_class_calib(n, m)) if m in self.models else list(empty)) for m in CLASS_METHODS}
        sel = res.get(self.calib_method, empty)
        have = self.calib_method in self.models
        out = []
        if with_unc:
            out.append(sel[2] if have else np.array([]))
        out.append(sel[0] if have else np.array([]))
        for m in CLASS_METHODS:
            r = res[m]
            out += [r[1], r[2], r[0]] if with_unc else [r[1], r[0]]
        return tuple(out)
