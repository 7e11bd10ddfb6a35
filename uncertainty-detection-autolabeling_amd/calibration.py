"""Calibrated box uncertainty on top of the driver.

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
