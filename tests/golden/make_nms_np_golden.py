"""Generates tests/golden/nms_np_golden.npz by running the REAL reference module
src/nms_np.py (numpy-only, importable in the build container).  Run once in the
container where /root/reference exists; the .npz (data only) is committed and is
what the tests read — /root/reference never travels to the GPU box.

    python tests/golden/make_nms_np_golden.py
"""
import os
import sys

sys.dont_write_bytecode = True          # never write into /root/reference
sys.path.insert(0, "/root/reference/src")
import numpy as np                       # noqa: E402
import nms_np                            # noqa: E402  (the reference module)


def make_dets(rng, n, span=300.0, clustered=True):
    """[n,5] x1,y1,x2,y2,score with DISTINCT scores (argsort()[::-1] is unstable on ties)."""
    if clustered:
        centres = rng.uniform(40, span - 40, size=(max(1, n // 6), 2))
        c = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 6, (n, 2))
    else:
        c = rng.uniform(0, span, (n, 2))
    wh = rng.uniform(8, 70, (n, 2))
    x1, y1 = c[:, 0] - wh[:, 0] / 2, c[:, 1] - wh[:, 1] / 2
    scores = rng.permutation(np.linspace(0.02, 0.98, n)) + rng.uniform(0, 1e-3, n)
    return np.column_stack([x1, y1, x1 + wh[:, 0], y1 + wh[:, 1], scores])


def main():
    rng = np.random.default_rng(20240607)
    out = {}
    cases = []
    for ci, (n, clustered) in enumerate([(1, False), (2, True), (17, True), (64, True),
                                          (200, True), (200, False), (500, True)]):
        dets = make_dets(rng, n, clustered=clustered)
        out["c%d_dets" % ci] = dets
        out["c%d_hard" % ci] = nms_np.hard_nms(dets.copy(), 0.5)
        out["c%d_hard03" % ci] = nms_np.hard_nms(dets.copy(), 0.3)
        out["c%d_diou" % ci] = nms_np.diou_nms(dets.copy(), 0.5)
        for m, cfg in (("gauss", dict(method="gaussian", sigma=None, iou_thresh=None, score_thresh=None)),
                       ("gauss2", dict(method="gaussian", sigma=0.25, iou_thresh=None, score_thresh=0.05)),
                       ("linear", dict(method="linear", sigma=None, iou_thresh=0.3, score_thresh=0.01))):
            out["c%d_%s" % (ci, m)] = nms_np.soft_nms(dets.copy(), cfg)
        cases.append(ci)
    # per_class_nms (y1,x1,y2,x2 boxes; classes; scale)
    for pi, (n, ncls, method) in enumerate([(120, 7, "gaussian"), (300, 10, "hard"), (40, 3, "gaussian"), (5, 7, "hard")]):
        d = make_dets(rng, n)
        boxes = d[:, [1, 0, 3, 2]].astype(np.float32)
        scores = d[:, 4].astype(np.float32)
        classes = rng.integers(0, ncls, n).astype(np.int32)
        cfg = dict(method=method, sigma=None, iou_thresh=None, score_thresh=None)
        det = nms_np.per_class_nms(boxes.copy(), scores.copy(), classes.copy(),
                                   np.array([7 + pi]), np.array([1.25], dtype=np.float32), ncls, 100, cfg)
        out["p%d_boxes" % pi], out["p%d_scores" % pi], out["p%d_classes" % pi] = boxes, scores, classes
        out["p%d_meta" % pi] = np.array([7 + pi, ncls, 0 if method == "gaussian" else 1])
        out["p%d_det" % pi] = det
    out["n_cases"] = np.array([len(cases), 4])
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nms_np_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
