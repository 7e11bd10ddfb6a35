"""GPU box: which serve deviates after a range demotion (first = replayed, second / third = ordinary runs), and where the first
non-finite value of the replayed run appears (every named buffer in op order; arena recycling off)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from common import LOSS_ATT, make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
keep = len(sys.argv) > 1 and sys.argv[1] == "keep"
p = make_params(uda_keep_buffers=keep, **LOSS_ATT)
w = dict(make_weights(p, seed=81))
k = [n for n in w if n.endswith("blocks_3/tpu_batch_normalization_1/gamma")]
q = [n for n in w if n.endswith("blocks_3/conv2d_1/kernel")]
w[k[0]] = w[k[0]] * np.float32(3.0e5)
w[q[0]] = w[q[0]] / np.float32(3.0e5)
a = make_images(2, 100, 180, seed=82)
runs = []
d = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
for i in range(3):
    det = d.serve(a)
    cls, box = d.head_outputs(2)
    runs.append((det, cls, box))
    print("serve", i, "demotions", d.range_demotions(), "finite", all(np.isfinite(x).all() for x in list(det) + cls + box))
    if i == 0:
        for j, x in enumerate(list(det)):
            if not np.isfinite(x).all():
                bad = ~np.isfinite(x)
                print("  det%d shape %s non-finite %d; by last-axis column: %s" % (j, x.shape, bad.sum(), bad.reshape(-1, x.shape[-1]).sum(0).tolist() if x.ndim > 1 else ""))
        for name, arrs in (("cls", cls), ("box", box)):
            for lv, x in enumerate(arrs):
                bad = ~np.isfinite(x)
                if bad.any():
                    print("  %s level %d shape %s non-finite %d; by channel: %s" % (name, lv, x.shape, bad.sum(), bad.reshape(-1, x.shape[-1]).sum(0).tolist()))
        if keep:
            order = sorted(d.plan.buffer_names.items(), key=lambda kv: kv[1])
            for name, bi in order:
                b = d.plan.bufs[bi]
                if b.kind != 0:
                    continue
                x = d.read_buffer(name, 2)
                bad = ~np.isfinite(x)
                if bad.any():
                    print("  first non-finite buffer: %s %s count %d of %d; rows %s; by channel (first 24): %s" % (
                        name, x.shape, bad.sum(), x.size, bad.reshape(x.shape[0], -1).sum(1).tolist(), bad.reshape(-1, x.shape[-1]).sum(0).tolist()[:24]))
                    break
            else:
                print("  every arena buffer is finite")
d.close()
for i in (1, 2):
    for name, x, y in [("det%d" % j, runs[0][0][j], runs[i][0][j]) for j in range(len(runs[0][0]))] + \
                      [("cls%d" % j, runs[0][1][j], runs[i][1][j]) for j in range(5)] + [("box%d" % j, runs[0][2][j], runs[i][2][j]) for j in range(5)]:
        if not np.array_equal(x, y):
            dd = np.abs(np.nan_to_num(x.astype(np.float64), nan=1e30, posinf=1e30, neginf=1e30) - np.nan_to_num(y.astype(np.float64), nan=1e30, posinf=1e30, neginf=1e30))
            print("serve 0 vs %d: %s differs: max %g, %d of %d elements" % (i, name, dd.max(), (dd > 0).sum(), dd.size))
print("serve 1 vs 2 equal:", all(np.array_equal(x, y) for x, y in zip(runs[1][0], runs[2][0])))
