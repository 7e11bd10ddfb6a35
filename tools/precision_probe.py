import os, subprocess, sys, numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT + "/tests")
import test_gpu_fullsize as T
runs = {}
for terms in ("3", "0", "6"):
    out = "/tmp/prec_%s.npz" % terms
    e = dict(os.environ, UDA_PW_TERMS=terms)
    r = subprocess.run([sys.executable, "-c", T.PRECISION_WORKER % {"root": ROOT}, out], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    runs[terms] = dict(np.load(out))
ex = runs["0"]
for t in ("3", "6"):
    g = runs[t]
    box_scale = np.maximum(ex["cb"][..., 2] - ex["cb"][..., 0], ex["cb"][..., 3] - ex["cb"][..., 1])[..., None]
    print("terms", t,
          "score max rel", float(np.abs(g["cs"] - ex["cs"]).max() / ex["cs"].max()),
          "box max / scale", float((np.abs(g["cb"] - ex["cb"]) / np.maximum(box_scale, 1.0)).max()),
          "ual rms rel", float(np.sqrt(np.mean((g["ual"] - ex["ual"]) ** 2)) / np.sqrt(np.mean(ex["ual"] ** 2))),
          "uep rms rel", float(np.sqrt(np.mean((g["uep"] - ex["uep"]) ** 2)) / np.sqrt(np.mean(ex["uep"] ** 2))),
          "uep max rel", float((np.abs(g["uep"] - ex["uep"]) / np.maximum(ex["uep"], 1e-2 * np.maximum(box_scale, 1.0))).max()),
          "cls same", float((g["cc"] == ex["cc"]).mean()),
          "same kept top20", [int((np.isin(g["b"][n, :20, 0], ex["b"][n, :, 0])).sum()) for n in range(2)])
    for key, groups in (("h_cls", [(0, 63)]), ("h_box", [(0, 36), (36, 72)])):
        for lo, hi in groups:
            a, b = g[key][..., lo:hi].astype(np.float64), ex[key][..., lo:hi].astype(np.float64)
            print("   ", key, lo, "rel rms", float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b * b))), "max/maxref", float(np.abs(a - b).max() / np.abs(b).max()))
