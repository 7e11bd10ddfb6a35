"""GPU box: pws_kernel (few pixels x many input channels) against pwb_kernel over the number of pixels of a launch - where the
threshold UDA_PW_SKINNY belongs.  usage: python tools/debug/pw_skinny_sweep.py   (runs itself twice: UDA_PW_SKINNY=1000000 / 0)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(rows, hw, cin, cout) for (hw, cin, cout) in ((960, 1152, 192), (960, 1152, 320), (960, 672, 192), (3840, 672, 112), (3840, 480, 80), (3840, 480, 112), (3840, 240, 80), (15360, 240, 40))
          for rows in (1, 2, 4, 10, 20)]
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
    import numpy as np
    import test_gpu_ops as T
    out = {}
    for (rows, hw, cin, cout) in SHAPES:
        rng = np.random.default_rng(1)
        x = rng.normal(0, 1, (rows, hw, cin)).astype(np.float32)
        w = (rng.normal(0, 1, (cin, cout)) / np.sqrt(cin)).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, cout).astype(np.float32); sh = rng.normal(0, 0.3, cout).astype(np.float32)
        se = rng.uniform(0.1, 1.0, (rows, cin)).astype(np.float32)
        _, ms = T._run(x, w, None, sc, sh, se, None, None, 1, 0, 16, reps=30)
        out["%d,%d,%d,%d" % (rows, hw, cin, cout)] = ms * 1e3
    print(json.dumps(out))
else:
    res = {}
    for mode in ("1000000", "0"):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, UDA_PW_SKINNY=mode), capture_output=True, text=True, cwd=ROOT)
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    for k in res["0"]:
        rows, hw, cin, cout = [int(v) for v in k.split(",")]
        print("%6d px (%2d x %5d) %4d -> %3d   skinny %6.1f us   tiled %6.1f us" % (rows * hw, rows, hw, cin, cout, res["1000000"][k], res["0"][k]))
