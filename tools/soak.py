"""Soak (run on the GPU box): serve_stream - the pipelined feed (prefetch on the copy stream, two input slots) AND, since round 4,
pipelined steps (uda_run_async / uda_collect: a step's post-process beside the next step's network) - against plain serve() over
many steps: any race between an upload and the kernels reading a slot, between a post-process and the next network's head writes,
or between the two output sets shows up as a mismatch."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from common import FULL_MC, make_params, make_weights
from uda_amd.infer_lib import KerasDriver

def run(image_size, hw, batch, n_batches, reps, T):
    p = make_params(image_size=image_size, **dict(FULL_MC, mc_dropoutsamp=T))
    d = KerasDriver("_", False, p["name"], batch, False, p, weights=make_weights(p, cls_spread=20.0))
    rng = np.random.default_rng(0)
    pool = []
    for i in range(n_batches):
        n = int(rng.integers(1, batch + 1))
        if i % 3 == 2:      # ragged batch
            pool.append([rng.integers(0, 256, (int(hw[0] * rng.uniform(0.5, 1.2)), int(hw[1] * rng.uniform(0.5, 1.2)), 3), dtype=np.uint8) for _ in range(n)])
        else:
            pool.append(rng.integers(0, 256, (n, hw[0], hw[1], 3), dtype=np.uint8))
    d.set_dropout_seed(3)
    want = [d.serve(b) for b in pool]
    order = [int(v) for v in rng.integers(0, n_batches, reps)]
    t0 = time.perf_counter()
    bad = 0
    for k, det in enumerate(d.serve_stream(pool[i] for i in order)):
        for g, r in zip(det, want[order[k]]):
            if not np.array_equal(g, r):
                bad += 1
                break
    dt = time.perf_counter() - t0
    print("%s batch<=%d T=%d: %d streamed steps, %d mismatches, %.1f ms/step, coop fallbacks %d not launched %d" % (
        image_size, batch, T, reps, bad, dt / reps * 1e3, d.nms_coop_fallbacks(), d.nms_coop_not_launched()))
    d.close()
    return bad

bad = run("192x128", (128, 192), 6, 12, 600, 3)
bad += run("1280x768", (768, 1280), 8, 6, 60, 10)
bad += run("1280x768", (768, 1280), 32, 4, 40, 10)          # the headline shape: here the overlap is real (4 ms of post-process)
sys.exit(1 if bad else 0)
