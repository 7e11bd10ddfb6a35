"""Round-4 GPU tests, all through the C ABI:

  * two chunk lanes (UDA_LANES=2, consecutive chunks concurrently on two streams): bit-identical to one lane - each lane owns its
    block-1 operand buffer (round-3 advisor finding: one shared buffer was overwritten by the other lane's prep);
  * uda_detections_device: the device-resident record buffer equals dist.pack_detections of the downloaded detections, the
    padding rows are zero, per-class mode included;
  * the device-resident gather over RCCL in a world of one returns what serve() returns;
  * fp16 pieces: an activation above 65504 inside the network fails the run loudly (no infinities returned), and the same
    weights run under three bf16 pieces.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from common import FULL_MC, LOSS_ATT, ROOT, make_images, make_params, make_weights

pytestmark = pytest.mark.gpu


def _driver(params, w, batch, **kw):
    from uda_amd.infer_lib import KerasDriver
    return KerasDriver("_", False, params["name"], batch_size=batch, model_params=params, weights=w, **kw)


LANES_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import FULL_MC, make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
p = make_params(**FULL_MC)
w = make_weights(p, seed=51, cls_spread=20.0)
d = KerasDriver("_", False, p["name"], 5, False, p, weights=w, chunk_images=1)     # five chunks per run
d.set_dropout_seed(13)
imgs = make_images(5, 100, 180, seed=52)
det = d.serve(imgs)
det2 = d.serve(imgs[::-1].copy())
cls, box = d.head_outputs(5)
np.savez(sys.argv[1], *det, *det2, cls[0], box[0], cls[4], box[4])
d.close()
print("saved")
"""


def test_two_chunk_lanes_are_bit_identical_to_one(tmp_path):
    """UDA_LANES=2 runs consecutive chunks on two (stream, arena) pairs.  The fused block-1 kernel reads a per-gate-row
    operand block that a small kernel prepares right before it: one shared buffer let lane 1's prep overwrite what lane
    0's kernel was still reading (silent corruption; the default of one lane is stream-ordered).  Each lane owns its
    buffer now: five one-image chunks per run, two runs, detections and head outputs bit for bit."""
    outs = {}
    for lanes in ("1", "2"):
        e = dict(os.environ, UDA_LANES=lanes)
        out = str(tmp_path / ("lanes%s.npz" % lanes))
        r = subprocess.run([sys.executable, "-c", LANES_WORKER % {"root": ROOT}, out], cwd=ROOT, env=e, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "saved" in r.stdout, (lanes, r.stdout[-1500:], r.stderr[-1500:])
        outs[lanes] = dict(np.load(out))
    assert outs["1"].keys() == outs["2"].keys() and len(outs["1"]) >= 12
    for k in outs["1"]:
        np.testing.assert_array_equal(outs["2"][k], outs["1"][k], err_msg=k)


def _read_device(ptr, shape):
    from uda_amd import capi
    hip = capi.load()           # hipMemcpy of the runtime the library itself is linked against (dlsym through its dependencies)
    out = np.empty(shape, np.float32)
    rc = hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(out.nbytes), C.c_int(2))      # device to host
    assert rc == 0, rc
    return out


@pytest.mark.parametrize("mode", ["global", "per_class"])
def test_device_resident_detection_records_equal_the_host_pack(mode):
    """uda_detections_device (what RCCL gathers from) against dist.pack_detections of the downloaded detections: the same
    float32 record per (image, detection), zero rows as padding up to the requested row count."""
    from uda_amd import dist as udist
    p = make_params(**FULL_MC)
    w = make_weights(p, seed=61, cls_spread=20.0)
    d = _driver(p, w, 4)
    d.set_dropout_seed(3)
    imgs = make_images(3, 100, 180, seed=62)
    det = d.serve(imgs, post_mode=mode)
    want, layout = udist.pack_detections(det)
    ptr, rows, lay = d.detections_device(rows=4, mode=d._mode(mode))
    assert lay == layout and rows == 4
    got = _read_device(ptr, (4, d.M, want.shape[-1]))
    np.testing.assert_array_equal(got[:3], want)
    assert not got[3].any()
    back = udist.unpack_detections(got[:3], lay)
    for g, r in zip(back, det):
        np.testing.assert_array_equal(g, r)
    with pytest.raises(RuntimeError):
        d.detections_device(rows=2, mode=d._mode(mode))          # fewer rows than images of the last run
    d.close()


RCCL_WORKER = r'''
import os, sys
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch                     # torch first: the HIP library then binds to the same runtime
import torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
from common import FULL_MC, make_images, make_params, make_weights
from uda_amd import dist as udist
from uda_amd.infer_lib import KerasDriver
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", rank))
torch.cuda.set_device(local)
tdev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=tdev)
p = make_params(**FULL_MC)
w = make_weights(p, seed=71, cls_spread=20.0)
imgs = make_images(5, 100, 180, seed=72)
d = KerasDriver("_", False, p["name"], 5, False, p, weights=w, device=local)
d.set_dropout_seed(17)
got = udist.serve_sharded(d, imgs, rank, world, device=tdev)          # device-resident gather (backend nccl)
dev_t, layout = udist.all_gather_detections_device(d, udist.shard_range(5, rank, world)[1] - udist.shard_range(5, rank, world)[0],
                                                   [udist.shard_range(5, r, world)[1] - udist.shard_range(5, r, world)[0] for r in range(world)],
                                                   tdev, to_host=False)
assert dev_t.is_cuda and dev_t.shape[1] == d.M
# pipelined: two batches in flight on every rank, each gathered from its own ticket (uda_collect_device)
imgs2 = make_images(5, 100, 180, seed=73)
lo, hi = udist.shard_range(5, rank, world)
counts = [udist.shard_range(5, r, world)[1] - udist.shard_range(5, r, world)[0] for r in range(world)]
d.set_image_offset(lo)
d.stage_images(imgs[lo:hi]); t0 = d.run_async()
d.stage_images(imgs2[lo:hi]); t1 = d.run_async()
got_p0 = udist.all_gather_detections_device(d, hi - lo, counts, tdev, ticket=t0)
got_p1 = udist.all_gather_detections_device(d, hi - lo, counts, tdev, ticket=t1)
if rank == 0:
    one = KerasDriver("_", False, p["name"], 5, False, p, weights=w, device=local)
    one.set_dropout_seed(17)
    want = one.serve(imgs)
    want2 = one.serve(imgs2)
    one.close()
    assert len(got) == len(want)
    for g, r in zip(got, want):
        assert g.dtype == r.dtype and np.array_equal(g, r), "device-resident gather differs from serve()"
    for g, r in zip(got_p0, want):
        assert np.array_equal(g, r), "pipelined gather (first ticket) differs from serve()"
    for g, r in zip(got_p1, want2):
        assert np.array_equal(g, r), "pipelined gather (second ticket) differs from serve()"
    print("rccl gather ok", world)
d.close()
dist.barrier(); dist.destroy_process_group()
'''


def _run_rccl(world, tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER % {"root": ROOT})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl gather ok %d" % world in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_device_resident_gather_over_rccl_world1(tmp_path):
    """e: under an RCCL process group the detections are packed on the device, all-gathered out of the handle's own buffer
    and downloaded once (dist.all_gather_detections_device); in a world of one the result must be what serve() returns."""
    _run_rccl(1, tmp_path)


def test_device_resident_gather_over_rccl_two_gpus(tmp_path):
    """The same with two ranks on two GPUs (ragged shards 3 + 2: zero-padded records).  Skipped on a one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run_rccl(2, tmp_path)


OVERFLOW_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import FULL_MC, LOSS_ATT, make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
p = make_params(**(FULL_MC if sys.argv[2] == "mc" else LOSS_ATT))
w = dict(make_weights(p, seed=81))
# blow the batch-norm scale behind block 3's depthwise conv up by 3e5 and shrink the projection kernel that follows by the
# same factor: the network computes what it computed before (the squeeze-excite gate saturates, nothing else changes
# scale), but the projection's INPUT - the operand its 1x1 contraction must split - now reaches ~1e6
k = [n for n in w if n.endswith("blocks_3/tpu_batch_normalization_1/gamma")]
q = [n for n in w if n.endswith("blocks_3/conv2d_1/kernel")]
assert len(k) == 1 and len(q) == 1, (k, q)
w[k[0]] = w[k[0]] * np.float32(3.0e5)
w[q[0]] = w[q[0]] / np.float32(3.0e5)
d = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
d.set_dropout_seed(5)
out = {}
a = make_images(2, 100, 180, seed=82)
b = make_images(2, 128, 192, seed=83)
det = d.serve(a)
out["n_after_first"] = np.int64(d.range_demotions())
cls, box = d.head_outputs(2)
det2 = d.serve(a)
out["n_after_second"] = np.int64(d.range_demotions())
for i, (x, y) in enumerate(zip(det, det2)):
    assert np.array_equal(x, y), "the second serve of the same batch differs"
    out["det_%%d" %% i] = x
for i, x in enumerate(cls + box):
    out["head_%%d" %% i] = x
# pipelined serves of further batches on the same handle (the op stays re-packed)
for j, dets in enumerate(d.serve_stream([b, a, b])):
    for i, x in enumerate(dets):
        out["stream%%d_%%d" %% (j, i)] = x
out["n_final"] = np.int64(d.range_demotions())
out["finite"] = np.bool_(all(np.isfinite(v).all() for k_, v in out.items() if k_.startswith(("det_", "head_", "stream"))))
np.savez(sys.argv[1], **out)
d.close()
print("saved")
"""

OVERFLOW_STREAM_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import FULL_MC, make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
p = make_params(**FULL_MC)
w = dict(make_weights(p, seed=81))
k = [n for n in w if n.endswith("blocks_3/tpu_batch_normalization_1/gamma")]
q = [n for n in w if n.endswith("blocks_3/conv2d_1/kernel")]
w[k[0]] = w[k[0]] * np.float32(3.0e5)
w[q[0]] = w[q[0]] / np.float32(3.0e5)
batches = [make_images(2, 100, 180, seed=82), make_images(2, 128, 192, seed=83), make_images(1, 90, 200, seed=84), make_images(2, 128, 192, seed=85)]
out = {}
for tag in ("stream", "serial"):
    d = KerasDriver("_", False, p["name"], 2, False, p, weights=w)      # a FRESH handle: the flag is first raised inside the stream
    d.set_dropout_seed(5)
    if tag == "stream":
        res = list(d.serve_stream(batches))
    else:
        res = [d.serve(x) for x in batches]
    out["n_" + tag] = np.int64(d.range_demotions())
    for j, dets in enumerate(res):
        for i, x in enumerate(dets):
            out["%%s%%d_%%d" %% (tag, j, i)] = x
    d.close()
np.savez(sys.argv[1], **out)
print("saved")
"""


def _overflow_run(worker, tmp_path, tag, scheme, *args):
    e = dict(os.environ, UDA_PW_SCHEME=scheme)
    e.pop("UDA_PW_TERMS", None)
    out = str(tmp_path / ("%s_%s.npz" % (tag, scheme)))
    r = subprocess.run([sys.executable, "-c", worker % {"root": ROOT}, out, *args], cwd=ROOT, env=e, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "saved" in r.stdout, (scheme, r.stdout[-1500:], r.stderr[-2500:])
    return dict(np.load(out)), r.stderr


@pytest.mark.parametrize("mode", ["det", "mc"])
def test_fp16_range_overflow_is_served_on_three_bf16_pieces(tmp_path, mode):
    """Two fp16 pieces cannot hold an operand above 65504.  Every kernel that splits operands raises its OP's flag word;
    every reader of a run's results looks first.  The reference computes in float32 and always returns
    (infer_lib.py:337-343) - so the handle re-packs the first flagged op with three bf16 pieces, serves the run again from
    its unchanged inputs and only then returns: finite detections, ONE demotion (the ops behind the offender only saw its
    infinities), no second demotion when the batch is served again or when further batches stream through the handle.
    What comes back equals, within the float32 bar, what a handle that runs EVERYTHING on three bf16 pieces returns."""
    got, err = _overflow_run(OVERFLOW_WORKER, tmp_path, "ovf_" + mode, "f16x2", mode)
    ref, _ = _overflow_run(OVERFLOW_WORKER, tmp_path, "ovf_" + mode, "bf16x3", mode)
    assert bool(got["finite"]) and bool(ref["finite"])
    assert int(got["n_after_first"]) == 1 and int(got["n_after_second"]) == 1 and int(got["n_final"]) == 1, (got["n_after_first"], got["n_final"])
    assert int(ref["n_final"]) == 0
    assert err.count("fp16 range: op") == 1 and "served again" in err
    heads = sorted(k for k in got if k.startswith("head_"))
    assert len(heads) == 10
    for k in heads:
        g, r = got[k].astype(np.float64), ref[k].astype(np.float64)
        assert np.sqrt(np.mean((g - r) ** 2)) <= 1e-5 * np.sqrt(np.mean(r * r)) + 1e-7, k
    for k in got:
        if k.startswith(("det_", "stream")) and k.endswith("_3"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg=k)        # valid_len
    for k in ("det_0", "det_1"):
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-3, atol=1e-3, err_msg=k)


def test_fp16_range_overflow_inside_a_stream_is_served_too(tmp_path):
    """The flag is first raised by the FIRST of several pipelined runs, with the second already queued behind it: everything
    in flight is let finish, the op is re-packed, that run is served again into its own output set, the queued run keeps its
    own flags (it overflowed as well: it is served again when it is collected).  The stream returns what one-at-a-time
    serves on a fresh handle return, bit for bit, and both handles re-packed the same single op."""
    got, err = _overflow_run(OVERFLOW_STREAM_WORKER, tmp_path, "ovf_stream", "f16x2")
    assert int(got["n_stream"]) == 1 and int(got["n_serial"]) == 1
    keys = sorted(k for k in got if k.startswith("stream"))
    assert len(keys) >= 4 * 4
    for k in keys:
        np.testing.assert_array_equal(got[k], got["serial" + k[len("stream"):]], err_msg=k)
        assert np.isfinite(got[k]).all(), k


STEM_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import FULL_MC, make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
p = make_params(**FULL_MC)
w = make_weights(p, seed=91, cls_spread=20.0)
d = KerasDriver("_", False, p["name"], 3, False, p, weights=w, chunk_images=2)
d.set_dropout_seed(19)
out = {}
for tag, hw in (("full", (128, 192)), ("small", (100, 180)), ("tiny", (3, 5))):        # raw size = / < the 192x128 network input: scale 1
    imgs = make_images(3, hw[0], hw[1], seed=92)
    det = d.serve(imgs)
    cls, box = d.head_outputs(3)
    pre, scales = d.preprocessed()
    for i, a in enumerate(list(det) + [cls[0], box[0], cls[4], box[4], pre, scales]):
        out["%%s_%%d" %% (tag, i)] = a
det = d.serve(make_images(2, 90, 200, seed=93))      # wider than the network input: resampled, the separate preprocess pass runs
for i, a in enumerate(det):
    out["resampled_%%d" %% i] = a
np.savez(sys.argv[1], **out)
d.close()
print("saved")
"""


def test_uint8_stem_is_bit_identical_to_the_preprocess_pass(tmp_path):
    """When no image of a uint8 batch is resampled (scale 1) the stem reads the raw images itself - normalisation through a
    768-entry table of the preprocess kernel's own expression - and the separate preprocess pass is skipped.  Same values, same
    accumulation order: heads and detections equal the two-pass path (UDA_STEM_U8=0) bit for bit, for raw sizes
    equal to and smaller than the network input (zero padding) down to 3 x 5 pixels; the float image is still available on
    demand (`preprocessed()`), and a batch that IS resampled takes the separate pass either way."""
    outs = {}
    for sw in ("1", "0"):
        e = dict(os.environ, UDA_STEM_U8=sw)
        out = str(tmp_path / ("stem%s.npz" % sw))
        r = subprocess.run([sys.executable, "-c", STEM_WORKER % {"root": ROOT}, out], cwd=ROOT, env=e, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "saved" in r.stdout, (sw, r.stdout[-1500:], r.stderr[-1500:])
        outs[sw] = dict(np.load(out))
    assert outs["1"].keys() == outs["0"].keys() and len(outs["1"]) >= 3 * 11
    for k in outs["1"]:
        np.testing.assert_array_equal(outs["1"][k], outs["0"][k], err_msg=k)


FUSEIN_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import FULL_MC, HEAD_MC, make_images, make_params, make_weights
from uda_amd import capi
from uda_amd.infer_lib import KerasDriver
out = {}
for tag, model, size, mc in (("d0", "efficientdet-d0", "192x128", FULL_MC), ("d0odd", "efficientdet-d0", "448x320", HEAD_MC),
                             ("d2", "efficientdet-d2", "256x128", FULL_MC)):
    p = make_params(image_size=size, model=model, **mc)
    w = make_weights(p, seed=41, cls_spread=20.0)
    d = KerasDriver("_", False, p["name"], 2, False, p, weights=w, chunk_images=2)
    d.set_dropout_seed(23)
    kinds = [o["kind"] for o in d.plan.ops]
    out[tag + "_nfuse"] = np.array([kinds.count(capi.OP_FUSE), sum(1 for o in d.plan.ops if o.get("fuse_in"))])
    W, H = [int(v) for v in size.split("x")]
    det = d.serve(make_images(2, H, W, seed=42))
    cls, box = d.head_outputs(2)
    for i, a in enumerate(list(det) + list(cls) + list(box)):
        out["%%s_%%d" %% (tag, i)] = a
    d.close()
np.savez(sys.argv[1], **out)
print("saved")
"""


def test_bifpn_fusion_inside_the_separable_conv_is_bit_identical(tmp_path):
    """UDA_FUSE_IN=1 (default): a BiFPN node's weighted fusion (identity / nearest-up / max-pooled inputs, swish) is computed
    by the node's separable conv for its own 18 x 18 input tile (sepf_kernel) - the fused tensor is never written, the 24
    (D0) / 40 (D2) fuse launches disappear.  Same expressions, same accumulation order: every head output and every detection
    equals the two-launch path (UDA_FUSE_IN=0) bit for bit - 64-, and 112-channel pyramids, maps that are and are not
    multiples of the 16 x 16 tile (40 x 56 ... 3 x 4), per-sample and shared inputs."""
    outs = {}
    for sw in ("1", "0"):
        e = dict(os.environ, UDA_FUSE_IN=sw)
        out = str(tmp_path / ("fin%s.npz" % sw))
        r = subprocess.run([sys.executable, "-c", FUSEIN_WORKER % {"root": ROOT}, out], cwd=ROOT, env=e, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0 and "saved" in r.stdout, (sw, r.stdout[-1500:], r.stderr[-1500:])
        outs[sw] = dict(np.load(out))
    assert outs["1"].keys() == outs["0"].keys()
    for tag, n in (("d0", 24), ("d0odd", 24), ("d2", 40)):
        assert list(outs["1"][tag + "_nfuse"]) == [0, n] and list(outs["0"][tag + "_nfuse"]) == [n, 0], (tag, outs["1"][tag + "_nfuse"])
    for k in outs["1"]:
        if not k.endswith("_nfuse"):
            np.testing.assert_array_equal(outs["1"][k], outs["0"][k], err_msg=k)


def _same(a, b, msg):
    assert len(a) == len(b), msg
    for i, (x, y) in enumerate(zip(a, b)):
        np.testing.assert_array_equal(x, y, err_msg="%s [%d]" % (msg, i))


@pytest.mark.parametrize("post_mode", ["global", "per_class"])
def test_pipelined_runs_return_what_synchronous_serves_return(post_mode):
    """uda_run_async / uda_collect: batch k's post-process (aggregate, NMS, gather) is not joined into the main stream, batch
    k + 1's network is queued before batch k's detections are fetched and only its head-writing ops wait.  Detections, image
    scales and the dropout stream are per run: six batches of different images, image counts and raw sizes (resampled and
    not) through `serve_stream` equal the same batches through `serve`, bit for bit, in order."""
    from common import FULL_MC, make_images, make_params, make_weights
    from uda_amd.infer_lib import KerasDriver
    p = make_params(**FULL_MC)
    w = make_weights(p, seed=5, cls_spread=20.0)
    batches = [make_images(3, 128, 192, seed=60), make_images(3, 128, 192, seed=61), make_images(2, 100, 180, seed=62),
               make_images(3, 90, 200, seed=63), make_images(1, 128, 192, seed=64), make_images(3, 128, 192, seed=65)]
    sync = KerasDriver("_", False, p["name"], 3, False, p, weights=w, chunk_images=2, post_mode=post_mode)
    sync.set_dropout_seed(7)
    want = [sync.serve(b) for b in batches]
    sync.close()
    pipe = KerasDriver("_", False, p["name"], 3, False, p, weights=w, chunk_images=2, post_mode=post_mode)
    pipe.set_dropout_seed(7)
    got = list(pipe.serve_stream(batches))
    assert len(got) == len(want)
    for k, (g, v) in enumerate(zip(got, want)):
        _same(g, v, "batch %d" % k)
    assert any(int(v[3].sum()) > 0 for v in want)
    # the same handle goes back to synchronous serving
    again = pipe.serve(batches[0])
    assert all(a.shape == b.shape for a, b in zip(again, want[0])) and np.isfinite(again[0]).all()
    pipe.close()


def test_pipelined_run_bookkeeping_is_enforced():
    """At most two runs in flight; tickets are collected once; a synchronous run, or anything that rewrites the head buffers,
    refuses while a pipelined run is in flight; the newest run may also be read through the ordinary readers."""
    from common import FULL_MC, make_images, make_params, make_weights
    from uda_amd.infer_lib import KerasDriver
    p = make_params(**FULL_MC)
    w = make_weights(p, seed=5, cls_spread=20.0)
    d = KerasDriver("_", False, p["name"], 2, False, p, weights=w, chunk_images=2)
    d.set_dropout_seed(3)
    imgs, imgs2 = make_images(2, 128, 192, seed=66), make_images(2, 120, 150, seed=67)
    ref, ref2 = d.serve(imgs), d.serve(imgs2)
    d.stage_images(imgs)
    t0 = d.run_async()
    d.stage_images(imgs2)                  # (synchronises: allowed while a run is in flight, it only costs the overlap)
    t1 = d.run_async()
    assert {t0, t1} == {0, 1}
    with pytest.raises(RuntimeError, match="two runs are in flight"):
        d.run_async()
    with pytest.raises(RuntimeError, match="in flight"):
        d.run_resident()
    with pytest.raises(RuntimeError, match="in flight"):
        d.predict_resident(np.zeros((1,) + tuple(d.image_size) + (3,), np.float32))
    first = d.collect(t0)
    _same(first, ref, "first pipelined run = the synchronous serve with the same seed")
    with pytest.raises(RuntimeError, match="not in flight"):
        d._ck(d._lib.uda_collect(d._h, t0, None, None, None, None, None), "uda_collect")
    newest = d._collect(2)                 # ordinary reader: the newest run's outputs
    second = d.collect(t1)
    _same(newest, second, "ordinary reader sees the newest run")
    _same(second, ref2, "second pipelined run = the synchronous serve of its own images")
    assert not np.array_equal(second[1], first[1])
    d.run_resident()                       # nothing in flight any more
    d.close()


PIPE_TIMEOUT_WORKER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from common import FULL_MC, make_images, make_params, make_weights
from uda_amd.infer_lib import KerasDriver
# 384x512 input = 36 828 candidates per image: the cooperative NMS takes two blocks per problem
p = make_params(image_size="512x384", **FULL_MC)
w = make_weights(p, seed=33, cls_spread=20.0)
imgs, imgs2 = make_images(2, 300, 480, seed=32), make_images(2, 280, 500, seed=34)
ref = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
ref.set_dropout_seed(9)
want, want2 = ref.serve(imgs), ref.serve(imgs2)
fb0 = ref.nms_coop_fallbacks()
ref.close()
d = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
d.set_dropout_seed(9)
# (a) the newest run: its collect redoes the post-process on the two-launch NMS - same detections, counted
d.stage_images(imgs)
t = d.run_async()
got = d.collect(t)
for g, r in zip(got, want):
    assert np.array_equal(g, r)
print("fallbacks", fb0, d.nms_coop_fallbacks())
d.close()
# (b) two in flight on a fresh handle (still on the cooperative NMS): the older run's candidates are gone - its collect fails
d = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
d.set_dropout_seed(9)
d.stage_images(imgs)
t0 = d.run_async()
d.prefetch_images(imgs2); d.swap_prefetched()       # (no synchronisation in between: t1 is queued right behind t0)
t1 = d.run_async()
try:
    d.collect(t0)
    print("older collect returned")
except RuntimeError as e:
    print("older collect raised:", str(e)[:160])
got2 = d.collect(t1)            # the newest one is redone
for g, r in zip(got2, want2):
    assert np.array_equal(g, r)
# an intervening synchronisation settles a run while it is still the newest one: then both collects succeed
d2 = KerasDriver("_", False, p["name"], 2, False, p, weights=w)
d2.set_dropout_seed(9)
d2.stage_images(imgs); u0 = d2.run_async()
d2.stage_images(imgs2); u1 = d2.run_async()          # stage_images synchronises: u0 is checked and redone there
for g, r in zip(d2.collect(u0), want):
    assert np.array_equal(g, r)
for g, r in zip(d2.collect(u1), want2):
    assert np.array_equal(g, r)
d2.close()
again = d.serve(imgs)           # the handle keeps working (two-launch NMS from now on)
for g, r in zip(again, want):
    assert np.array_equal(g, r)
d.close()
print("pipe timeout ok")
"""


def test_pipelined_run_with_a_timed_out_nms_barrier_fails_or_redoes_loudly():
    """UDA_NMS_COOP_SPIN=1 forces the cooperative NMS's time-out (test hook).  A pipelined run that is still the newest one is
    redone by its collect like a synchronous run (same detections, uda_nms_coop_fallbacks counts it); a run whose candidates a
    newer run has already replaced cannot be redone: its collect raises instead of returning garbage, the newer run is
    redone and the handle keeps serving on the two-launch NMS.  Without the hook nothing of this happens (other tests)."""
    e = dict(os.environ, UDA_NMS_COOP_SPIN="1")
    r = subprocess.run([sys.executable, "-c", PIPE_TIMEOUT_WORKER % {"root": ROOT}], cwd=ROOT, env=e, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "pipe timeout ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    fb = [l for l in r.stdout.splitlines() if l.startswith("fallbacks")][0].split()
    assert int(fb[1]) >= 1 and int(fb[2]) >= 1, fb          # reference handle and pipelined handle both fell back (and said so)
    assert "older collect raised:" in r.stdout and "cooperative NMS" in r.stdout, r.stdout[-1500:]
