"""The N>1 path on CPU: world_size-2 `gloo` process group, contiguous image shards, one
all-gather of the packed detection records (the same code runs over RCCL with backend nccl)."""
import os
import socket
import subprocess
import sys

import numpy as np

from common import ROOT
from uda_amd import dist as udist


def test_shard_ranges_cover_the_batch():
    for n in (0, 1, 5, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [udist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_det(n, seed, M=100, C=7, bc=12, cc=8):
    r = np.random.default_rng(seed)
    return (r.normal(size=(n, M, bc)).astype(np.float32), r.uniform(size=(n, M)).astype(np.float32),
            r.normal(size=(n, M, cc)).astype(np.float32), r.integers(0, M + 1, n).astype(np.int32),
            r.normal(size=(n, M, C)).astype(np.float32))


def test_pack_unpack_roundtrip():
    det = _fake_det(3, 0)
    packed, layout = udist.pack_detections(det)
    assert packed.shape == (3, 100, 12 + 1 + 8 + 7 + 1)
    for a, b in zip(udist.unpack_detections(packed, layout), det):
        np.testing.assert_array_equal(a, b)
    det2 = (det[0][..., :4], det[1], det[2][..., 0], det[3])
    p2, l2 = udist.pack_detections(det2)
    out2 = udist.unpack_detections(p2, l2)
    assert len(out2) == 4 and out2[2].ndim == 2
    for a, b in zip(out2, det2):
        np.testing.assert_array_equal(a, b)


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch.distributed as dist
from uda_amd import dist as udist
from test_dist_gloo import _fake_det
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n_total = int(sys.argv[1])
full = _fake_det(n_total, 123)
a, b = udist.shard_range(n_total, rank, world)
local = tuple(x[a:b] for x in full)
got = udist.all_gather_detections(local)
ok = all(np.array_equal(g, f) for g, f in zip(got, full))
dist.barrier()
dist.destroy_process_group()
print("RANK", rank, "OK" if ok else "MISMATCH", got[0].shape)
sys.exit(0 if ok else 1)
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_world(n_total, world=2):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT}, str(n_total)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "OK" in o


def test_all_gather_detections_world2_even_shards():
    _run_world(6)


def test_all_gather_detections_world2_ragged_and_empty_shards():
    _run_world(5)      # shards of 3 and 2 images
    _run_world(1)      # rank 1 owns no image


RESHARD_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from uda_amd import dist as udist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n_total, n_members = int(sys.argv[1]), int(sys.argv[2])
shapes_c = [(4, 5, 6), (2, 3, 6)]
shapes_b = [(4, 5, 8), (2, 3, 8)]
def heads(m):
    r = np.random.default_rng(100 + m)
    return ([r.normal(size=(n_total,) + s).astype(np.float32) for s in shapes_c],
            [r.normal(size=(n_total,) + s).astype(np.float32) for s in shapes_b])
owned = {m: heads(m) for m in range(n_members) if udist.member_owner(m, world) == rank}
cls, box = udist.reshard_member_heads(owned, n_members, n_total, rank, world)
a, b = udist.shard_range(n_total, rank, world)
ok = True
for m in range(n_members):
    c, bx = heads(m)
    for l in range(2):
        ok &= np.array_equal(cls[l][m], c[l][a:b]) and np.array_equal(box[l][m], bx[l][a:b])
ok &= cls[0].shape == (n_members, b - a, 4, 5, 6)
dist.barrier()
dist.destroy_process_group()
print("RANK", rank, "OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
'''


def _run_reshard(n_total, n_members, world=2):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", RESHARD_WORKER % {"root": ROOT}, str(n_total), str(n_members)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "OK" in o


def test_ensemble_reshard_world2():
    """Members striped over the ranks, heads re-sharded by image (all-to-all-v of point-to-point transfers)."""
    _run_reshard(5, 3)      # rank 0 owns members 0 and 2, rank 1 member 1; shards of 3 and 2 images
    _run_reshard(4, 1)      # rank 1 owns no member
    _run_reshard(1, 2)      # rank 1 owns no image


def test_mc_samples_striped_over_ranks_reshard_like_ensemble_members():
    """MC-sample sharding (north_star "optionally MC samples"; dist.SampleShardedDriver): rank r runs samples t = r (mod world)
    of every image, a sample is a "member" of the ensemble exchange - T = 10 over two ranks for ONE image (the reference's
    batch-1 protocol: rank 1 owns no image and still contributes five samples), T = 5 over two ranks (3 + 2 samples), and a
    head that carries no sample axis (zero-width payload)."""
    from uda_amd import dist as udist
    assert udist.local_samples(10, 1, 4) == [1, 5, 9] and udist.local_samples(5, 0, 2) == [0, 2, 4]
    assert all(udist.sample_owner(t, 3) == udist.member_owner(t, 3) for t in range(9))
    _run_reshard(1, 10)
    _run_reshard(3, 5)
    _run_reshard(2, 4, world=3)
