"""Multi-GPU layer of the path: contiguous image shards, one process per GPU, and ONE
exchange — an all-gather of the packed detection records (RCCL over xGMI when the backend
is "nccl"; "gloo" on CPU for tests).  The reference has no inference-time collective
(SURVEY §2 'Parallelism'); this is designed fresh per SURVEY §8e: images are independent, so
there is no data-path collective, and the gather is KB-scale (latency-bound).

torch is imported lazily and only here: it provides the process group, nothing else.
"""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous [start, stop) of the images owned by `rank` (remainder to the low ranks)."""
    base, rem = divmod(int(n_total), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_detections(det):
    """(boxes [n,M,bc], scores [n,M], classes [n,M(,cc)], valid [n](, logits [n,M,C]))
    -> (float32 [n, M, cols], layout) with valid_len broadcast into the last column."""
    boxes, scores, classes, valid = det[:4]
    logits = det[4] if len(det) > 4 else None
    n, M = scores.shape
    if classes.ndim == 2:
        classes = classes[..., None]
    parts = [boxes, scores[..., None], classes]
    if logits is not None:
        parts.append(logits)
    parts.append(np.broadcast_to(valid.astype(np.float32)[:, None, None], (n, M, 1)))
    layout = dict(box=boxes.shape[-1], cls=classes.shape[-1], logits=0 if logits is None else logits.shape[-1])
    return np.ascontiguousarray(np.concatenate(parts, axis=-1), dtype=np.float32), layout


def unpack_detections(packed, layout):
    b, c, l = layout["box"], layout["cls"], layout["logits"]
    boxes = packed[..., :b]
    scores = packed[..., b]
    classes = packed[..., b + 1:b + 1 + c]
    if c == 1:
        classes = classes[..., 0]
    off = b + 1 + c
    out = [boxes, scores, classes, packed[:, 0, -1].astype(np.int32)]
    if l:
        out.append(packed[..., off:off + l])
    return tuple(out)


def all_gather_detections(det, device=None, group=None, counts=None):
    """Every rank gets the detections of all images, in global image order.  Ranks may own
    different numbers of images (ragged shards are padded to the largest shard for the
    collective and trimmed afterwards).  `counts` = images per rank when the caller knows them
    (contiguous shards of a known batch: `shard_range`), which saves the size exchange - one
    collective and one host synchronisation per step; otherwise the sizes are gathered first."""
    import torch
    import torch.distributed as dist
    packed, layout = pack_detections(det)
    world = dist.get_world_size(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    if counts is None:
        n_local = torch.tensor([packed.shape[0]], dtype=torch.int64, device=dev)
        sizes = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(sizes, n_local, group=group)
        counts = [int(c.item()) for c in sizes]
    else:
        counts = [int(c) for c in counts]
        if len(counts) != world or counts[dist.get_rank(group)] != packed.shape[0]:
            raise ValueError("counts %s do not describe this rank's %d images" % (counts, packed.shape[0]))
    n_max = max(counts)
    if packed.shape[0] == n_max:
        buf = packed
    else:
        buf = np.zeros((n_max,) + packed.shape[1:], np.float32)
        buf[:packed.shape[0]] = packed
    t = torch.from_numpy(buf).to(dev)
    # one collective into one tensor and ONE copy back (not a device-to-host copy per rank)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=dev)
    if hasattr(dist, "all_gather_into_tensor") and dist.get_backend(group) != "gloo":
        dist.all_gather_into_tensor(out, t, group=group)
    else:
        parts = [out[r] for r in range(world)]
        dist.all_gather(parts, t, group=group)
    host = out.cpu().numpy()
    full = host.reshape((world * n_max,) + packed.shape[1:]) if all(c == n_max for c in counts) else \
        np.concatenate([host[r, :c] for r, c in enumerate(counts)], axis=0)
    return unpack_detections(full, layout)


_GATHER_BUF = {}     # (device, world, n_max, M, cols) -> the receive tensor of the device-resident gather, reused across steps


def all_gather_detections_device(driver, n_local, counts, device, group=None, to_host=True, ticket=None):
    """The same gather WITHOUT a host hop: the handle packs its detections into one device-resident record
    buffer (`uda_detections_device`, padded with zero rows to the largest shard), RCCL all-gathers straight out of it into
    a reusable tensor, and ONE device-to-host copy of the gathered records follows on the ranks that want them
    (`to_host=False` returns the device tensor [world * n_max, M, cols] and the layout instead).  Before: download of the
    local detections, re-upload of the packed copy, gather, download - three PCIe crossings and two synchronisations per
    step for an 11 KB / image record that already sat in a device buffer.
    ticket: gather the detections of that pipelined run (`run_async`) instead of the last synchronous one - the next run's
    network is already queued behind it and keeps the GPU busy while the records travel."""
    import torch
    import torch.distributed as dist
    dev = torch.device(device)
    world = dist.get_world_size(group)
    counts = [int(c) for c in counts]
    if len(counts) != world or counts[dist.get_rank(group)] != int(n_local):
        raise ValueError("counts %s do not describe this rank's %d images" % (counts, n_local))
    n_max = max(counts)
    if n_local > 0:
        ptr, rows, layout = driver.detections_device(rows=n_max) if ticket is None else driver.collect_device(ticket, rows=n_max)
        cols = layout["box"] + 1 + layout["cls"] + layout["logits"] + 1
        t = torch.as_tensor(DevArray(ptr, (rows, driver.M, cols)), device=dev)
    else:       # an empty shard (more ranks than images): zero records of the right shape
        det = driver.empty_detections()
        layout = dict(box=det[0].shape[-1], cls=1 if det[2].ndim == 2 else det[2].shape[-1], logits=det[4].shape[-1] if len(det) > 4 else 0)
        cols = layout["box"] + 1 + layout["cls"] + layout["logits"] + 1
        t = torch.zeros((n_max, driver.M, cols), dtype=torch.float32, device=dev)
    # Two receive tensors per shape, used in turn: with pipelined tickets (or to_host=False) the caller may still hold - and
    # RCCL may still be filling - the previous call's tensor when the next gather is posted.
    key = (str(dev), world, n_max, driver.M, cols)
    pair = _GATHER_BUF.get(key)
    if pair is None:
        pair = _GATHER_BUF[key] = [[torch.empty((world, n_max, driver.M, cols), dtype=torch.float32, device=dev) for _ in range(2)], 0]
    out = pair[0][pair[1]]
    pair[1] ^= 1
    work = dist.all_gather_into_tensor(out, t, group=group, async_op=True)
    # the handle's packed-record buffer (`t` aliases it: one per handle) is rewritten by the next detections_device /
    # collect_device: the collective must have READ it before this returns
    work.wait()
    if not to_host:
        # The handle packs its next records on a stream of its own, which nothing orders behind RCCL's: wait for the gather
        # here (tens of microseconds) - the returned device tensor is complete and stays valid until the second next gather
        # of this shape.
        torch.cuda.current_stream(dev).synchronize()
        return out.reshape(world * n_max, driver.M, cols), layout
    host = out.cpu().numpy()             # the one download (synchronises the collective's stream)
    full = host.reshape((world * n_max, driver.M, cols)) if all(c == n_max for c in counts) else \
        np.concatenate([host[r, :c] for r, c in enumerate(counts)], axis=0)
    return unpack_detections(full, layout)


def gather_detections(driver, det_or_none, n_local, counts, device=None, group=None):
    """One entry point for both process-group kinds: RCCL (`nccl`) gathers the handle's device-resident records
    (`all_gather_detections_device`), a CPU group (`gloo`, tests) goes through host arrays (`all_gather_detections`).
    det_or_none: the local detections already on the host, or None (they are then collected only on the host path)."""
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl" and device is not None:
        return all_gather_detections_device(driver, n_local, counts, device, group=group)
    det = det_or_none
    if det is None:
        det = driver._collect(n_local) if n_local > 0 else driver.empty_detections()
    return all_gather_detections(det, device=device, group=group, counts=counts)


def serve_sharded(driver, images, rank, world, device=None, group=None):
    """Image-sharded serve: each rank runs the path on its contiguous shard, then one
    all-gather returns the full batch's detections on every rank."""
    import torch.distributed as dist
    start, stop = shard_range(len(images), rank, world)
    driver.set_image_offset(start)      # same dropout masks as the unsharded batch
    counts = [shard_range(len(images), r, world)[1] - shard_range(len(images), r, world)[0] for r in range(world)]
    if dist.get_backend(group) == "nccl" and device is not None:      # detections stay on the device until they are gathered
        if stop > start:
            driver.serve_resident(images[start:stop])
        return all_gather_detections_device(driver, stop - start, counts, device, group=group)
    if stop > start:
        det = driver.serve(images[start:stop])
    else:  # more ranks than images: this rank contributes an empty shard
        det = driver.empty_detections()
    return all_gather_detections(det, device=device, group=group, counts=counts)


# ---------------------------------------------------------------------------------------------- ensemble striping
def member_owner(member, world):
    """Rank that runs ensemble member `member` (members striped round-robin over the ranks)."""
    return member % world


def reshard_member_heads(owned, n_members, n_total, rank, world, device=None, group=None):
    """Deep-ensemble exchange (BASELINE configs[3]): every member ran on ONE rank for ALL images;
    afterwards every rank needs ALL members for ITS contiguous image shard.  One all-to-all-v built from batched
    point-to-point transfers (each GPU pair uses its own xGMI link; no ring): rank r sends to rank j the head
    outputs of the members it owns restricted to shard j.

    owned: {member: (cls_levels, box_levels)} with per-level float32 arrays [n_total, ...] (any trailing shape).
    Returns (cls_levels, box_levels) with arrays [n_members, n_local, ...] for this rank's shard."""
    import torch
    import torch.distributed as dist
    dev = torch.device(device) if device is not None else torch.device("cpu")
    a, b = shard_range(n_total, rank, world)
    mine = sorted(owned)
    assert mine == [m for m in range(n_members) if member_owner(m, world) == rank], "member ownership must be round-robin"
    template = None
    if mine:
        c0, b0 = owned[mine[0]]
        template = [tuple(x.shape[1:]) for x in c0] + [tuple(x.shape[1:]) for x in b0]
        n_cls = len(c0)
    # every rank needs the per-image payload layout; ranks that own no member learn it from rank 0 (which owns member 0)
    meta = [template, n_cls if mine else None]
    dist.broadcast_object_list(meta, src=0, group=group)
    template, n_cls = meta
    per_image = int(sum(int(np.prod(s)) for s in template))

    def flat(m, s, e):
        c, bx = owned[m]
        return np.concatenate([x[s:e].reshape(e - s, -1) for x in list(c) + list(bx)], axis=1)

    ops, recv = [], {}
    send_keep = []
    for j in range(world):                                   # what I send to rank j
        s, e = shard_range(n_total, j, world)
        if e == s or not mine:
            continue
        payload = np.ascontiguousarray(np.stack([flat(m, s, e) for m in mine]), np.float32)   # [owned, n_j, per_image]
        if j == rank:
            recv[rank] = torch.from_numpy(payload)
            continue
        t = torch.from_numpy(payload).to(dev)
        send_keep.append(t)
        ops.append(dist.P2POp(dist.isend, t, j, group=group))
    for r in range(world):                                   # what I receive from rank r
        theirs = [m for m in range(n_members) if member_owner(m, world) == r]
        if r == rank or not theirs or b == a:
            continue
        t = torch.empty((len(theirs), b - a, per_image), dtype=torch.float32, device=dev)
        recv[r] = t
        ops.append(dist.P2POp(dist.irecv, t, r, group=group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    full = np.zeros((n_members, b - a, per_image), np.float32)
    for r, t in recv.items():
        theirs = [m for m in range(n_members) if member_owner(m, world) == r]
        full[theirs] = t.cpu().numpy()
    outs, off = [], 0
    for shp in template:
        k = int(np.prod(shp))
        outs.append(full[:, :, off:off + k].reshape((n_members, b - a) + tuple(shp)))
        off += k
    return outs[:n_cls], outs[n_cls:]


# ---------------------------------------------------------------------------------------------- MC-sample sharding
def sample_owner(t, world):
    """Rank that runs MC sample t (samples striped round-robin: rank r owns t = r, r + world, ...)."""
    return t % world


def local_samples(T, rank, world):
    """The global sample indices rank `rank` runs."""
    return list(range(rank, T, world))


class SampleShardedDriver:
    """north_star: "images (and optionally MC samples) shard".  The reference's own callers serve ONE image per call
    (validate_model.py:476-522, infer_model.py:554-581): image sharding then uses one GPU of the node.  Here the T MC samples of
    a batch are striped over the ranks instead - rank r runs the network for samples t = r (mod world) of EVERY image (its
    handle is built for T_r = |{t}| samples and told which global samples they are: `set_sample_shard`, so the ranks together
    draw exactly the dropout masks of one handle that runs all T) - and the head outputs are then re-sharded by image with the
    ensemble's all-to-all-v (`reshard_member_heads`: a sample is a "member"), so that the owner of an image holds its T rows in
    order t = 0 ... T - 1 and aggregates them with the same sequential float32 sums as a single process: detections are
    bit-identical to one GPU.  One more exchange than image sharding (the head outputs: 11 MB per image and sample at
    D0 / 1280 x 768) in return for 1 / world of the network time per image."""

    def __init__(self, model_name, batch_size, model_params, weights, rank, world, device=0, post_mode="global"):
        from . import infer_lib
        p = dict(model_params)
        self.T = int(p["mc_dropoutsamp"]) if p.get("mc_dropout") else 1
        if self.T < world:
            raise ValueError("%d MC samples cannot be striped over %d ranks" % (self.T, world))
        self.rank, self.world = rank, world
        self.mine = local_samples(self.T, rank, world)
        pn = dict(p, mc_dropoutsamp=len(self.mine), uda_force_sample_axis=True)      # (one sample on a rank: same lowering as several)
        self.net = infer_lib.ServingDriver(model_name, batch_size, True, pn, weights=weights, device=device)
        self.net.set_sample_shard(rank, world, self.T)
        self.post = infer_lib.ServingDriver(model_name, batch_size, False, p, device=device, post_only=True, post_mode=post_mode)
        self.stacked = (bool(self.post.plan.cls_stacked), bool(self.post.plan.box_stacked))      # the reference's stacking rule, full T

    def set_dropout_seed(self, seed):
        self.net.set_dropout_seed(seed)

    def close(self):
        self.net.close()
        self.post.close()

    def serve(self, images, device=None, group=None, post_mode=None):
        """uint8 images (every rank passes the SAME batch) -> the full batch's detections on every rank."""
        n_total = len(images)
        self.net.set_image_offset(0)
        n = self.net.run_network(images)
        cls, box = self.net.head_outputs(n)                  # stacked heads: [T_r, n, ...] per level; the others [n, ...]
        _, scales = self.net.preprocessed_scales(n)
        a, b = shard_range(n_total, self.rank, self.world)
        # stacked heads come back [T_r, n, ...] (a rank with ONE sample: [n, ...], the driver drops a sample axis of one)
        cls = [c[None] if (self.stacked[0] and c.ndim == 4) else c for c in cls]
        box = [x[None] if (self.stacked[1] and x.ndim == 4) else x for x in box]
        owned = {}
        for j, t in enumerate(self.mine):
            owned[t] = ([np.ascontiguousarray(c[j]) if self.stacked[0] else np.zeros((n, 0), np.float32) for c in cls],
                        [np.ascontiguousarray(x[j]) if self.stacked[1] else np.zeros((n, 0), np.float32) for x in box])
        rc, rb = reshard_member_heads(owned, self.T, n_total, self.rank, self.world, device=device, group=group)
        counts = [shard_range(n_total, r, self.world)[1] - shard_range(n_total, r, self.world)[0] for r in range(self.world)]
        if b > a:
            # a head that is not stacked is deterministic: every rank computed the same tensor - take this rank's own copy
            cls_in = [rc[l] if self.stacked[0] else cls[l][a:b] for l in range(len(cls))]
            box_in = [rb[l] if self.stacked[1] else box[l][a:b] for l in range(len(box))]
            det = self.post.postprocess(cls_in, box_in, np.asarray(scales[a:b], np.float32), post_mode=post_mode)
        else:
            det = self.post.empty_detections()
        return all_gather_detections(det, device=device, group=group, counts=counts)


def serve_sample_sharded(driver, images, device=None, group=None):
    """`SampleShardedDriver.serve` under the name the other sharded entry points use."""
    return driver.serve(images, device=device, group=group)


class DevArray:
    """Zero-copy view of a device buffer owned by a HIP handle for torch (`torch.as_tensor(DevArray(...), device=...)`):
    float32, C-contiguous, exposed through `__cuda_array_interface__` (version 2)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = dict(shape=tuple(int(v) for v in shape), typestr="<f4", data=(int(ptr), False),
                                             version=2, strides=None)


def _head_tensor(torch, drv, level, which, dev, rows_expected=None):
    """The handle's head-output buffer of (level, which) as a torch tensor [capacity, rows_per_image, floats_per_row]."""
    ptr, per, rows = drv.head_outputs_device(level, which)
    if rows_expected is not None and rows != rows_expected:
        raise ValueError("head buffer (level %d, %s) holds %d rows per image, the exchange needs %d: ensemble members must "
                         "be deterministic networks (one row per image; a strided view is not a contiguous send buffer)"
                         % (level, "box" if which else "class", rows, rows_expected))
    return torch.as_tensor(DevArray(ptr, (drv._cap, rows, per)), device=dev)


def reshard_member_heads_device(member_drivers, post_driver, n_members, n_total, rank, world, device, group=None):
    """The ensemble exchange of `reshard_member_heads` WITHOUT a host hop: the members' head outputs are sent
    straight out of their handles' device buffers and land in the sample slots of the aggregating handle - batched
    point-to-point transfers over RCCL (every GPU pair uses its own xGMI link), 707 MB per member at BASELINE configs[3]
    that never cross PCIe.  Shards that stay on the rank are device-to-device copies."""
    import torch
    import torch.distributed as dist
    dev = torch.device(device)
    a, b = shard_range(n_total, rank, world)
    mine = sorted(member_drivers)
    assert mine == [m for m in range(n_members) if member_owner(m, world) == rank], "member ownership must be round-robin"
    for drv in member_drivers.values():
        drv.synchronize()                       # the members' heads are complete before torch's stream reads them
    post_driver.synchronize()
    levels = len(post_driver.plan.level_hw)
    ops, pending, keep, recv_from = [], [], [], []
    stage_off = 0
    for lvl in range(levels):
        for which in (0, 1):
            dst = _head_tensor(torch, post_driver, lvl, which, dev, rows_expected=n_members)   # [cap, M, per]
            # one row per image in a member handle: [cap, 1, per] -> [cap, per] is contiguous, and so is every [s:e] slice
            srcs = {m: _head_tensor(torch, member_drivers[m], lvl, which, dev, rows_expected=1)[:, 0, :] for m in mine}
            assert all(t.is_contiguous() for t in srcs.values())
            for j in range(world):                                            # what this rank sends
                s, e = shard_range(n_total, j, world)
                if e == s:
                    continue
                for m in mine:
                    if j == rank:
                        dst[:e - s, m, :].copy_(srcs[m][s:e])
                    else:
                        ops.append(dist.P2POp(dist.isend, srcs[m][s:e], j, group=group))
            if b > a:
                for r in range(world):                                        # what this rank receives
                    if r == rank:
                        continue
                    for m in (mm for mm in range(n_members) if member_owner(mm, world) == r):
                        # (a sample slot of the aggregating buffer is a strided view - row pitch n_members x per - and RCCL
                        # receives into contiguous memory: the receives land in ONE reusable staging tensor, carved in
                        # posting order, and are scattered into their slots afterwards)
                        k = (b - a) * dst.shape[2]
                        pending.append((dst, m, stage_off, k))
                        stage_off += k
                        recv_from.append(r)
                        ops.append("recv")
            keep.append((dst, srcs))
    # every rank posts its operations in the same global order (level, head, peer, member): matching sends and receives.
    # The sends of all (level, head) pairs were appended first per pair, the receives right behind them - keep that order.
    stage = None
    if pending:
        key = (str(dev), "ensemble-stage")
        stage = _GATHER_BUF.get(key)
        if stage is None or stage.numel() < stage_off:
            stage = _GATHER_BUF[key] = torch.empty((stage_off,), dtype=torch.float32, device=dev)
    ordered = []
    ri = 0
    for op in ops:
        if op == "recv":
            dst, m, off, k = pending[ri]
            ordered.append(dist.P2POp(dist.irecv, stage[off:off + k].view(b - a, dst.shape[2]), recv_from[ri], group=group))
            ri += 1
        else:
            ordered.append(op)
    if ordered:
        for w in dist.batch_isend_irecv(ordered):
            w.wait()
    for dst, m, off, k in pending:
        dst[:b - a, m, :].copy_(stage[off:off + k].view(b - a, dst.shape[2]))
    torch.cuda.synchronize(dev)                 # the aggregating handle's stream may read its buffers now
    post_driver.heads_written_externally(b - a)
    return b - a


def serve_ensemble_striped(member_drivers, post_driver, images, n_members, rank, world, device=None, group=None,
                           post_mode=None):
    """Deep ensemble with the members striped over the ranks: member m runs on rank m % world for the whole batch,
    the head outputs are re-sharded by image (`reshard_member_heads`), every rank aggregates / decodes / NMSes its
    image shard with all members as the sample axis, one all-gather returns the batch's detections everywhere.

    member_drivers: {member: ServingDriver} for the members this rank owns (deterministic networks, batch = all
    images); post_driver: an aggregating ServingDriver planned with mc_dropoutsamp = n_members (see EnsembleDriver)."""
    n = len(images)
    owned = {}
    scales = None
    for m, drv in sorted(member_drivers.items()):
        drv._feed(images)
        drv._ck(drv._lib.uda_run(drv._h, -1, 0), "uda_run")
        owned[m] = drv
        _, scales = drv.preprocessed_scales(n)
    import os
    import torch.distributed as dist
    # UDA_ENSEMBLE_EXCHANGE=host forces the exchange through host arrays under RCCL too (the device-resident
    # point-to-point branch moves 707 MB per member without touching PCIe, but a two-GPU RCCL run of it has only
    # ever been possible on the driver's multi-GPU node, never on the builder's one-GPU box)
    on_device = (dist.get_backend(group) == "nccl" and device is not None
                 and os.environ.get("UDA_ENSEMBLE_EXCHANGE", "device") != "host")
    if on_device:
        reshard_member_heads_device(owned, post_driver, n_members, n, rank, world, device, group=group)
    else:       # CPU process group (gloo: tests): through host arrays
        owned = {m: drv.head_outputs(n) for m, drv in owned.items()}
        cls_lv, box_lv = reshard_member_heads(owned, n_members, n, rank, world, device=device, group=group)
    a, b = shard_range(n, rank, world)
    if scales is None:      # a rank without a member: the image scale depends only on the raw size (dataloader.py:123-135)
        h, w = np.asarray(images).shape[1:3]
        H, W = post_driver.image_size
        s = min(np.float32(H) / np.float32(h), np.float32(W) / np.float32(w))
        scales = np.full((n,), np.float32(1.0) / np.float32(s), np.float32)
    if b > a:
        sc = np.asarray(scales, np.float32)[a:b]
        if on_device:
            cls_lv, box_lv = post_driver.device_heads(b - a)      # resident: post-processed where the exchange left them
        det = post_driver.postprocess(cls_lv, box_lv, sc, post_mode=post_mode, collect=not on_device)
    else:
        det = post_driver.empty_detections()
    counts = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
    if on_device:       # the records are gathered where the post-process left them
        return all_gather_detections_device(post_driver, b - a, counts, device, group=group)
    return all_gather_detections(det, device=device, group=group, counts=counts)
