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


def all_gather_detections(det, device=None, group=None):
    """Every rank gets the detections of all images, in global image order.  Ranks may own
    different numbers of images (ragged shards are padded to the largest shard for the
    collective and trimmed afterwards)."""
    import torch
    import torch.distributed as dist
    packed, layout = pack_detections(det)
    world = dist.get_world_size(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    n_local = torch.tensor([packed.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    buf = np.zeros((n_max,) + packed.shape[1:], np.float32)
    buf[:packed.shape[0]] = packed
    t = torch.from_numpy(buf).to(dev)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    full = np.concatenate([o.cpu().numpy()[:c] for o, c in zip(outs, counts)], axis=0)
    return unpack_detections(full, layout)


def serve_sharded(driver, images, rank, world, device=None, group=None):
    """Image-sharded serve: each rank runs the path on its contiguous shard, then one
    all-gather returns the full batch's detections on every rank."""
    start, stop = shard_range(len(images), rank, world)
    driver.set_image_offset(start)      # same dropout masks as the unsharded batch
    if stop > start:
        det = driver.serve(images[start:stop])
    else:  # more ranks than images: this rank contributes an empty shard
        det = driver.empty_detections()
    return all_gather_detections(det, device=device, group=group)
