#!/usr/bin/env python
"""Headline benchmark: images x MC-samples / s (and p50 detect latency) of the MC-dropout
EfficientDet-D0 path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one batch of synthetic input already resident in HBM:
preprocess -> EfficientDet-D0 x T (Philox MC dropout) -> MC mean/std + variance-propagating
decode -> NonMaxSuppressionV5 -> packed detections, for `--batch` images per GPU (weak scaling:
every rank owns its own shard of images; the only exchange is the all-gather of the KB-scale
detection records at the end of each step).  Workload = BASELINE.json configs[1]: D0, 32 synthetic
KITTI-resolution images (1280x768), T=10, full MC dropout + loss attenuation, C=7.
`--config 2|3|4` selects the per-GPU share of BASELINE configs[2] (BDD-like, C=10, T=20), [3]
(5-member deep ensemble, members striped over the ranks) or [4] (D2 at 1024x1024, T=30, per-class NMS).

The main leg (value, ms_per_step, roofline, kernel_ms_per_step) runs the shipped default: float32 tensors and accumulators,
1x1 products as two fp16 pieces per operand with three cross terms on the matrix cores (UDA_PW_SCHEME=f16x2, ~2^-22 per
product: float32-class arithmetic, held to the same parity bars as the six-term bf16 scheme - tests/test_gpu_ops.py,
tests/test_gpu_round3.py, tests/test_gpu_fullsize.py).

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline       dominant kernel kind: algorithmic bytes / HIP-event device time vs the 8 TB/s HBM peak
  cpu_baseline   the CPU oracle (restatement, not TF) timed on a bounded sample in a subprocess
  precision      the same workload under the other schemes - bf16x3 (three bf16 pieces, six cross terms), bf16x2 (two bf16
                 pieces, three terms: narrower than float32, a side figure only) and exact f32-input MFMA - each measured in
                 a child process that carries the switch in its environment before its first GPU call
  configs        BASELINE configs[2], [3], [4] (per-GPU share) with the main protocol, each in a child process
  head_only      head-only MC dropout (configs/train/*_head.yaml; SURVEY 8d: 1.545 GMAC per image x sample)
  nms_spread_scores   the headline workload with a spread score distribution (--cls-spread 20: few confident clusters)
  rccl_world1    the headline through the process-group path (device-resident all-gather of the detections) in a world of one
  h2d_inclusive_pipelined / h2d_inclusive_serial   ms_per_step with a fresh uint8 batch fed inside every step (the reference
                 times serve(image) including the feed, validate_model.py:154-158): pipelined = host bytes -> pinned staging ->
                 DMA on the copy stream into the second input slot while the current batch computes
                 (uda_prefetch_images_u8 / uda_swap_prefetched); serial = the upload inside the step, nothing hidden
  ranks          (N > 1) what RCCL actually formed: world size and every rank's device (name, uuid, PCI bus id)
  p50_detect_latency_ms   batch-1 serve() of one image, T as configured, upload and download included
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
KIND_NAMES = {1: "stem", 2: "pw", 3: "dw", 4: "se", 5: "fuse", 6: "pool", 7: "mbx", 8: "sep", 16: "aggregate", 17: "nms",
              18: "preprocess"}
# kernels behind each op kind (tools/traffic_from_pmc.py groups rocprofv3 kernel names with the same table)
KIND_KERNELS = {"stem": "stem_u8_kernel | stem16_kernel", "pw": "pwb_kernel", "dw": "dw_kernel", "se": "se_kernel", "fuse": "fuse_kernel",
                "pool": "fuse_kernel", "mbx": "mbxb_kernel+mbxd_kernel+mbxp_kernel", "sep": "sep_kernel+sepf_kernel", "aggregate": "aggregate_reg_kernel",
                "nms": "nms_coop_kernel", "preprocess": "preprocess_kernel"}
LAYERWISE_MB_PER_UNIT = {("efficientdet-d0", "1280x768", 7): 1798.7}   # SURVEY 8d, full MC
CONFIG_NAMES = {1: "BASELINE configs[1]", 2: "BASELINE configs[2], per-GPU share", 3: "BASELINE configs[3]",
                4: "BASELINE configs[4], per-GPU share"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4], help="BASELINE.json configs index (1 = the headline)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step")
    ap.add_argument("--samples", type=int, default=None, help="MC samples T")
    ap.add_argument("--image-size", default=None, help="WxH as the reference writes it")
    ap.add_argument("--raw-size", default=None, help="WxH of the raw uint8 images (default: the network size)")
    ap.add_argument("--classes", type=int, default=None)
    ap.add_argument("--variant", default="full", choices=["full", "head"],
                    help="full: mc_dropoutrate=0.05 everywhere; head: class/box head dropout only")
    ap.add_argument("--chunk", type=int, default=None, help="images per pass of the op list")
    ap.add_argument("--model", default=None)
    ap.add_argument("--post-mode", default=None, choices=["global", "per_class"])
    ap.add_argument("--ensemble", type=int, default=0, help="deep ensemble of this many deterministic members instead of MC dropout")
    ap.add_argument("--cls-spread", type=float, default=1.0,
                    help="scale of the class-head output layer: 1 = the reference initialiser (near-tied scores), "
                         "larger = spread-out scores as a trained head gives (side measurement, never the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the precision children, the H2D-inclusive and the batch-1 legs")
    ap.add_argument("--child", action="store_true", help="(internal) GPU leg only, short JSON")
    ap.add_argument("--force-dist", action="store_true", help="use the process group (RCCL) path even at world size 1")
    ap.add_argument("--protocol", default="pipelined", choices=["pipelined", "serial"],
                    help="pipelined (default): step k + 1's network is queued before step k's detections are fetched, step k's "
                         "post-process runs beside it (uda_run_async / uda_collect: what ServingDriver.serve_stream does); "
                         "serial: one step at a time, as rounds 1-3 timed it")
    ap.add_argument("--cpu-sample-images", type=int, default=1)
    a = ap.parse_args()
    preset = {1: dict(batch=32, samples=10, image_size="1280x768", classes=7, model="efficientdet-d0", post_mode="global"),
              2: dict(batch=32, samples=20, image_size="1280x768", raw_size="1280x720", classes=10, model="efficientdet-d0",
                      post_mode="global"),
              3: dict(batch=8, samples=1, image_size="1280x768", classes=7, model="efficientdet-d0", post_mode="global", ensemble=5),
              4: dict(batch=2, samples=30, image_size="1024x1024", classes=7, model="efficientdet-d2", post_mode="per_class")}[a.config]
    for k, v in preset.items():
        if getattr(a, k) in (None, 0):
            setattr(a, k, v)
    if a.raw_size is None:
        a.raw_size = a.image_size
    if a.chunk is None:
        a.chunk = a.batch
    return a


def make_params(a):
    from uda_amd import hparams_config
    cfg = hparams_config.get_efficientdet_config(a.model)
    over = dict(image_size=a.image_size, num_classes=a.classes, loss_attenuation=True, enable_softmax=True)
    if not a.ensemble:
        over.update(mc_dropout=True, mc_dropoutsamp=a.samples)
        over.update(dict(mc_dropoutrate=0.05) if a.variant == "full" else dict(mc_classheadrate=0.05, mc_boxheadrate=0.05))
    if a.config == 4:           # eval settings (eval.py:75)
        over.update(nms_configs=dict(max_nms_inputs=5000))
    cfg.override(over)
    p = cfg.as_dict()
    p["is_training_bn"] = False
    return p


def cpu_baseline(a):
    """Oracle timed in a clean subprocess (no GPU runtime in that process)."""
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--image-size", a.image_size,
           "--classes", str(a.classes), "--samples", str(max(a.samples, 1)), "--images", str(a.cpu_sample_images),
           "--variant", a.variant, "--model", a.model]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
        return {"value": None, "unit": "images*MC-samples/s", "cores": None, "kind": "port",
                "sample": "failed: %r" % (e,)}


SCHEME_NOTE = {"f16x2": "two fp16 pieces per operand, 3 cross terms (~2^-22 per product; float32-class, the default)",
               "bf16x3": "three bf16 pieces per operand, 6 cross terms (~2^-24 per product; float32-equivalent)",
               "bf16x2": "two bf16 pieces per operand, 3 cross terms (~2^-17 per product; narrower than float32)",
               "f32": "exact f32-input MFMA (unfused deep blocks and separable convs: a debugging reference)"}


def child_leg(a, extra=(), env=None, same_shape=True):
    """One more measurement with the main leg's protocol (--steps / --warmup) in a child process: the library reads its
    switches once, at its first call, and every leg starts from a fresh GPU context.  Returns the child's short JSON."""
    cmd = [sys.executable, os.path.abspath(__file__), "--child", "--no-cpu-baseline", "--no-side", "--steps", str(a.steps),
           "--warmup", str(a.warmup), "--protocol", a.protocol]
    if same_shape:
        cmd += ["--config", str(a.config), "--batch", str(a.batch), "--samples", str(a.samples), "--image-size", a.image_size,
                "--raw-size", a.raw_size, "--classes", str(a.classes), "--variant", a.variant, "--chunk", str(a.chunk), "--model", a.model,
                "--post-mode", a.post_mode, "--ensemble", str(a.ensemble)]
    cmd += list(extra)
    e = dict(os.environ)
    e.update(env or {})
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=e)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as ex:
        return {"ms_per_step": None, "error": repr(ex)}


def rank_devices(dist, torch, rank, local_rank):
    """What RCCL formed: every rank reports the device it computes on; rank 0 gets the list (a SCALE run can then show N ranks
    on N distinct GPUs, not N processes on one)."""
    pr = torch.cuda.get_device_properties(local_rank)
    me = {"rank": rank, "local_device": local_rank, "name": pr.name, "uuid": str(getattr(pr, "uuid", "")),
          "pci_bus_id": getattr(pr, "pci_bus_id", None), "pci_device_id": getattr(pr, "pci_device_id", None),
          "visible": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))}
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, me)
    ids = {(g["uuid"], g["pci_bus_id"], g["visible"], g["local_device"]) for g in got}
    return {"world_size_formed": dist.get_world_size(), "backend": dist.get_backend(), "distinct_devices": len(ids), "devices": got}


def log(msg):
    print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    tdev = None
    if a.gpus > 1 and world != a.gpus:
        sys.exit("bench.py --gpus %d needs one process per GPU: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                 "--master-addr 127.0.0.1 --master-port <port> bench.py --gpus %d ... (WORLD_SIZE is %d)" % (a.gpus, a.gpus, a.gpus, world))
    if world > 1 or a.force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "RANK" not in os.environ:           # --force-dist in a plain process: a world of one over RCCL
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
        import torch                    # torch first: the HIP library then binds to the same runtime
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        tdev = torch.device("cuda", local_rank)
        # RCCL prints a version banner on stdout at communicator creation; stdout carries exactly one
        # JSON line (the driver's contract), so the banner is sent to stderr.
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=tdev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        ranks_info = rank_devices(dist, torch, rank, local_rank)
        world = dist.get_world_size()          # what RCCL formed, not what the environment asked for
    else:
        ranks_info = None

    from uda_amd import plan as plan_mod, weights as weights_mod
    from uda_amd.infer_lib import EnsembleDriver, KerasDriver

    params = make_params(a)
    W_, H_ = [int(v) for v in a.raw_size.lower().split("x")]
    images = np.random.default_rng(2 + rank).integers(0, 256, (a.batch, H_, W_, 3), dtype=np.uint8)
    scheme = plan_mod.pw_scheme()

    if a.ensemble:
        return ensemble_main(a, params, images, rank, world, local_rank, dist, tdev, ranks_info)

    w = weights_mod.init_weights(params, seed=0, cls_spread=a.cls_spread)
    drv = KerasDriver("_", False, a.model, a.batch, False, params, weights=w, device=local_rank,
                      chunk_images=min(a.chunk, a.batch), post_mode=a.post_mode)
    log("driver ready: %s" % (drv.plan.summary(),))
    drv.set_image_offset(rank * a.batch)           # Philox rows of the global (weak-scaled) batch
    t_up = time.perf_counter()
    drv.stage_images(images)                       # PCIe leg, outside the timed region
    upload_s = time.perf_counter() - t_up

    def barrier():
        if dist is not None:
            dist.barrier()
        drv.synchronize()

    def step(upload=False):
        if upload:
            drv.stage_images(images)
        if dist is not None:
            # the detections stay in the handle's device buffer until RCCL has gathered them; ONE download of the gathered
            # records (dist.all_gather_detections_device)
            from uda_amd.dist import all_gather_detections_device
            drv.run_resident(sync=False)
            return all_gather_detections_device(drv, a.batch, [a.batch] * world, tdev)
        drv.run_resident(sync=True)
        return drv._collect(a.batch)

    # warm-up; the last warm-up step also ranks the kernel kinds by device time.  The very first call of a handle is
    # set-up (lazy allocations, and the global NMS probes whether a score prefix suffices for this score distribution,
    # see DESIGN.md section 5), so there are always at least two untimed steps (`warmup_steps_run` in the line).
    kinds = [1, 2, 3, 4, 5, 6, 7, 8, 16, 17, 18]
    calib = {}
    n_warm = max(2, a.warmup)
    for i in range(n_warm):
        if i == n_warm - 1:
            drv.profile_enable(kinds)
        step()
        if i == n_warm - 1:
            calib = {k: drv.profile_read(k) for k in kinds}
            drv.profile_enable([])
        log("warm-up step %d done; kernel ms by kind: %s" % (i, {KIND_NAMES[k]: round(v[0], 1) for k, v in calib.items()}))
    dominant = max(calib, key=lambda k: calib[k][0])
    drv.profile_enable([dominant])                 # HIP events around that kind only, inside the timed region

    def consume(ticket):
        """step `ticket`'s detections: to the host (one GPU) / gathered over RCCL from the device-resident records (ranks)"""
        if dist is not None:
            from uda_amd.dist import all_gather_detections_device
            return all_gather_detections_device(drv, a.batch, [a.batch] * world, tdev, ticket=ticket)
        return drv.collect(ticket)

    def timed_steps(protocol, n_steps):
        """EXACTLY n_steps steps between two barriers; every step = network x T + post-process + its detections fetched."""
        barrier()
        lat_ = []
        t0_ = time.perf_counter()
        if protocol == "serial":
            for _ in range(n_steps):
                ts = time.perf_counter()
                step()
                lat_.append(time.perf_counter() - ts)
        else:
            ts = time.perf_counter()
            prev = None
            for _ in range(n_steps):
                cur = drv.run_async()              # queued: this step's network starts behind the previous step's network ...
                if prev is not None:
                    consume(prev)                  # ... while the previous step's post-process finishes beside it
                    lat_.append(time.perf_counter() - ts)
                    ts = time.perf_counter()
                prev = cur
            consume(prev)
            lat_.append(time.perf_counter() - ts)
        barrier()
        return time.perf_counter() - t0_, lat_

    if a.protocol == "pipelined":
        consume(drv.run_async())                   # untimed: the second output set and the events of the pipelined runs
        drv.profile_read(dominant)                 # (reset: the dominant kind's events cover the timed region only)
    elapsed, lat = timed_steps(a.protocol, a.steps)
    dom_ms, dom_launches = drv.profile_read(dominant)
    drv.profile_enable([])
    for i, l in enumerate(lat):
        log("timed step %d: %.1f ms" % (i + 1, l * 1e3))
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    costs = plan_mod.op_costs(drv.plan, a.batch)
    units = world * a.batch * a.samples * a.steps
    value = units / elapsed
    other_protocol = None
    if not a.no_side or a.child:
        oth = "serial" if a.protocol == "pipelined" else "pipelined"
        if oth == "pipelined":
            consume(drv.run_async())
        e2, _ = timed_steps(oth, a.steps)
        if dist is not None:
            import torch
            t2 = torch.tensor([e2], dtype=torch.float64, device=tdev)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            e2 = float(t2.item())
        other_protocol = {"protocol": oth, "ms_per_step": round(e2 / a.steps * 1e3, 2), "value": round(units / e2, 2)}

    if a.child:
        if rank == 0:
            print(json.dumps({"ms_per_step": round(elapsed / a.steps * 1e3, 2), "value": round(value, 2), "unit": "images*MC-samples/s",
                              "steps": a.steps, "warmup": a.warmup, "UDA_PW_SCHEME": scheme, "protocol": a.protocol,
                              "other_protocol": other_protocol,
                              "workload": "%s, %d images (%s raw, %s network), T=%d (%s), C=%d, %s" % (
                                  a.model, a.batch, a.raw_size, a.image_size, a.samples, a.variant, a.classes, a.post_mode) +
                                          ("" if a.cls_spread == 1.0 else ", class-predict kernel x %g" % a.cls_spread) +
                                          (", process-group path (RCCL, world %d)" % world if dist is not None else ""),
                              "kernel_ms_per_step": {KIND_NAMES.get(k, str(k)): round(v[0], 2) for k, v in calib.items()},
                              "nms_coop_fallbacks": drv.nms_coop_fallbacks(), "nms_coop_not_launched": drv.nms_coop_not_launched()}), flush=True)
        drv.close()
        return

    # ---- side legs (rank 0 of a single-GPU run only; never part of `value`)
    side = {}
    if world == 1 and not a.no_side:
        k_side = max(3, a.steps)
        fresh = [images, np.ascontiguousarray(images[::-1])]           # two different host batches, alternating
        drv.stage_images(fresh[0])
        drv.synchronize()
        t1 = time.perf_counter()
        for k in range(k_side):
            drv.run_resident(sync=False)                   # queue this step's kernels
            drv.prefetch_images(fresh[(k + 1) & 1])        # next batch: host bytes -> pinned -> DMA on the copy stream, meanwhile
            drv._collect(a.batch)                          # this step's detections (synchronises the compute stream)
            drv.swap_prefetched()
        drv.synchronize()
        side["h2d_inclusive_pipelined"] = {"ms_per_step": round((time.perf_counter() - t1) / k_side * 1e3, 2),
                                           "note": "a fresh uint8 batch (%.0f MB, pageable host memory) fed in every step: gathered into the "
                                                   "handle's pinned staging buffer and uploaded on the copy stream under the previous step's "
                                                   "kernels (uda_prefetch_images_u8 / uda_swap_prefetched)" % (images.nbytes / 1e6)}
        step(upload=True)                                   # warm-up of the serial path (first-touch of its staging path)
        drv.synchronize()
        t2 = time.perf_counter()
        for k in range(k_side):
            drv.stage_images(fresh[k & 1])                  # the upload inside the step, nothing hidden (what serve(images) does)
            drv.run_resident(sync=True)
            drv._collect(a.batch)
        side["h2d_inclusive_serial"] = {"ms_per_step": round((time.perf_counter() - t2) / k_side * 1e3, 2),
                                        "note": "the same with the upload serial inside every step (the reference's protocol, "
                                                "validate_model.py:154-158): comparable with rounds 1-2's h2d_inclusive"}
    coop_fb, pfx_fb, coop_nl = drv.nms_coop_fallbacks(), drv.nms_prefix_fallbacks(), drv.nms_coop_not_launched()
    summary = drv.plan.summary()
    drv.close()
    if world == 1 and not a.no_side:
        d1 = KerasDriver("_", False, a.model, 1, False, params, weights=w, device=local_rank, chunk_images=1, post_mode=a.post_mode)
        one = images[:1]
        for _ in range(3):
            d1.serve(one)
        l1 = []
        for _ in range(20):
            ts = time.perf_counter()
            d1.serve(one)
            l1.append(time.perf_counter() - ts)
        side["p50_detect_latency_ms"] = round(float(np.median(l1)) * 1e3, 2)
        side["detect_latency_note"] = "batch 1, T=%d, serve(image) = upload + preprocess + network x T + post-process + download; 20 calls" % a.samples
        coop_fb += d1.nms_coop_fallbacks()
        coop_nl += d1.nms_coop_not_launched()
        d1.close()

    if rank == 0:
        roof = None
        if dominant in costs and dom_launches:
            cst = costs[dominant]
            per_launch_bytes = cst["bytes"] / cst["launches"]
            avg_ms = dom_ms / dom_launches
            achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": KIND_KERNELS.get(KIND_NAMES.get(dominant, ""), str(dominant)),
                    "op_kind": KIND_NAMES.get(dominant, str(dominant)),
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "hbm_achieved": round(achieved, 1), "hbm_peak": HBM_PEAK_GBS, "hbm_frac": round(achieved / HBM_PEAK_GBS, 4),
                    "avg_launch_ms": round(avg_ms, 4), "launches": int(dom_launches),
                    "algorithmic_bytes_per_launch": int(per_launch_bytes),
                    "tflops": round(cst["flops"] / cst["launches"] / (avg_ms * 1e-3) / 1e12, 2),
                    "share_of_step": round(dom_ms / a.steps / (elapsed / a.steps * 1e3), 3)}
            if KIND_NAMES.get(dominant, "") == "mbx":
                roof["note"] = ("fused MBConv front halves: bound by vector issue, not by HBM - SQ counters of the committed profile "
                                "(profiles/r05_sq.txt, this scheme): VALU 73 % busy over the family, 75-88 % in blocks 1-10, at the 1.8-2.0 GHz the "
                                "chip holds under them; two transcendentals per swish = 60 % of block 1's vector time, instruction counts at the "
                                "floor of the algorithm (DESIGN.md 4.5-4.7); plain streaming kernels reach 4.5-5.7 TB/s on this box "
                                "(tools/micro/hbm_rates.hip)")
            # What bounds the dominant family, from the committed counter profile of this command (profiles/roofline_inputs.json,
            # written by tools/summarize_r05.py from the SQ / FETCH / WRITE passes): the fused MBConv front halves are bound by
            # vector issue - `bound` = "valu", `achieved` / `frac` = share of the VALU issue cycles that are busy, re-scaled
            # with THIS run's launch duration (the instruction count of a launch does not change) - and `traffic` = counter
            # bytes per launch; the HBM view stays beside it (hbm_*).
            ri = None
            try:
                ri = json.load(open(os.path.join(ROOT, "profiles", "roofline_inputs.json")))
            except Exception:
                pass
            kn = KIND_NAMES.get(dominant, "")
            if ri is not None:
                tr = ri.get("traffic_bytes_per_launch", {}).get(kn)
                if tr:
                    roof["traffic"] = int(tr)
                    roof["traffic_over_algorithmic"] = round(tr / per_launch_bytes, 3)
                    roof["traffic_source"] = ri.get("source", "") + ": rocprofv3 --pmc FETCH_SIZE x2 (gfx950) + WRITE_SIZE, separate passes; not measured in this run"
                fam = ri.get("families", {}).get(kn)
                if kn == "mbx" and fam and fam.get("valu_busy"):
                    busy = fam["valu_busy"] * fam["avg_launch_us"] / (avg_ms * 1e3)
                    roof.update({"bound": "valu", "achieved": round(100.0 * busy, 1), "peak": 100.0, "unit": "% of VALU issue cycles",
                                 "frac": round(busy, 4), "valu_clock_ghz": round(fam.get("clock_ghz") or 0.0, 2),
                                 "valu_source": "4 x SQ_ACTIVE_INST_VALU / (32 x SQ_BUSY_CYCLES) over the family's launches in the committed "
                                                "SQ pass, times committed / live launch duration"})
            # `traffic` (HBM bytes per launch from the PMC counters) cannot be collected inside this process: it comes from
            # separate rocprofv3 --pmc passes.  The figure of the last committed profile of this command is quoted beside it.
            tf = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tf):
                try:
                    roof["traffic_committed_profile"] = {"bytes_per_launch": json.load(open(tf)).get(KIND_NAMES.get(dominant, ""), None),
                                                         "source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, "
                                                                   "FETCH x2 on gfx950; not measured in this run)"}
                except Exception:
                    pass
        lw = LAYERWISE_MB_PER_UNIT.get((a.model, a.image_size, a.classes)) if a.variant == "full" else None
        pipeline = None
        if lw:      # whole conv stack against the layer-wise byte count SURVEY 8d prices the path with (fusion may beat it)
            pipeline = {"layerwise_MB_per_unit": lw, "layerwise_GBps": round(lw * value / 1e3, 1),
                        "note": "bookkeeping against SURVEY 8d's unfused layer-wise byte count, NOT an achieved HBM rate: the fused "
                                "kernels move far less than that",
                        "gflop_per_unit": 16.97, "tflops": round(16.97 * value / 1e3, 2)}
        step_block = None
        try:
            ri = json.load(open(os.path.join(ROOT, "profiles", "roofline_inputs.json")))
            tb = ri.get("traffic_bytes_per_launch", {})
            pl_ = drv.plan
            units, i_ = {}, 0          # launches per kind and chunk: the ops of a head layer share ONE launch (launch_group)
            while i_ < len(pl_.ops):
                o_ = pl_.ops[i_]
                units[o_["kind"]] = units.get(o_["kind"], 0) + 1
                i_ += max(1, o_.get("launch_group", 0))
            chunks = -(-a.batch // pl_.chunk_images)
            by = 0.0
            for k_, n_ in units.items():
                nm = KIND_NAMES.get(k_, "")
                if nm in tb:
                    by += tb[nm] * n_ * chunks
            for nm in ("aggregate", "nms"):
                by += tb.get(nm, 0.0)
            exp_b = 2.0 * sum(4.0 * a.batch * (pl_.T if pl_.bufs[o["out"]].per_sample else 1) * pl_.bufs[o["out"]].H * pl_.bufs[o["out"]].W * pl_.bufs[o["out"]].C
                              for o in pl_.ops if o["kind"] == 7)
            ms = elapsed / a.steps * 1e3
            step_block = {"counter_bytes_per_step": int(by), "GBps": round(by / ms / 1e6, 1), "frac_of_hbm_peak": round(by / ms / 1e6 / HBM_PEAK_GBS, 3),
                          "expanded_tensor_round_trip_bytes": int(exp_b),
                          "note": "whole step on the committed FETCH / WRITE counters (bytes per launch and kind x this plan's launches) over THIS run's "
                                  "ms_per_step; the round trip = the 6x-expanded depthwise outputs written by the fused MBConv kernels and read "
                                  "back by the 1x1 projections (algorithmic bytes, both directions)"}
        except Exception:
            pass
        line = {
            "metric": "images*MC-samples/sec, EfficientDet-D0 MC-dropout serve (preprocess+net xT+decode+NMS)",
            "value": round(value, 2), "unit": "images*MC-samples/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "warmup_steps_run": n_warm, "ms_per_step": round(elapsed / a.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if scheme == "f32" else "f32 (tensors and accumulators float32; 1x1 products on the matrix cores as %s)" % SCHEME_NOTE[scheme],
            "data": "synthetic",
            "p50_step_ms": round(float(np.median(lat)) * 1e3, 2),
            "protocol": {"name": a.protocol,
                         "note": "pipelined: the K timed steps are queued the way ServingDriver.serve_stream serves a stream of batches - "
                                 "step k + 1's network (uda_run_async) goes in before step k's detections are fetched (uda_collect), so "
                                 "step k's aggregate / NMS / gather (latency-bound launches, ~4 ms) run beside step k + 1's backbone; every "
                                 "step still does all of its work and hands its detections to the host inside the timed region (two "
                                 "barriers around exactly K steps).  serial: one step at a time (rounds 1-3).  Both are measured in this run.",
                         "other": other_protocol},
            "config": {"workload": "%s: %s, %d synthetic images (%s raw, %s network) per GPU, MC-dropout T=%d (%s), loss attenuation, "
                                   "C=%d, %s soft-NMS" % (CONFIG_NAMES[a.config], a.model, a.batch, a.raw_size, a.image_size, a.samples,
                                                          a.variant, a.classes, a.post_mode),
                       "images_per_gpu": a.batch, "mc_samples": a.samples, "chunk_images": a.chunk,
                       "weights": "random init (reference initialisers), seed 0" +
                                  ("" if a.cls_spread == 1.0 else ", class-predict kernel x %g" % a.cls_spread),
                       "contraction": "UDA_PW_SCHEME=%s: %s" % (scheme, SCHEME_NOTE[scheme]),
                       "sharding": "images across ranks, all-gather of detections", "plan": summary},
            "kernel_ms_per_step": {KIND_NAMES.get(k, str(k)): round(v[0], 2) for k, v in calib.items()},
            "h2d_upload_ms": round(upload_s * 1e3, 1),
            "nms_prefix_redone_images": pfx_fb,
            "nms_coop_fallbacks": coop_fb,
            "nms_coop_not_launched": coop_nl,
            "ranks": ranks_info,
            "roofline": roof,
            "step": step_block,
            "pipeline": pipeline,
        }
        if other_protocol is not None:      # both protocols at the top level too
            tag = other_protocol["protocol"]
            line["ms_per_step_" + tag] = other_protocol["ms_per_step"]
            line["value_" + tag] = other_protocol["value"]
        line.update(side)
        if world == 1 and not a.no_side:
            log("GPU part done (%.2f units/s); child legs: other schemes, other BASELINE configs, side regimes ..." % value)
            prec = {"main_leg_scheme": scheme,
                    "note": "same workload and the same --steps / --warmup protocol, each in a child process whose environment "
                            "carries the switch before its first GPU call; what each scheme does to heads, candidates and "
                            "detections: tests/test_gpu_fullsize.py (margin-aware), tests/test_gpu_round3.py"}
            for sch in ("bf16x3", "bf16x2", "f32"):
                if sch != scheme:
                    prec[sch] = child_leg(a, env={"UDA_PW_SCHEME": sch})
                    prec[sch]["scheme"] = SCHEME_NOTE[sch]
                    log("scheme %s: %s ms/step" % (sch, prec[sch].get("ms_per_step")))
            line["precision"] = prec
            if scheme in ("f16x2", "bf16x3"):
                line["value_fp32_equivalent"] = line["value"]           # the main leg already runs float32-class products
            line["value_bf16x2"] = prec.get("bf16x2", {}).get("value")  # narrower than float32: a side figure, never `value`
            if a.config == 1 and a.variant == "full" and a.cls_spread == 1.0:
                cfgs = {}
                for n in (2, 3, 4):
                    cfgs[str(n)] = child_leg(a, ["--config", str(n)], same_shape=False)
                    cfgs[str(n)]["config"] = CONFIG_NAMES[n]
                    log("configs[%d]: %s ms/step" % (n, cfgs[str(n)].get("ms_per_step")))
                line["configs"] = cfgs
                ho = child_leg(a, ["--variant", "head"], same_shape=False)
                ho["algorithmic_GMAC_per_unit"] = 1.545      # (backbone + FPN) / T + heads at D0, T = 10
                ho["note"] = "head-only MC dropout (mc_classheadrate = mc_boxheadrate = 0.05, mc_dropoutrate = 0): backbone + BiFPN once per image"
                line["head_only"] = ho
                line["nms_spread_scores"] = child_leg(a, ["--cls-spread", "20"], same_shape=False)
                line["rccl_world1"] = child_leg(a, ["--force-dist"], same_shape=False)
                log("head-only %s, spread scores %s, process-group path %s ms/step" % (
                    ho.get("ms_per_step"), line["nms_spread_scores"].get("ms_per_step"), line["rccl_world1"].get("ms_per_step")))
        if world == 1 and not a.no_cpu_baseline:
            log("timing the CPU oracle on a bounded sample ...")
            line["cpu_baseline"] = cpu_baseline(a)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def ensemble_main(a, params, images, rank, world, local_rank, dist, tdev, ranks_info=None):
    """BASELINE configs[3]: M deterministic members (independent random-init weight sets), aggregated like MC samples.
    One GPU: EnsembleDriver.  Several ranks: members striped round-robin, head outputs re-sharded by image over RCCL,
    every rank aggregates / NMSes its image shard, one all-gather of the detections (dist.serve_ensemble_striped)."""
    from uda_amd import weights as weights_mod
    from uda_amd.infer_lib import EnsembleDriver, ServingDriver
    M = a.ensemble
    n_total = a.batch * world                      # weak scaling: the batch grows with the ranks, members stay M
    if world == 1:
        ws = [weights_mod.init_weights(params, seed=40 + m) for m in range(M)]
        ens = EnsembleDriver(ws, a.model, batch_size=a.batch, model_params=params, device=local_rank, chunk_images=a.chunk)
        run = lambda: ens.serve(images)
        close = ens.close
    else:
        from uda_amd import dist as udist
        allimg = np.concatenate([np.random.default_rng(2 + r).integers(0, 256, images.shape, dtype=np.uint8) for r in range(world)])
        mine = {m: ServingDriver(a.model, n_total, False, params, weights=weights_mod.init_weights(params, seed=40 + m),
                                 device=local_rank, chunk_images=a.chunk) for m in range(M) if udist.member_owner(m, world) == rank}
        pm = dict(params, mc_dropout=True, mc_dropoutrate=1e-9, mc_dropoutsamp=M)
        post = ServingDriver(a.model, n_total, False, pm, post_only=True, device=local_rank, chunk_images=1)
        run = lambda: udist.serve_ensemble_striped(mine, post, allimg, M, rank, world, device=tdev)

        def close():
            for d in list(mine.values()) + [post]:
                d.close()
    for _ in range(max(2, a.warmup)):
        run()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    if dist is not None:
        import torch
        torch.cuda.synchronize()
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    units = n_total * M * a.steps
    if rank == 0 and a.child:
        print(json.dumps({"ms_per_step": round(elapsed / a.steps * 1e3, 2), "value": round(units / elapsed, 2), "unit": "images*members/s",
                          "steps": a.steps, "warmup": a.warmup, "UDA_PW_SCHEME": __import__("uda_amd.plan", fromlist=["x"]).pw_scheme(),
                          "workload": "%d-member deep ensemble of %s, %d images (%s) per GPU, uploads included" % (M, a.model, a.batch, a.image_size)}),
              flush=True)
    elif rank == 0:
        print(json.dumps({
            "metric": "images*ensemble-members/sec, 5-member deep ensemble of EfficientDet-D0 (preprocess+net+aggregate+decode+NMS)",
            "value": round(units / elapsed, 2), "unit": "images*members/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (1x1 products: UDA_PW_SCHEME=%s)" % __import__("uda_amd.plan", fromlist=["x"]).pw_scheme(), "data": "synthetic",
            "config": {"workload": "%s: %d-member deep ensemble of %s, %d synthetic images (%s) per GPU, members striped over %d rank(s), "
                                   "uploads included" % (CONFIG_NAMES[3], M, a.model, a.batch, a.image_size, world),
                       "images_per_gpu": a.batch, "members": M}, "ranks": ranks_info}), flush=True)
    close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
