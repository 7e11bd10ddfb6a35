#!/usr/bin/env python
"""Map a rocprofv3 --kernel-trace CSV of bench.py onto the op list: per-op duration,
achieved algorithmic GB/s and TFLOP/s.   python tools/analyze_trace.py trace.csv [--chunk 16]"""
import argparse
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uda_amd import capi, hparams_config, plan as plan_mod, weights as weights_mod  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--chunk", type=int, default=16)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--samples", type=int, default=10)
ap.add_argument("--variant", default="full")
ap.add_argument("--model", default="efficientdet-d0")
ap.add_argument("--image-size", default="1280x768")
ap.add_argument("--classes", type=int, default=7)
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--fetch", default="", help="counter_collection.csv of a --pmc FETCH_SIZE run of the same command")
ap.add_argument("--write", default="", help="counter_collection.csv of a --pmc WRITE_SIZE run")
a = ap.parse_args()

cfg = hparams_config.get_efficientdet_config(a.model)
over = dict(image_size=a.image_size, num_classes=a.classes, mc_dropout=True, mc_dropoutsamp=a.samples, loss_attenuation=True,
            enable_softmax=True)
over.update(dict(mc_dropoutrate=0.05) if a.variant == "full" else dict(mc_classheadrate=0.05, mc_boxheadrate=0.05))
cfg.override(over)
p = cfg.as_dict()
pl = plan_mod.Plan(p, weights_mod.init_weights(p, 0), chunk_images=a.chunk, max_images=a.batch)

def pmc_per_op(path, counter, scale):
    """bytes per op of the first chunk of the last step, in op order (same selection as the trace)."""
    rr = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rr.sort(key=lambda r: int(r["Dispatch_Id"]))
    ix = [i for i, r in enumerate(rr) if "::stem" in r["Kernel_Name"]]      # the stem opens a chunk's op list
    rr = rr[ix[-1]:]
    cv = [r for r in rr if any(n in r["Kernel_Name"] for n in names)]
    return [float(r["Counter_Value"]) * 1024 * scale for r in cv[:len(units)]]


rows = list(csv.DictReader(open(a.trace)))
names = ("stem_kernel", "stem16_kernel", "stem_u8_kernel", "sepf_kernel", "pw_kernel", "pwb_kernel", "pws_kernel", "pwb_shared_kernel", "dw_kernel", "se_kernel", "fuse_kernel", "mbx_kernel", "mbxb_kernel", "mbxd_kernel", "mbxp_kernel", "sep_kernel")
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "::stem" in r["Kernel_Name"]]      # the stem opens a chunk's op list (last chunk of the last step)
rows = rows[idx[-1]:]
conv = [r for r in rows if any(n in r["Kernel_Name"] for n in names)]
# launch units: one op, or the ops of a head layer that share a launch (launch_group, plan.py)
units, i = [], 0
while i < len(pl.ops):
    n = max(1, pl.ops[i].get("launch_group", 0))
    units.append(pl.ops[i:i + n])
    i += n
nops = len(units)
assert len(conv) % nops == 0, (len(conv), nops)
first = conv[:nops]                       # first chunk of the last step
T = pl.T
res = []
size = lambda bi: a.chunk * (T if pl.bufs[bi].per_sample else 1) * pl.bufs[bi].H * pl.bufs[bi].W * pl.bufs[bi].C
for unit, r in zip(units, first):
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if len(unit) > 1:                     # separable convs of one layer on all pyramid levels
        by = sum(size(o["out"]) + size(o["ins"][0]) for o in unit)
        fl = sum(2 * a.chunk * (T if pl.bufs[o["out"]].per_sample else 1) * pl.bufs[o["out"]].H * pl.bufs[o["out"]].W *
                 (9 * pl.bufs[o["ins"][0]].C + pl.bufs[o["ins"][0]].C * pl.bufs[o["out"]].C) for o in unit)
        ob = pl.bufs[unit[0]["out"]]
        desc = "sep %d->%d x%d levels" % (pl.bufs[unit[0]["ins"][0]].C, ob.C, len(unit))
        res.append((dur, desc, ob.name.rsplit("-", 1)[0], by * 4, fl, r["Kernel_Name"].split("(")[0][-16:], r["VGPR_Count"]))
        continue
    o = unit[0]
    ob = pl.bufs[o["out"]]
    by = size(o["out"]) + sum(size(i) for i in (o["ins"][:1] if o["kind"] == capi.OP_SE else o["ins"]))
    for k in ("se_scale", "residual", "se_partial"):
        if o[k] >= 0 and o["kind"] != capi.OP_SE:
            by += size(o[k])
    fl = 0
    rows_ = a.chunk * (T if ob.per_sample else 1)
    if o["kind"] == capi.OP_PW:
        cin = pl.bufs[o["ins"][0]].C
        fl = 2 * rows_ * ob.H * ob.W * cin * ob.C
        desc = "pw %4d->%-4d @%dx%d" % (cin, ob.C, ob.H, ob.W)
    elif o["kind"] == capi.OP_DW:
        fl = 2 * rows_ * ob.H * ob.W * ob.C * o["k"] ** 2
        desc = "dw k%d s%d C=%-4d @%dx%d" % (o["k"], o["stride"], ob.C, ob.H, ob.W)
    elif o["kind"] == capi.OP_MBX:
        ib = pl.bufs[o["ins"][0]]
        fl = 2 * rows_ * (ib.H * ib.W * ib.C * ob.C + ob.H * ob.W * ob.C * o["k"] ** 2)
        desc = "mbx %d->%d k%d s%d @%dx%d" % (ib.C, ob.C, o["k"], o["stride"], ob.H, ob.W)
    elif o["kind"] == capi.OP_SEP:
        ib = pl.bufs[o["ins"][0]]
        fl = 2 * rows_ * ob.H * ob.W * (9 * ib.C + ib.C * ob.C)
        desc = "sep %d->%d @%dx%d" % (ib.C, ob.C, ob.H, ob.W) + (" fin%d" % len(o["ins"]) if o.get("fuse_in") else "")
    else:
        desc = {1: "stem", 4: "se", 5: "fuse", 6: "pool"}[o["kind"]] + " C=%d @%dx%d" % (ob.C, ob.H, ob.W)
    res.append((dur, desc, ob.name, by * 4, fl, r["Kernel_Name"].split("(")[0][-16:], r["VGPR_Count"]))
fetch = pmc_per_op(a.fetch, "FETCH_SIZE", 2.0) if a.fetch else None   # gfx950: 64 B counted per 128-B request
write = pmc_per_op(a.write, "WRITE_SIZE", 1.0) if a.write else None
if fetch or write:
    res = [r + ((fetch[i] if fetch else 0.0) + (write[i] if write else 0.0),) for i, r in enumerate(res)]
    if write:       # write counter against the op's output size (a ratio well above 1: partially written lines)
        outs = [sum(size(o["out"]) + (size(o["se_partial"]) if o["se_partial"] >= 0 and o["kind"] != capi.OP_SE else 0) for o in unit) * 4 for unit in units]
        res = [r + (write[i] / max(outs[i], 1),) for i, r in enumerate(res)]
tot = sum(r[0] for r in res)
print("chunk of %d images: %d ops in %d launches, %.2f ms kernel time" % (a.chunk, len(pl.ops), nops, tot / 1e3))
for kind in ("pw", "dw", "mbx", "sep", "se", "fuse", "stem", "pool"):
    sel = [r for r in res if r[1].split()[0] == kind]
    if sel:
        t = sum(r[0] for r in sel)
        print("  %-5s %8.2f ms  %7.1f GB/s  %6.1f TFLOP/s" % (kind, t / 1e3, sum(r[3] for r in sel) / t / 1e3,
                                                              sum(r[4] for r in sel) / t / 1e6))
print("top ops:")
for r in sorted(res, reverse=True)[:a.top]:
    dur, desc, name, by, fl, kn, vg = r[:7]
    extra = "  hbm %7.1f MB = %.2fx alg, %6.0f GB/s" % (r[7] / 1e6, r[7] / by, r[7] / dur / 1e3) if len(r) > 7 else ""
    if len(r) > 8:
        extra += "  wr %.2fx" % r[8]
    print("  %8.1f us  %-28s %-22s %7.1f GB/s %6.1f TF/s  %s v%s%s" % (dur, desc, name, by / dur / 1e3, fl / dur / 1e6, kn, vg, extra))
