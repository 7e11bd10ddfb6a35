#!/usr/bin/env python
"""gpurun_out/<tag>_* (tools/collect_r05.sh <tag> "<bench args>") -> profiles/<tag>_*:
  <tag>_kernel_stats.csv  rocprofv3 --kernel-trace --stats       <tag>_per_op.txt  every op with duration, rate, FETCH / WRITE traffic
  <tag>_traffic.json      HBM bytes per launch and kernel kind   <tag>_sq.txt      SQ pass(es) as collected (tools/pmc_table.py)
  <tag>_bench.json        the bench line of the same configuration
usage: tools/summarize_r05.py <tag> [analyze_trace args, e.g. --variant head | --model efficientdet-d2 --image-size 1024x1024 --batch 2 --chunk 2 --samples 30]"""
import json
import os
import subprocess
import sys

tag = sys.argv[1]
extra = sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = lambda n: os.path.join(root, "gpurun_out", "%s_%s" % (tag, n))
out = os.path.join(root, "profiles")
o = lambda n: os.path.join(out, "%s_%s" % (tag, n))
open(o("kernel_stats.csv"), "w").write(open(g("kernel_stats.csv")).read())
subprocess.check_call([sys.executable, os.path.join(root, "tools", "traffic_from_pmc.py"), g("FETCH_SIZE.csv"), g("WRITE_SIZE.csv"), o("traffic.json")])
args = [sys.executable, os.path.join(root, "tools", "analyze_trace.py"), g("kernel_trace.csv"), "--fetch", g("FETCH_SIZE.csv"),
        "--write", g("WRITE_SIZE.csv"), "--top", "70"] + (extra if "--chunk" in extra else extra + ["--chunk", "32"])
r = subprocess.run(args, capture_output=True, text=True)
open(o("per_op.txt"), "w").write(r.stdout + r.stderr[-2000:])
with open(o("sq.txt"), "w") as f:
    f.write("# SQ counters per kernel, mean per launch (rocprofv3 --pmc passes of the same command, tools/collect_r05.sh; tools/pmc_table.py)\n")
    for i in (1, 2, 3, 4):
        if os.path.exists(g("sq%d.txt" % i)):
            f.write("\n## pass %d\n" % i + open(g("sq%d.txt" % i)).read())
line = [l for l in open(g("bench.json")) if l.startswith("{")][-1]
open(o("bench.json"), "w").write(line)
b = json.loads(line)
print(tag, b["ms_per_step"], b["value"], b["kernel_ms_per_step"])
print(r.stdout[:6000])

# ---- what bench.py's `roofline` block quotes from the committed profile (headline tag only): VALU-busy share and clock of
# the dominant family from SQ pass 1, counter traffic per launch and kind, launches per step
if tag == "r05":
    import csv
    short = lambda n: n.replace("void ", "").replace("uda::", "").split("(")[0]
    stats = {short(r["Name"]): r for r in csv.DictReader(open(g("kernel_stats.csv")))}
    rows = [l.rstrip("\n") for l in open(g("sq1.txt"))]
    cols = rows[0].split()[2:]
    fam = {}
    for l in rows[1:]:
        name, rest = l[:34].strip().replace("uda::", ""), l[34:].split()
        if not rest:
            continue
        d = dict(zip(cols, [float(x) for x in rest[1:]]))
        n = int(rest[0])
        key = [k for k in stats if k.startswith(name[:30])]
        if not key:
            continue
        kind = "mbx" if name.startswith("mbx") else ("pw" if name.startswith("pwb") else ("sep" if name.startswith("sep") else None))
        if kind is None:
            continue
        valu = d.get("ACTIVE_INST_VALU", d.get("SQ_ACTIVE_INST_VALU", 0.0))
        busy = d.get("SQ_BUSY_CYCLES", 0.0)
        us = float(stats[key[0]]["AverageNs"]) / 1e3
        f = fam.setdefault(kind, dict(valu4=0.0, busy32=0.0, us=0.0, n=0))
        f["valu4"] += n * 4.0 * valu
        f["busy32"] += n * 32.0 * busy
        f["us"] += n * us
        f["n"] += n
    t = json.load(open(o("traffic.json")))
    step_launches = {k: v["launches"] / 1.0 for k, v in t.items()}       # (FETCH pass: 1 timed + 1 warm-up step... recorded as collected)
    inputs = {"source": "profiles/r05_* (tools/collect_r05.sh r05, tools/summarize_r05.py r05)",
              "families": {k: dict(valu_busy=v["valu4"] / v["busy32"] if v["busy32"] else None,
                                   clock_ghz=v["busy32"] / 32.0 / 32.0 / (v["us"] * 1e-6) / 1e9 if v["us"] else None,
                                   avg_launch_us=v["us"] / v["n"]) for k, v in fam.items()},
              "traffic_bytes_per_launch": {k: v["hbm_bytes_per_launch"] for k, v in t.items()},
              "launches_in_counter_pass": {k: v["launches"] for k, v in t.items()}}
    json.dump(inputs, open(os.path.join(out, "roofline_inputs.json"), "w"), indent=1)
    print(json.dumps(inputs["families"], indent=1))
