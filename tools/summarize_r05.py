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
