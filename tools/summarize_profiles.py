#!/usr/bin/env python
"""Copy the judged summaries of a tools/collect_profiles.sh run from gpurun_out/ into profiles/."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
stats = newest(os.path.join(root, "gpurun_out", tag + "_stats", "*", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(out, tag + "_kernel_stats.csv"))
fetch = newest(os.path.join(root, "gpurun_out", tag + "_fetch", "*", "*counter_collection.csv"))
write = newest(os.path.join(root, "gpurun_out", tag + "_write", "*", "*counter_collection.csv"))
traffic = os.path.join(out, tag + "_traffic.json")
subprocess.check_call([sys.executable, os.path.join(root, "tools", "traffic_from_pmc.py"), fetch, write, traffic])
t = json.load(open(traffic))
json.dump({k: v["hbm_bytes_per_launch"] for k, v in t.items()}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
bench = os.path.join(root, "gpurun_out", tag + "_bench.json")
line = [l for l in open(bench) if l.startswith("{")][-1]
open(os.path.join(out, tag + "_bench.json"), "w").write(line)
b = json.loads(line)
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(out, tag + "_summary.md"), "w") as f:
    f.write("# %s profile summary (MI355X, `python bench.py`, BASELINE configs[1])\n\n" % tag)
    f.write("bench line: **%.1f %s**, %.2f ms/step, p50 %.2f ms; roofline %s\n\n" % (
        b["value"], b["unit"], b["ms_per_step"], b.get("p50_step_ms", b.get("p50_serve_latency_ms", 0.0)), json.dumps(b["roofline"])))
    f.write("cpu_baseline: %s\n\n" % json.dumps(b.get("cpu_baseline")))
    f.write("## rocprofv3 --kernel-trace --stats (3 timed + 1 warm-up step; `%s_kernel_stats.csv`)\n\n" % tag)
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:24]:
        f.write("| `%s` | %s | %.2f | %.1f | %s |\n" % (r["Name"].replace("void ", "").replace("uda::", "")[:60], r["Calls"],
                                                      float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    f.write("\n## HBM traffic from PMC (`%s_traffic.json`; FETCH_SIZE x2 on gfx950, separate passes)\n\n" % tag)
    f.write("| kernel | launches | read MB/launch | write MB/launch |\n|---|---|---|---|\n")
    for k, v in t.items():
        f.write("| `%s` | %d | %.1f | %.1f |\n" % (k, v["launches"], v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))
print(open(os.path.join(out, tag + "_summary.md")).read()[:1500])
